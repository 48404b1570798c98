"""Write ``tests/golden/upstream_<case>.npz`` from the REAL upstream packages.

Run this ONCE on any machine that has the reference's dependencies installed --

    pip install faster-whisper==1.2.1 ctranslate2==4.7.1 openai-whisper==20250625 silero-vad==6.2.1 auditok==0.3.0 soundfile
    python -c "import torch; [torch.hub.load('snakers4/silero-vad:' + v, 'silero_vad', onnx=False, trust_repo=True) for v in ('v3.1', 'v4.0')]"
    python scripts/make_upstream_fixtures.py --include-archives          # then commit tests/golden/upstream_*

-- and the wheel-gated tests (tests/test_upstream_wheels.py, test_pooling_host.py::test_pcm16_round_trip_matches_soundfile,
test_segmenters.py::test_silero_torchscript_archives_light_up_when_present) run from the fixtures on every box without the
wheels, this offline build container included: the oracle's parity stops being "unpinned" for that stage.

What is stored: the packages' OUTPUTS on inputs that both sides regenerate from seeds (tests/upstream_cases.py holds the
case functions and the seeds) -- never package source.  ``--include-archives`` also copies the two torch.hub Silero archives
(MIT-licensed model files) next to the fixtures, which lets the graph loader be pinned against the REAL v3.1 / v4.0 graphs offline.

    python scripts/make_upstream_fixtures.py --status        # per case: live | fixture | unpinned
    python scripts/make_upstream_fixtures.py [case ...]       # only some cases
"""
import argparse
import json
import platform
import shutil
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from tests import upstream_cases as U       # noqa: E402

PACKAGES = ("faster_whisper", "ctranslate2", "whisper", "silero_vad", "auditok", "soundfile", "torch", "numpy")


def versions():
    out = {"python": platform.python_version(), "platform": platform.platform()}
    for name in PACKAGES:
        try:
            mod = __import__(name)
            out[name] = getattr(mod, "__version__", "present")
        except Exception as e:          # noqa: BLE001
            out[name] = f"absent ({type(e).__name__})"
    return out


def write(cases, include_archives: bool):
    report = {}
    U.GOLDEN.mkdir(parents=True, exist_ok=True)
    for case in cases:
        if case in U.LIVE_ONLY:
            report[case] = {"status": "live-only (nothing to commit)"}
            continue
        t0 = time.perf_counter()
        try:
            arrays = U.CASES[case]()
        except (ImportError, U.Unavailable, OSError) as e:
            report[case] = {"status": "skipped", "why": f"{type(e).__name__}: {e}"}
            continue
        archive = str(arrays.get("archive_path", "")) if "archive_path" in arrays else ""
        if "archive_path" in arrays:
            arrays = dict(arrays, archive_path=np.asarray(""))          # a path of the generating machine means nothing elsewhere
        np.savez_compressed(U.fixture_path(case), **arrays)
        entry = {"status": "written", "file": str(U.fixture_path(case).relative_to(ROOT)) if ROOT in U.fixture_path(case).parents else str(U.fixture_path(case)),
                 "bytes": U.fixture_path(case).stat().st_size, "arrays": len(arrays), "seconds": round(time.perf_counter() - t0, 1)}
        if include_archives and archive and Path(archive).exists():
            shutil.copyfile(archive, U.archive_path(case))
            entry["archive"] = U.archive_path(case).name
        report[case] = entry
    return report


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("cases", nargs="*", help=f"default: all of {sorted(U.CASES)}")
    ap.add_argument("--status", action="store_true", help="print, per case, whether it is live / has a fixture / is unpinned, and exit")
    ap.add_argument("--include-archives", action="store_true", help="also copy the torch.hub Silero v3.1 / v4.0 archives into tests/golden/")
    args = ap.parse_args()
    cases = args.cases or sorted(U.CASES)
    unknown = [c for c in cases if c not in U.CASES]
    if unknown:
        raise SystemExit(f"unknown case(s) {unknown}; known: {sorted(U.CASES)}")
    if args.status:
        for c in cases:
            print(f"{c:28s} {U.status(c)}")
        return 0
    report = write(cases, args.include_archives)
    manifest = {"generated": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime()), "versions": versions(), "cases": report}
    (U.GOLDEN / "upstream_manifest.json").write_text(json.dumps(manifest, indent=1) + "\n")
    print(json.dumps(manifest, indent=1))
    return 0 if any(r.get("status") == "written" for r in report.values()) else 1


if __name__ == "__main__":
    sys.exit(main())
