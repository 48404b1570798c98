"""Golden vectors produced by RUNNING THE REFERENCE'S OWN CODE in the build container:

    python tests/golden/make_reference_fixtures.py

The pure-Python pieces of the hot path that import standalone are loaded by
file path from /root/reference (exactly like the reference's tests do,
tests/test_vad_threshold_padding_e2e.py:41-70) and driven with seeded inputs; the outputs are committed as
tests/golden/reference_grouping.json so the parity tests do not need /root/reference at run time.

  * ``group_segments``            (speech_segmentation/backends/ten.py:31-73)
  * ``SegmentationResult.to_legacy_format`` (speech_segmentation/base.py:98-122)
  * ``should_force_full_transcribe``        (modules/vad_failover.py:26-57)
  * ``SegmentFilterHelper.should_filter``   (modules/segment_filters.py:80-103), logprob gate
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np

REF = "/root/reference/whisperjav/modules"
HERE = os.path.dirname(os.path.abspath(__file__))


def load(name, path, package=None):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def main():
    # package skeleton so that ten.py's "from ..base import ..." resolves
    for pkg in ("whisperjav", "whisperjav.modules", "whisperjav.modules.speech_segmentation",
                "whisperjav.modules.speech_segmentation.backends"):
        m = types.ModuleType(pkg)
        m.__path__ = []
        sys.modules.setdefault(pkg, m)
    base = load("whisperjav.modules.speech_segmentation.base", f"{REF}/speech_segmentation/base.py")
    ten = load("whisperjav.modules.speech_segmentation.backends.ten", f"{REF}/speech_segmentation/backends/ten.py")
    failover = load("ref_vad_failover", f"{REF}/vad_failover.py")
    filters = load("ref_segment_filters", f"{REF}/segment_filters.py")

    rng = np.random.default_rng(20240923)
    cases = []
    for _ in range(60):
        n = int(rng.integers(0, 14))
        t = 0.0
        segs = []
        for _ in range(n):
            t += float(rng.choice([0.05, 0.3, 0.9, 1.1, 2.7, 4.5])) * float(rng.uniform(0.5, 1.5))
            dur = float(rng.uniform(0.2, 5.0))
            segs.append((round(t, 4), round(t + dur, 4)))
            t += dur
        max_group = float(rng.choice([6.0, 8.0, 29.0]))
        chunk = float(rng.choice([1.0, 2.5, 4.0]))
        ref_segs = [base.SpeechSegment(start_sec=a, end_sec=b, start_sample=int(a * 16000), end_sample=int(b * 16000))
                    for a, b in segs]
        groups = ten.group_segments(ref_segs, max_group, chunk)
        res = base.SegmentationResult(segments=ref_segs, groups=groups, method="x", audio_duration_sec=t + 1.0,
                                      parameters={})
        cases.append({"segments": segs, "max_group_duration_s": max_group, "chunk_threshold_s": chunk,
                      "groups": [[(s.start_sec, s.end_sec) for s in g] for g in groups],
                      "legacy": res.to_legacy_format(), "coverage": res.speech_coverage_sec})
    fo = []
    for _ in range(40):
        dur = float(rng.choice([30.0, 119.0, 121.0, 400.0, 700.0, 2000.0]))
        k = int(rng.integers(0, 5))
        groups = [[{"start_sec": float(a), "end_sec": float(a + rng.uniform(0.1, 3.0))} for a in rng.uniform(0, dur, size=int(rng.integers(0, 3)))]
                  for _ in range(k)]
        fo.append({"groups": groups, "duration": dur, "force": bool(failover.should_force_full_transcribe(groups, dur))})
    fl = []
    for _ in range(40):
        cfg = dict(enabled=bool(rng.integers(0, 2)), logprob_threshold=float(rng.choice([-1.0, -0.5])),
                   logprob_margin=float(rng.choice([0.0, 0.2])), drop_nonverbal_vocals=False)
        helper = filters.SegmentFilterHelper(filters.SegmentFilterConfig(**cfg))
        lp, dur = float(rng.uniform(-1.6, 0.0)), float(rng.uniform(0.2, 4.0))
        drop, reason, thr = helper.should_filter(avg_logprob=lp, duration=dur, text="こんにちは")
        fl.append({"cfg": cfg, "avg_logprob": lp, "duration": dur, "drop": drop, "reason": reason, "threshold": thr})
    # ---- the reference's Silero backends driven with a fake scorer (its own test seam) -------------
    from unittest.mock import MagicMock
    fake = types.ModuleType("silero_vad")
    box = {"stamps": []}
    fake.get_speech_timestamps = lambda audio, model, **kw: [dict(t) for t in box["stamps"]]
    fake.load_silero_vad = lambda *a, **k: MagicMock()
    sys.modules["silero_vad"] = fake
    silero = load("whisperjav.modules.speech_segmentation.backends.silero", f"{REF}/speech_segmentation/backends/silero.py")
    silero_v6 = load("whisperjav.modules.speech_segmentation.backends.silero_v6",
                     f"{REF}/speech_segmentation/backends/silero_v6.py")
    sil = []
    for _ in range(50):
        n_samples = int(rng.choice([40000, 160000, 464000]))
        k = int(rng.integers(0, 8))
        marks = sorted(int(x) for x in rng.integers(0, n_samples, size=2 * k))
        stamps = [{"start": marks[2 * i], "end": max(marks[2 * i] + 1, marks[2 * i + 1])} for i in range(k)]
        kw = dict(version=str(rng.choice(["v3.1", "v4.0"])), start_pad_samples=int(rng.choice([0, 3200, 11200])),
                  end_pad_samples=int(rng.choice([0, 6400, 20800])), chunk_threshold_s=float(rng.choice([1.0, 2.5, 4.0])),
                  max_group_duration_s=float(rng.choice([6.0, 29.0])))
        seg = silero.SileroSpeechSegmenter(**kw)
        seg._model = MagicMock()
        seg._utils = (None,) * 5
        seg._get_speech_timestamps = lambda audio, model, **kw2: [dict(t) for t in stamps]
        res = seg.segment(np.zeros(n_samples, dtype=np.float32), sample_rate=16000)
        sil.append({"kw": kw, "n_samples": n_samples, "stamps": stamps, "name": seg.name,
                    "segments": [(s.start_sample, s.end_sample, s.start_sec, s.end_sec) for s in res.segments],
                    "groups": [[(s.start_sample, s.end_sample) for s in g] for g in res.groups],
                    "legacy": res.to_legacy_format()})
    v6 = []
    for _ in range(30):
        n_samples = int(rng.choice([40000, 160000, 464000]))
        k = int(rng.integers(0, 8))
        marks = sorted(int(x) for x in rng.integers(0, n_samples, size=2 * k))
        box["stamps"] = [{"start": marks[2 * i], "end": max(marks[2 * i] + 1, marks[2 * i + 1])} for i in range(k)]
        kw = dict(threshold=0.35, chunk_threshold_s=float(rng.choice([1.0, 2.5])),
                  max_group_duration_s=float(rng.choice([6.0, 29.0])))
        seg = silero_v6.SileroV6SpeechSegmenter(**kw)
        res = seg.segment(np.zeros(n_samples, dtype=np.float32), sample_rate=16000)
        v6.append({"kw": kw, "n_samples": n_samples, "stamps": list(box["stamps"]), "params": seg._get_parameters(),
                   "segments": [(s.start_sample, s.end_sample, s.start_sec, s.end_sec) for s in res.segments],
                   "groups": [[(s.start_sample, s.end_sample) for s in g] for g in res.groups]})
    with open(os.path.join(HERE, "reference_grouping.json"), "w") as f:
        json.dump({"group_cases": cases, "failover_cases": fo, "filter_cases": fl, "silero_cases": sil,
                   "silero_v6_cases": v6}, f)
    print("wrote", len(cases), len(fo), len(fl), len(sil), len(v6))


if __name__ == "__main__":
    main()
