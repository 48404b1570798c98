"""Generates tests/golden/reference_scenes.json by running the REFERENCE's own two-pass scene driver
(/root/reference/whisperjav/modules/scene_detection_backends/auditok_backend.py, imported from source) on seeded
synthetic audio.  The third-party pieces that are not installable offline are replaced at import time:
``auditok.split`` by oracle/auditok_ref.split (the restatement under test elsewhere), ``soundfile`` / ``librosa`` by
stubs (nothing is read or written: ``save_scene_wav`` is patched out).  What this pins is the reference's driver
logic -- pass-1 / pass-2 parameters, the direct / granular / brute-force branches, min-duration filtering, clamping,
scene order -- bit for bit, independent of who wrote the tokenizer.

Run from the repo root inside the build container:  python tests/golden/make_scene_fixtures.py
"""
import importlib
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from oracle import auditok_ref  # noqa: E402
from whisperjav_amd import synth  # noqa: E402


class _Region:
    def __init__(self, start, end):
        self.start, self.end = start, end


def _fake_split(data, sampling_rate=16000, channels=1, sample_width=2, min_dur=0.2, max_dur=5, max_silence=0.3,
                energy_threshold=50, drop_trailing_silence=False, **kw):
    assert channels == 1 and sample_width == 2
    pcm = np.frombuffer(data, dtype=np.int16)
    return iter([_Region(a, b) for a, b in auditok_ref.split(pcm, sampling_rate, min_dur, max_dur, max_silence, energy_threshold,
                                                              drop_trailing_silence)])


def _install_stubs():
    aud = types.ModuleType("auditok")
    aud.split = _fake_split
    sys.modules["auditok"] = aud
    for name in ("soundfile", "librosa"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["soundfile"].SoundFileError = type("SoundFileError", (Exception,), {})
    # the package __init__ files pull in the whole application: load the three modules we need by path instead
    pkg = types.ModuleType("whisperjav"); pkg.__path__ = ["/root/reference/whisperjav"]
    mods = types.ModuleType("whisperjav.modules"); mods.__path__ = ["/root/reference/whisperjav/modules"]
    sdb = types.ModuleType("whisperjav.modules.scene_detection_backends")
    sdb.__path__ = ["/root/reference/whisperjav/modules/scene_detection_backends"]
    sys.modules.update({"whisperjav": pkg, "whisperjav.modules": mods, "whisperjav.modules.scene_detection_backends": sdb})
    return importlib.import_module("whisperjav.modules.scene_detection_backends.auditok_backend")


CASES = [
    dict(seed=1, seconds=95.0, noisy=False, cfg={}),
    dict(seed=2, seconds=140.0, noisy=True, cfg={}),
    dict(seed=3, seconds=61.0, noisy=False, cfg=dict(max_duration=12.0, pass1_max_silence=0.6, pass2_max_silence=0.3)),
    dict(seed=4, seconds=80.0, noisy=True, cfg=dict(max_duration=8.0, min_duration=1.0, pass2_energy_threshold=80)),   # brute force
    dict(seed=5, seconds=33.0, noisy=False, cfg=dict(pass1_energy_threshold=20, pad_edges_s=0.25)),
    dict(seed=6, seconds=7.3, noisy=False, cfg=dict(pass1_energy_threshold=95)),                                        # nothing found
    dict(seed=7, seconds=200.0, noisy=True, cfg=dict(pass1_max_silence=2.5, max_duration=29.0)),
    # gates above the -45 dBFS noise floor: the silence logic of both passes is exercised
    dict(seed=8, seconds=200.0, noisy=False, cfg=dict(pass1_energy_threshold=52, pass2_energy_threshold=56)),
    dict(seed=9, seconds=150.0, noisy=False, cfg=dict(pass1_energy_threshold=55, pass2_energy_threshold=60, max_duration=10.0,
                                                       pass1_max_silence=1.0, pass2_max_silence=0.4)),
    dict(seed=10, seconds=120.0, noisy=False, cfg=dict(pass1_energy_threshold=50, pass2_energy_threshold=58, max_duration=6.0,
                                                        min_duration=0.5, pass2_min_duration=0.6, pad_edges_s=0.1)),
    dict(seed=11, seconds=45.0, noisy=False, cfg=dict(pass1_energy_threshold=60, pass2_energy_threshold=64, max_duration=3.0,
                                                       pass1_max_silence=0.3, pass2_max_silence=0.15)),
]


def main():
    backend = _install_stubs()
    saved = []
    backend.save_scene_wav = lambda audio, sr, idx, out_dir, base: saved.append(len(audio)) or f"{base}_scene_{idx:04d}.wav"
    out = []
    for case in CASES:
        audio = synth.speech_like(case["seconds"], seed=case["seed"], noisy=case["noisy"])
        det = backend.AuditokSceneDetector(config=backend.AuditokSceneConfig(**case["cfg"]))
        total = len(audio) / 16000
        story = det._detect_pass1(audio, 16000, total)
        saved.clear()
        scenes, counters = det._process_story_lines(story, audio, 16000, total, None, "clip")
        out.append({**case, "story": [(r.start, r.end) for r in story],
                    "scenes": [(s.start_sec, s.end_sec, s.detection_pass, s.metadata.get("split_method")) for s in scenes],
                    "scene_samples": list(saved), "counters": counters})
        print(case["seed"], len(story), len(scenes), counters)
    defaults = backend.AuditokSceneConfig()
    with open(os.path.join(HERE, "reference_scenes.json"), "w") as f:
        json.dump({"cases": out, "config_defaults": {k: getattr(defaults, k) for k in defaults.__dataclass_fields__}}, f)


if __name__ == "__main__":
    main()
