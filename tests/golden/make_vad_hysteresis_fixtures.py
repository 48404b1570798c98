"""Generates tests/golden/reference_vad_hysteresis.json: the hysteresis path of the Silero-compatible state machine
(onset at threshold, candidate offset below threshold - 0.15, confirmation after the minimum silence, minimum speech
length, speech running to the end of the audio) as computed by the REFERENCE's own pure-Python port,
``WhisperSegSpeechSegmenter._probs_to_segments`` (/root/reference/whisperjav/modules/speech_segmentation/backends/
whisperseg.py:419-571, imported from source), on seeded probability tracks with 32 ms frames, no padding and no
maximum-speech split (the two features where that port and silero-vad differ by design).

Unit correspondence (the port counts whole frames, silero-vad samples): a segment is kept by silero-vad when it is
LONGER than min_speech samples, by the port when it has AT LEAST int(min_speech_ms / 32) frames -- hence the port is
run with min_speech_ms + 32; min_silence_ms is a multiple of 32 so both confirm the silence on the same frame.

Run from the repo root inside the build container:  python tests/golden/make_vad_hysteresis_fixtures.py
"""
import importlib
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")

CASES = [dict(threshold=0.5, min_speech_ms=250, min_silence_ms=96), dict(threshold=0.35, min_speech_ms=100, min_silence_ms=160),
         dict(threshold=0.7, min_speech_ms=400, min_silence_ms=320), dict(threshold=0.12, min_speech_ms=60, min_silence_ms=32)]


def tracks(seed: int, count: int):
    rng = np.random.default_rng(seed)
    for _ in range(count):
        n = int(rng.integers(1, 220))
        yield np.clip(np.cumsum(rng.normal(0, 0.25, n)) * 0.3 + 0.5 + rng.normal(0, 0.1, n), 0, 1).astype(np.float32)


def main():
    for name, path in (("whisperjav", "/root/reference/whisperjav"), ("whisperjav.modules", "/root/reference/whisperjav/modules"),
                       ("whisperjav.modules.speech_segmentation", "/root/reference/whisperjav/modules/speech_segmentation"),
                       ("whisperjav.modules.speech_segmentation.backends",
                        "/root/reference/whisperjav/modules/speech_segmentation/backends")):
        m = types.ModuleType(name)
        m.__path__ = [path]
        sys.modules[name] = m
    ws = importlib.import_module("whisperjav.modules.speech_segmentation.backends.whisperseg")
    out = []
    for ci, cfg in enumerate(CASES):
        seg = ws.WhisperSegSpeechSegmenter.__new__(ws.WhisperSegSpeechSegmenter)
        seg.threshold = cfg["threshold"]
        seg.min_speech_duration_ms = cfg["min_speech_ms"] + 32
        seg.min_silence_duration_ms = cfg["min_silence_ms"]
        seg.speech_pad_ms = 0
        seg.max_speech_duration_s = 0
        seg._frame_duration_ms = 32
        rows = []
        for p in tracks(100 + ci, 250):
            res = seg._probs_to_segments(p, len(p) * 512 / 16000)
            rows.append([[s.start_sample, s.end_sample] for s in res])
        out.append({**cfg, "seed": 100 + ci, "count": 250, "segments": rows})
        print(ci, sum(len(r) for r in rows))
    with open(os.path.join(HERE, "reference_vad_hysteresis.json"), "w") as f:
        json.dump(out, f)


if __name__ == "__main__":
    main()
