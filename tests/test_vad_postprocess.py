"""Silero-compatible hysteresis state machine: the product's ``vad.regions_from_probs`` and the oracle's
``silero_ref.speech_timestamps`` against fixtures computed by the REFERENCE's own pure-Python port
(tests/golden/make_vad_hysteresis_fixtures.py runs backends/whisperseg.py:_probs_to_segments from source)."""
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_vad_hysteresis.json")


def _tracks(seed, count):
    rng = np.random.default_rng(seed)
    for _ in range(count):
        n = int(rng.integers(1, 220))
        yield np.clip(np.cumsum(rng.normal(0, 0.25, n)) * 0.3 + 0.5 + rng.normal(0, 0.1, n), 0, 1).astype(np.float32)


@pytest.mark.parametrize("case", range(4))
def test_hysteresis_path_equals_reference_port(case):
    from oracle import silero_ref
    from whisperjav_amd import vad
    cfg = json.load(open(GOLD))[case]
    n_segments = 0
    for p, want in zip(_tracks(cfg["seed"], cfg["count"]), cfg["segments"]):
        kw = dict(threshold=cfg["threshold"], min_speech_duration_ms=cfg["min_speech_ms"],
                  min_silence_duration_ms=cfg["min_silence_ms"], speech_pad_ms=0, max_speech_duration_s=float("inf"))
        got = vad.regions_from_probs(p, len(p) * 512, **kw)
        assert [[r["start"], r["end"]] for r in got] == want
        ref = silero_ref.speech_timestamps(p, len(p) * 512, **kw)
        assert [[r["start"], r["end"]] for r in ref] == want
        n_segments += len(want)
    assert n_segments > 150
