"""Scene detection, CPU side: the oracle's two-pass restatement against fixtures produced by the
REFERENCE's own driver (tests/golden/make_scene_fixtures.py), the tokenizer's upstream behaviours, and the host
tokenizer of whisperjav_amd/scenes.py against the oracle's."""
import json
import os

import numpy as np
import pytest

from oracle import auditok_ref

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_scenes.json")


def _cases():
    with open(GOLD) as f:
        return json.load(f)


@pytest.mark.parametrize("idx", range(11))
def test_two_pass_oracle_equals_reference_driver(idx):
    from whisperjav_amd import synth
    case = _cases()["cases"][idx]
    audio = synth.speech_like(case["seconds"], seed=case["seed"], noisy=case["noisy"])
    cfg = auditok_ref.SceneConfig(**case["cfg"])
    total = len(audio) / 16000
    story = auditok_ref.split(auditok_ref.to_pcm16(audio), 16000, cfg.pass1_min_duration, cfg.pass1_max_duration,
                              min(total * 0.95, cfg.pass1_max_silence), cfg.pass1_energy_threshold)
    assert [list(x) for x in story] == case["story"]
    got = auditok_ref.two_pass_scenes(audio, 16000, cfg)
    assert [[a, b, p] for a, b, p in got] == [[a, b, p] for a, b, p, _ in case["scenes"]]       # exact float equality
    assert [int(b * 16000) - int(a * 16000) for a, b, _ in got] == case["scene_samples"]
    assert all(b - a <= cfg.max_duration + 1e-9 for a, b, _ in got)


def test_config_defaults_match_reference():
    from whisperjav_amd import scenes
    ref = _cases()["config_defaults"]
    mine = scenes.AuditokSceneConfig()
    for k, v in ref.items():
        if k.startswith(("bandpass", "drc", "skip_assist")):
            continue            # assistive processing block: not carried (off by default)
        assert getattr(mine, k) == v, k
    oracle_cfg = auditok_ref.SceneConfig()
    for k in oracle_cfg.__dataclass_fields__:
        assert getattr(oracle_cfg, k) == ref[k], k


def test_tokenizer_upstream_behaviours():
    tok = auditok_ref.tokenize
    v = np.array([0, 1, 1, 1, 0, 0, 1, 1, 0, 0, 0, 1], dtype=bool)
    # silence of 2 tolerated, trailing silence dropped, the lone last frame is shorter than min_length
    assert tok(v, 2, 100, 2) == [(1, 7)]
    assert tok(v, 2, 100, 1) == [(1, 3), (6, 7)]
    # max_length cuts keep going back to back; the remainder is delivered although shorter than min_length
    assert tok(np.ones(11, dtype=bool), 3, 4, 1) == [(0, 3), (4, 7), (8, 10)]
    assert tok(np.ones(9, dtype=bool), 3, 4, 1) == [(0, 3), (4, 7), (8, 8)]
    assert tok(np.zeros(7, dtype=bool), 1, 4, 1) == []
    with pytest.raises(ValueError):
        auditok_ref.split(np.zeros(16000, np.int16), 16000, 1.0, 0.5, 0.1, 30)
    with pytest.raises(ValueError):
        auditok_ref.split(np.zeros(16000, np.int16), 16000, 0.1, 0.5, 0.5, 30)


def test_host_tokenizer_equals_oracle_on_random_flags():
    from whisperjav_amd import scenes
    rng = np.random.default_rng(0)
    for _ in range(4000):
        n = int(rng.integers(1, 80))
        v = rng.random(n) < rng.choice([0.2, 0.5, 0.8])
        mn = int(rng.integers(1, 6))
        mx = int(rng.integers(mn, 14))
        ms = int(rng.integers(0, mx))
        assert scenes.tokenize_flags(v, mn, mx, ms) == auditok_ref.tokenize(v, mn, mx, ms, True)


def test_energy_formula_and_quantisation():
    pcm = auditok_ref.to_pcm16(np.array([0.5, -0.5, 1.0, -1.0, 0.99999, 3.0517578125e-05], dtype=np.float32))
    assert pcm.tolist() == [16383, -16383, 32767, -32767, 32766, 0]
    e = auditok_ref.frame_energies(np.array([100] * 800 + [0] * 800 + [7] * 3, dtype=np.int16), 800)
    assert e[0] == pytest.approx(40.0) and e[1] == pytest.approx(-200.0) and e[2] == pytest.approx(20 * np.log10(7))
