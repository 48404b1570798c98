"""The committed golden vectors are reproducible from the oracle (guards both against drift)."""
import os

import numpy as np
import torch

from oracle import decoding, logmel, whisper_ref
from tests import helpers
from whisperjav_amd import synth, weights as pweights

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_logmel_golden():
    g = np.load(os.path.join(GOLDEN, "golden_logmel.npz"))
    audio = synth.speech_like(30.0, seed=1234)
    fw128 = logmel.window_features(audio, 128, "fw")
    assert int(g["frames_fw"]) == 3001
    assert np.array_equal(fw128[:, g["cols"]], g["fw128"])
    assert np.array_equal(logmel.window_features(audio, 80, "fw")[:, g["cols"]], g["fw80"])
    assert np.array_equal(logmel.window_features(audio[: 16000 * 11], 128, "ow")[:, g["cols"]], g["ow128"])
    assert abs(float(fw128.astype(np.float64).sum()) - float(g["fw128_sum"])) < 1e-6 * abs(float(g["fw128_sum"]))


def test_small_model_golden():
    g = np.load(os.path.join(GOLDEN, "golden_small.npz"))
    d = helpers.small_dims()
    assert list(d.as_dict().values()) == g["dims"].tolist()
    w = pweights.synth_weights(d, seed=int(g["seed"]))
    oracle = whisper_ref.WhisperOracle(helpers.oracle_dims(d), w)
    mel = logmel.window_features(synth.speech_like(30.0, seed=1234), 80, "fw")[None]
    cfg = decoding.FilterConfig(suppress_tokens=tuple(int(t) for t in g["suppress"]), max_initial_timestamp_index=50)
    with torch.no_grad():
        enc = oracle.encode(torch.from_numpy(mel))
        res = decoding.greedy_decode(oracle, enc, g["prompt"].tolist(), len(g["tokens"]), cfg)
    assert res.tokens[0] == g["tokens"].tolist()
    assert np.allclose(res.token_logprob[0], g["token_logprob"], atol=1e-4)
    probe = enc[0][g["probe_t"]][:, g["probe_d"]].numpy()
    assert np.allclose(probe, g["enc_probe"], atol=1e-4)
    # timestamp grammar of the decoded sequence: starts with a timestamp <= 1.0 s, stamps never decrease
    tb = decoding.TokenLayout.for_vocab(d.n_vocab).timestamp_begin
    seq = res.tokens[0]
    assert tb <= seq[0] <= tb + 50
    stamps = [t for t in seq if t >= tb]
    assert stamps == sorted(stamps)
