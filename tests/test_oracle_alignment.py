"""Pins oracle/alignment.py against the independent implementation importable offline (transformers'
``_median_filter`` / ``_dynamic_time_warping``, which restate whisper/timing.py as well)."""
import numpy as np
import pytest
import torch

from oracle import alignment


def test_median_filter_matches_transformers():
    from transformers.models.whisper.generation_whisper import _median_filter
    rng = np.random.default_rng(0)
    for shape, width in (((3, 17, 40), 7), ((2, 5, 9), 3), ((1, 4, 3), 7), ((2, 3, 64), 5)):
        x = rng.standard_normal(shape).astype(np.float32)
        ref = _median_filter(torch.from_numpy(x), width).numpy()
        assert np.array_equal(alignment.median_filter(x, width), ref)


@pytest.mark.parametrize("n,m,seed", [(1, 1, 0), (5, 40, 1), (23, 150, 2), (40, 7, 3)])
def test_dtw_matches_transformers(n, m, seed):
    from transformers.models.whisper.generation_whisper import _dynamic_time_warping
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((n, m)).astype(np.float32)
    x[rng.integers(0, n, 5), rng.integers(0, m, 5)] = 0.0      # exact ties exercise the tie-breaking order
    ti, fi = alignment.dtw(x)
    rti, rfi = _dynamic_time_warping(x.astype(np.float64))
    assert np.array_equal(ti, rti) and np.array_equal(fi, rfi)
    assert ti[0] == 0 and fi[0] == 0 and ti[-1] == n - 1 and fi[-1] == m - 1
    assert np.all(np.diff(ti) >= 0) and np.all(np.diff(fi) >= 0)


def test_find_alignment_is_monotone_and_consistent():
    from tests import helpers
    from oracle import logmel as olm
    d = helpers.small_dims()
    oracle, _ = helpers.make_oracle(d, seed=21)
    mel = torch.from_numpy(helpers.synth_mel(1, d.n_mels, seed=3))
    with torch.no_grad():
        xa = oracle.encode(mel)
    from oracle import decoding
    lay = decoding.TokenLayout.for_vocab(d.n_vocab)
    text = [11, 500, 7, 7, 1234, 42, 9]
    ti, fi, probs, matrix = alignment.find_alignment(oracle, xa, [lay.sot, lay.sot + 1, lay.sot + 101], lay.no_timestamps,
                                                     text, lay.eot, 1200, [(0, 1), (1, 0), (1, 1)])
    assert matrix.shape == (len(text) + 1, 600)          # text tokens + the <|notimestamps|> row, eot dropped
    assert probs.shape == (len(text),) and np.all((probs >= 0) & (probs <= 1))
    assert ti[-1] == len(text) and fi[-1] == 599 and np.all(np.diff(ti) >= 0) and np.all(np.diff(fi) >= 0)
