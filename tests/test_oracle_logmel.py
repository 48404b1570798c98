"""Pin the log-mel oracle against transformers.WhisperFeatureExtractor (independent implementation)."""
import numpy as np
import pytest

from oracle import logmel as L


@pytest.mark.parametrize("n_mels", [80, 128])
def test_filterbank_equals_hf(n_mels):
    from transformers import WhisperFeatureExtractor
    fe = WhisperFeatureExtractor(feature_size=n_mels)
    ours = L.mel_filterbank(n_mels)
    assert ours.shape == (n_mels, 201) and ours.dtype == np.float32
    assert np.array_equal(ours, fe.mel_filters.T.astype(np.float32))


@pytest.mark.parametrize("n_mels,seconds", [(80, 7.3), (128, 20.0)])
def test_ow_semantics_match_hf(n_mels, seconds):
    from transformers import WhisperFeatureExtractor
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(int(16000 * seconds)) * 0.1).astype(np.float32)
    hf = WhisperFeatureExtractor(feature_size=n_mels)(x, sampling_rate=16000, return_tensors="np")
    ref = hf["input_features"][0]
    got = L.window_features(x, n_mels, "ow")
    assert got.shape == ref.shape == (n_mels, 3000)
    assert np.abs(got - ref).max() < 5e-5


def test_fw_vs_ow_semantics():
    rng = np.random.default_rng(1)
    n = 16000 * 6
    x = (rng.standard_normal(n) * 0.05).astype(np.float32)
    fw = L.logmel_fw(x, 128)
    assert fw.shape == (128, n // 160 + 1)          # faster-whisper: (N + 160) // 160 frames
    win_fw = L.window_features(x, 128, "fw")
    win_ow = L.window_features(x, 128, "ow")
    # identical away from the clip end, different padding values after it
    assert np.array_equal(win_fw[:, :590], win_ow[:, :590])
    assert np.all(win_fw[:, 700:] == 0.0)
    floor = (win_ow.max() * 4.0 - 4.0 - 8.0 + 4.0) / 4.0
    assert np.allclose(win_ow[:, 700:], floor, atol=1e-6)


def test_float32_path_close_to_float64():
    rng = np.random.default_rng(2)
    x = (rng.standard_normal(16000 * 10) * 0.2).astype(np.float32)
    a = L.logmel_fw(x, 128, dtype=np.float32)
    b = L.logmel_fw(x, 128, dtype=np.float64)
    assert np.abs(a - b).max() < 2e-5
