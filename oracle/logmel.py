"""Oracle: Whisper log-mel front end (TEST INFRASTRUCTURE, never shipped).

Restates, in NumPy, the two upstream feature extractors WhisperJAV's hot path
transits (neither is vendored under /root/reference; call sites:
whisperjav/modules/faster_whisper_pro_asr.py:819 and
whisperjav/modules/whisper_pro_asr.py:433):

  * ``logmel_fw``  -- faster-whisper 1.2.1 ``feature_extractor.FeatureExtractor
    .__call__``: pad 160 zero samples, periodic Hann(400), centred STFT with
    reflect padding, hop 160, drop the last frame, ``|X|^2``, Slaney mel filter
    bank, ``log10(clip(.,1e-10))``, clamp to (clip max - 8), ``(x+4)/4``.
    ``pad_or_trim_fw`` then zero-pads the *features* to 3000 frames.
  * ``logmel_ow``  -- openai-whisper 20250625 ``audio.log_mel_spectrogram`` with
    ``padding=N_SAMPLES``: 30 s of zero *audio* is appended before the STFT,
    so frames past the content carry the clamp floor instead of 0.0.

Validated against ``transformers.WhisperFeatureExtractor`` (an independent
implementation of the same formula) in tests/test_oracle_logmel.py.
"""
from __future__ import annotations

import numpy as np

SAMPLE_RATE = 16000
N_FFT = 400
HOP = 160
N_BINS = N_FFT // 2 + 1          # 201
CHUNK_FRAMES = 3000              # 30 s
CHUNK_SAMPLES = 480000


def hann_periodic(n: int = N_FFT, dtype=np.float64) -> np.ndarray:
    """Periodic Hann window, ``np.hanning(n + 1)[:-1]`` == ``torch.hann_window(n)``."""
    k = np.arange(n, dtype=np.float64)
    return (0.5 - 0.5 * np.cos(2.0 * np.pi * k / n)).astype(dtype)


def mel_filterbank(n_mels: int, n_fft: int = N_FFT, sr: int = SAMPLE_RATE) -> np.ndarray:
    """Slaney-scale, Slaney-normalised triangular filter bank, float32 ``[n_mels, n_fft//2+1]``.

    Same construction as ``librosa.filters.mel`` (which produced openai-whisper's
    ``assets/mel_filters.npz``) and faster-whisper's ``get_mel_filters``.
    """
    bin_hz = np.fft.rfftfreq(n_fft, d=1.0 / sr)
    # mel axis: linear below 1 kHz (200/3 Hz per mel), logarithmic above
    lin_step = 200.0 / 3.0
    knee_hz = 1000.0
    knee_mel = knee_hz / lin_step
    log_step = np.log(6.4) / 27.0
    top_mel = knee_mel + np.log((sr / 2.0) / knee_hz) / log_step
    mel_pts = np.linspace(0.0, top_mel, n_mels + 2)
    hz_pts = np.where(mel_pts >= knee_mel,
                      knee_hz * np.exp(log_step * (mel_pts - knee_mel)),
                      lin_step * mel_pts)
    width = np.diff(hz_pts)
    dist = hz_pts[:, None] - bin_hz[None, :]
    rising = -dist[:-2] / width[:-1, None]
    falling = dist[2:] / width[1:, None]
    tri = np.maximum(0.0, np.minimum(rising, falling))
    tri *= (2.0 / (hz_pts[2:n_mels + 2] - hz_pts[:n_mels]))[:, None]
    return tri.astype(np.float32)


def frame_count(n_samples: int) -> int:
    """Frames the centred STFT yields before the last one is dropped: ``n // hop``."""
    return n_samples // HOP


def stft_power(x: np.ndarray, dtype=np.float32) -> np.ndarray:
    """``|STFT|^2`` of a 1-D signal, centred, reflect-padded, last frame dropped.

    Returns ``[201, len(x)//160]``.  ``dtype`` float32 mirrors upstream (NumPy 2.x
    pocketfft runs natively in float32); float64 is the high-precision yardstick.
    """
    x = np.asarray(x, dtype=dtype)
    if x.shape[0] <= N_FFT // 2:
        raise ValueError("reflect padding needs more than n_fft/2 samples")
    xp = np.pad(x, (N_FFT // 2, N_FFT // 2), mode="reflect")
    n_frames = 1 + (xp.shape[0] - N_FFT) // HOP
    idx = np.arange(N_FFT)[None, :] + HOP * np.arange(n_frames)[:, None]
    frames = xp[idx] * hann_periodic(N_FFT, dtype)[None, :]
    spec = np.fft.rfft(frames, axis=-1)
    if dtype == np.float32:
        spec = spec.astype(np.complex64)
    power = (np.abs(spec) ** 2).astype(dtype)
    return power[:-1].T  # drop last frame -> [201, n_frames-1]


def _log_compress(mel_power: np.ndarray) -> np.ndarray:
    log_spec = np.log10(np.maximum(mel_power, 1e-10))
    log_spec = np.maximum(log_spec, log_spec.max() - 8.0)
    return (log_spec + 4.0) / 4.0


def logmel_fw(audio: np.ndarray, n_mels: int = 128, padding: int = 160,
              dtype=np.float32) -> np.ndarray:
    """faster-whisper semantics. Returns ``[n_mels, (len(audio)+padding)//160]``."""
    audio = np.asarray(audio, dtype=np.float32)
    if padding:
        audio = np.pad(audio, (0, padding))
    power = stft_power(audio, dtype)
    filt = mel_filterbank(n_mels).astype(dtype)
    return _log_compress(filt @ power).astype(dtype)


def pad_or_trim_fw(features: np.ndarray, length: int = CHUNK_FRAMES) -> np.ndarray:
    """faster-whisper ``pad_or_trim``: cut or zero-pad the frame axis to ``length``."""
    n = features.shape[-1]
    if n >= length:
        return features[..., :length]
    out = np.zeros(features.shape[:-1] + (length,), dtype=features.dtype)
    out[..., :n] = features
    return out


def logmel_ow(audio: np.ndarray, n_mels: int = 128, padding: int = CHUNK_SAMPLES,
              dtype=np.float32) -> np.ndarray:
    """openai-whisper semantics (zero *audio* appended before the STFT).

    Returns ``[n_mels, (len(audio)+padding)//160]``; ``whisper.transcribe`` then slices
    ``mel[:, seek:seek+3000]``.
    """
    audio = np.asarray(audio, dtype=np.float32)
    if padding:
        audio = np.pad(audio, (0, padding))
    power = stft_power(audio, dtype)
    filt = mel_filterbank(n_mels).astype(dtype)
    return _log_compress(filt @ power).astype(dtype)


def window_features(audio: np.ndarray, n_mels: int, mode: str, dtype=np.float32) -> np.ndarray:
    """The ``[n_mels, 3000]`` encoder input for one <=30 s clip under either semantics."""
    if mode == "fw":
        return pad_or_trim_fw(logmel_fw(audio, n_mels, dtype=dtype))
    if mode == "ow":
        return logmel_ow(audio, n_mels, dtype=dtype)[:, :CHUNK_FRAMES]
    raise ValueError(f"unknown mel mode {mode!r}")
