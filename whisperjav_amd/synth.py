"""Seeded synthetic "Japanese-speech-shaped" 16 kHz audio.

There is no network and no media in the build / benchmark environment, so every measurement and
parity test runs on audio generated here: utterances are a jittered harmonic source shaped by three
formant resonances and amplitude-modulated at the mora rate, separated by log-normal silences, over
a pink noise floor.  The statistics follow the reference's own VAD tuning rationale ("majority of
JA subs < 3 s with ~800 ms gaps", /root/reference/whisperjav/config/components/vad/silero.py:90-93).
"""
from __future__ import annotations

import numpy as np

SR = 16000


def _pink(rng: np.random.Generator, n: int) -> np.ndarray:
    white = rng.standard_normal(n)
    spec = np.fft.rfft(white)
    f = np.arange(spec.shape[0], dtype=np.float64)
    f[0] = 1.0
    pink = np.fft.irfft(spec / np.sqrt(f), n)
    return pink / (np.sqrt(np.mean(pink ** 2)) + 1e-12)


def _utterance(rng: np.random.Generator, dur_s: float) -> np.ndarray:
    n = int(dur_s * SR)
    t = np.arange(n) / SR
    f0 = np.exp(rng.uniform(np.log(110.0), np.log(280.0)))
    drift = 1.0 + 0.03 * np.sin(2 * np.pi * rng.uniform(2.0, 5.0) * t + rng.uniform(0, 2 * np.pi))
    phase = 2 * np.pi * np.cumsum(f0 * drift) / SR
    formants = (rng.uniform(300, 800), rng.uniform(900, 2400), rng.uniform(2500, 3200))
    widths = (90.0, 140.0, 220.0)
    sig = np.zeros(n)
    for h in range(1, 31):
        fh = f0 * h
        if fh > 0.45 * SR:
            break
        gain = sum(1.0 / (1.0 + ((fh - fc) / bw) ** 2) for fc, bw in zip(formants, widths)) / h ** 0.5
        sig += gain * np.sin(h * phase + rng.uniform(0, 2 * np.pi))
    mora = 7.5
    env_phase = (t * mora) % 1.0
    env = np.where(env_phase < 0.7, 0.5 - 0.5 * np.cos(2 * np.pi * env_phase / 0.7), 0.0)
    ramp = np.minimum(1.0, np.minimum(t, t[::-1]) / 0.02)
    sig = sig * env * ramp
    rms = np.sqrt(np.mean(sig ** 2)) + 1e-12
    return sig * (10 ** (-18 / 20) / rms)


def speech_like(duration_s: float, seed: int = 1234, noisy: bool = False, floor_db: float = -45.0, snr_db: float = 10.0) -> np.ndarray:
    """float32 mono audio in [-1, 1] of exactly ``duration_s`` seconds.  ``floor_db``: level of the pink noise floor (dBFS;
    -45 = a noisy room: 45 dB in auditok's energy scale, ABOVE the reference's 32 / 38 dB scene gates; -66 = a studio
    recording on which those gates open and close); ``noisy``: plus hum and pink noise ``snr_db`` below the speech level."""
    rng = np.random.default_rng(seed)
    n = int(round(duration_s * SR))
    out = np.zeros(n, dtype=np.float64)
    pos = int(rng.uniform(0.1, 0.6) * SR)
    next_chapter = 90.0
    while pos < n:
        dur = float(np.clip(rng.lognormal(np.log(1.8), 0.5), 0.3, 5.0))
        utt = _utterance(rng, dur)
        end = min(n, pos + utt.shape[0])
        out[pos:end] += utt[: end - pos]
        gap = float(np.clip(rng.lognormal(np.log(0.8), 0.6), 0.15, 6.0))
        if end / SR > next_chapter:
            gap = float(rng.uniform(2.5, 6.0))
            next_chapter += 90.0
        pos = end + int(gap * SR)
    out += _pink(rng, n) * 10 ** (floor_db / 20)
    if noisy:
        speech_rms = 10 ** (-18 / 20)
        noise = _pink(rng, n) + 0.5 * np.sin(2 * np.pi * 50 * np.arange(n) / SR) + 0.3 * np.sin(
            2 * np.pi * 100 * np.arange(n) / SR)
        noise *= speech_rms * 10 ** (-snr_db / 20) / (np.sqrt(np.mean(noise ** 2)) + 1e-12)
        out += noise
    return np.clip(out, -1.0, 1.0).astype(np.float32)


def speech_like_long(duration_s: float, seed: int = 1234, noisy: bool = False, chunk_s: float = 600.0,
                     workers: int = 0, floor_db: float = -45.0, snr_db: float = 10.0) -> np.ndarray:
    """Long recordings (the 120-minute benchmark file) as independent ``chunk_s`` pieces with seeds ``seed + 7919 i``,
    generated in parallel (a single 115 M-sample FFT for the pink floor takes minutes); the result depends only on
    (duration_s, seed, noisy, chunk_s).  Same per-chunk statistics as ``speech_like``.

    The workers are plain ``python -m whisperjav_amd.synth`` SUBPROCESSES writing .npy files -- not ``multiprocessing``:
    a forked child of a process that has initialised the HIP runtime (or merely loaded libwjhip.so) can hang at exit
    (measured: a 22-minute stall of bench.py on the GPU box), and spawn pools re-import ``__main__``."""
    n_chunks = max(1, int(np.ceil(duration_s / chunk_s)))
    jobs = [(min(chunk_s, duration_s - i * chunk_s), seed + 7919 * i) for i in range(n_chunks)]
    if n_chunks == 1:
        return speech_like(duration_s, seed=seed, noisy=noisy, floor_db=floor_db, snr_db=snr_db)
    import os
    import subprocess
    import sys
    import tempfile
    workers = workers or min(n_chunks, max(1, (os.cpu_count() or 2) - 1), 16)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""), OMP_NUM_THREADS="1")
    parts = [None] * n_chunks
    with tempfile.TemporaryDirectory(prefix="wj_synth_") as tmp:
        pending = list(range(n_chunks))
        running = {}
        while pending or running:
            while pending and len(running) < workers:
                i = pending.pop(0)
                out = os.path.join(tmp, f"{i}.npy")
                cmd = [sys.executable, "-m", "whisperjav_amd.synth", out, repr(jobs[i][0]), str(jobs[i][1]), str(int(noisy)),
                       repr(float(floor_db)), repr(float(snr_db))]
                running[i] = (subprocess.Popen(cmd, env=env, cwd=root), out)
            for i, (proc, out) in list(running.items()):
                rc = proc.wait() if len(running) >= workers or not pending else proc.poll()
                if rc is None:
                    continue
                if rc != 0:
                    raise RuntimeError(f"synthetic audio worker {i} failed (exit code {rc})")
                parts[i] = np.load(out)
                del running[i]
    return np.concatenate(parts)


if __name__ == "__main__":      # worker: python -m whisperjav_amd.synth OUT.npy DURATION SEED NOISY [FLOOR_DB SNR_DB]
    import sys
    extra = dict(floor_db=float(sys.argv[5]), snr_db=float(sys.argv[6])) if len(sys.argv) > 6 else {}
    np.save(sys.argv[1], speech_like(float(sys.argv[2]), seed=int(sys.argv[3]), noisy=bool(int(sys.argv[4])), **extra))
