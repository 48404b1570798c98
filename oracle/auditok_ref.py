"""Oracle: energy-gate scene detection on the CPU (TEST INFRASTRUCTURE, never shipped).

Restates (a) ``auditok.split`` of auditok 0.3.0 (``uv.lock:249-251``; not installed here and not vendored in the
reference) -- 50 ms analysis windows, energy = 20 log10(max(sqrt(mean(x^2)), 1e-10)) of the PCM16 samples, and the
``StreamTokenizer`` state machine with ``drop_trailing_silence`` -- and (b) the two-pass driver that calls it,
``AuditokSceneDetector._detect_pass1 / _process_story_lines / _detect_pass2``
(/root/reference/whisperjav/modules/scene_detection_backends/auditok_backend.py:367-567, brute-force fallback
``utils.py:153-200``).  The tokenizer is written frame by frame exactly like the upstream class (states SILENCE,
POSSIBLE_SILENCE, NOISE; ``init_min = 0`` so POSSIBLE_NOISE is unreachable) so that it can be read against it.

PARITY UNPINNED: no auditok wheel offline and the reference's tests pin no scene boundary; (b) is checked against the
reference source by reading, (a) against auditok's published algorithm.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import List, Optional, Tuple

import numpy as np

ANALYSIS_WINDOW = 0.05
_EPS = 1e-10


def to_pcm16(audio: np.ndarray) -> np.ndarray:
    """``(audio_data * 32767).astype(np.int16)`` (auditok_backend.py:385): float32 product, truncation toward zero."""
    return (np.asarray(audio, dtype=np.float32) * 32767).astype(np.int16)


def frame_energies(pcm: np.ndarray, block: int) -> np.ndarray:
    """Energy in dB of consecutive ``block``-sample frames (the last one may be short)."""
    n = len(pcm)
    out = np.empty((n + block - 1) // block, dtype=np.float64)
    x = pcm.astype(np.float64)
    for i in range(len(out)):
        f = x[i * block:(i + 1) * block]
        out[i] = 20.0 * np.log10(max(np.sqrt(np.mean(f ** 2)), _EPS))
    return out


def _nb_windows(duration: float, window: float, round_fn, eps: float = 0.0) -> int:
    if duration == 0:
        return 0
    return int(round_fn(duration / window + eps))


def tokenize(valid: np.ndarray, min_length: int, max_length: int, max_silence: int, drop_trailing_silence: bool = True,
             strict_min_length: bool = False) -> List[Tuple[int, int]]:
    """StreamTokenizer over per-frame validity flags -> [(start_frame, end_frame inclusive)]."""
    SILENCE, POSSIBLE_SILENCE, NOISE = 0, 1, 3
    state, data_len, silence_len, start_frame, contiguous = SILENCE, 0, 0, 0, False
    tokens: List[Tuple[int, int]] = []

    def end_of_detection(current: int, truncated: bool = False) -> None:
        nonlocal data_len, silence_len, start_frame, contiguous
        if not truncated and drop_trailing_silence and silence_len > 0:
            data_len -= silence_len            # data[0:-silence_len]
        if data_len >= min_length or (data_len > 0 and not strict_min_length and contiguous):
            tokens.append((start_frame, start_frame + data_len - 1))
            data_len = 0
            if truncated:
                start_frame = current + 1
                contiguous = True
            else:
                contiguous = False
            return
        contiguous = False
        data_len = 0

    for cur, ok in enumerate(valid):
        if state == SILENCE:
            if ok:
                silence_len = 0
                start_frame = cur
                data_len += 1
                state = NOISE                  # init_min == 0
                if data_len >= max_length:
                    end_of_detection(cur, True)
        elif state == NOISE:
            if ok:
                data_len += 1
                if data_len >= max_length:
                    end_of_detection(cur, True)
            elif max_silence <= 0:
                state = SILENCE
                end_of_detection(cur)
            else:
                silence_len = 1
                data_len += 1
                state = POSSIBLE_SILENCE
                if data_len == max_length:
                    end_of_detection(cur, True)
        else:                                  # POSSIBLE_SILENCE
            if ok:
                data_len += 1
                silence_len = 0
                state = NOISE
                if data_len >= max_length:
                    end_of_detection(cur, True)
            elif silence_len >= max_silence:
                state = SILENCE
                if silence_len < data_len:
                    end_of_detection(cur)
                else:
                    data_len = 0
                    silence_len = 0
            else:
                data_len += 1
                silence_len += 1
                if data_len >= max_length:
                    end_of_detection(cur, True)
    if state in (NOISE, POSSIBLE_SILENCE) and data_len > 0 and data_len > silence_len:
        end_of_detection(len(valid))
    return tokens


def split(pcm: np.ndarray, sr: int, min_dur: float, max_dur: float, max_silence: float, energy_threshold: float,
          drop_trailing_silence: bool = True) -> List[Tuple[float, float]]:
    """``auditok.split(bytes, sampling_rate=sr, channels=1, sample_width=2, ...)`` -> [(start_s, end_s)]."""
    block = int(sr * ANALYSIS_WINDOW)
    block_dur = block / sr
    min_length = _nb_windows(min_dur, ANALYSIS_WINDOW, math.ceil)
    max_length = _nb_windows(max_dur, ANALYSIS_WINDOW, math.floor, _EPS)
    max_cs = _nb_windows(max_silence, ANALYSIS_WINDOW, math.floor, _EPS)
    if min_length > max_length:
        raise ValueError("'min_dur' is higher than 'max_dur' in analysis windows")
    if max_cs >= max_length:
        raise ValueError("'max_silence' is higher than or equal to 'max_dur' in analysis windows")
    energies = frame_energies(pcm, block)
    out = []
    for f0, f1 in tokenize(energies >= energy_threshold, min_length, max_length, max_cs, drop_trailing_silence):
        n_samples = min((f1 + 1) * block, len(pcm)) - f0 * block
        start = f0 * block_dur
        out.append((start, start + n_samples / sr))
    return out


@dataclass
class SceneConfig:
    """Defaults of ``AuditokSceneConfig`` (auditok_backend.py:35-93)."""
    max_duration: float = 29.0
    min_duration: float = 0.2
    pass1_min_duration: float = 0.3
    pass1_max_duration: float = 2700.0
    pass1_max_silence: float = 1.8
    pass1_energy_threshold: int = 32
    pass2_min_duration: float = 0.3
    pass2_max_duration: Optional[float] = None
    pass2_max_silence: float = 0.94
    pass2_energy_threshold: int = 38
    brute_force_fallback: bool = True
    brute_force_chunk_s: Optional[float] = None
    pad_edges_s: float = 0.0

    def __post_init__(self):
        if self.pass2_max_duration is None:
            self.pass2_max_duration = max(self.max_duration - 1.0, self.min_duration)
        if self.brute_force_chunk_s is None:
            self.brute_force_chunk_s = self.max_duration


def two_pass_scenes(audio: np.ndarray, sr: int, cfg: Optional[SceneConfig] = None) -> List[Tuple[float, float, int]]:
    """The reference's two-pass strategy without the WAV writing: [(start_s, end_s, detection_pass)]."""
    cfg = cfg or SceneConfig()
    total = len(audio) / sr
    story = split(to_pcm16(audio), sr, cfg.pass1_min_duration, cfg.pass1_max_duration,
                  min(total * 0.95, cfg.pass1_max_silence), cfg.pass1_energy_threshold)

    def clamp(s: float, e: float) -> Tuple[float, float]:
        s2, e2 = max(0.0, s - cfg.pad_edges_s), min(total, e + cfg.pad_edges_s)
        return s2, max(e2, s2)

    scenes: List[Tuple[float, float, int]] = []
    for r0, r1 in story:
        dur = r1 - r0
        if cfg.min_duration <= dur <= cfg.max_duration:
            scenes.append((*clamp(r0, r1), 1))
            continue
        region = audio[int(r0 * sr): int(r1 * sr)]
        subs = split(to_pcm16(region), sr, cfg.pass2_min_duration, cfg.pass2_max_duration,
                     min(dur * 0.95, cfg.pass2_max_silence), cfg.pass2_energy_threshold)
        if subs:
            for s0, s1 in subs:
                a, b = r0 + s0, r0 + s1
                if b - a < cfg.min_duration:
                    continue
                scenes.append((*clamp(a, b), 2))
        elif cfg.brute_force_fallback:
            n = int(np.ceil(dur / max(cfg.brute_force_chunk_s, cfg.min_duration)))
            for i in range(n):
                a = r0 + i * cfg.brute_force_chunk_s
                b = min(r0 + (i + 1) * cfg.brute_force_chunk_s, r1)
                if b - a < cfg.min_duration:
                    continue
                scenes.append((*clamp(a, b), 2))
    return scenes
