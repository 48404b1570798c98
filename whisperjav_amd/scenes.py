"""Scene detection on the MI355X: the reference's two-pass energy gate with the frame energies computed on the device.

Mirrors ``AuditokSceneDetector`` (/root/reference/whisperjav/modules/scene_detection_backends/auditok_backend.py:
config :35-93, ``detect_scenes`` :229-322, pass 1 :367-399, story lines :401-519, pass 2 :521-567) behind the
``SceneDetector`` protocol of ``scene_detection_backends/base.py`` -- register it in
``scene_detection_backends/factory.py:24`` as ``"auditok-hip": "whisperjav_amd.scenes.HipAuditokSceneDetector"``.

What runs where.  ``auditok.split`` is a pure-Python generator over 50 ms frames (the serial prefix before any GPU
work).  Here the clip goes to HBM once, ``wj_frame_sumsq`` returns the exact integer sum of squares of
the PCM16-quantised samples of every analysis frame (pass 1: one launch for the file; pass 2: one launch for all
oversized story lines), and the host evaluates auditok's energy formula in float64 from those integers (bit-identical
to auditok's own numbers) and runs the tokenizer on the boolean frame flags.  ``split_clip`` hands the scenes over in
memory; ``detect_scenes`` writes the PCM16 WAVs like the reference (PCM16 quantisation is part of today's numerics).

There is no CPU fallback: without the HIP library / device the constructor raises.
"""
from __future__ import annotations

import math
import time
import wave
from dataclasses import dataclass, field
from pathlib import Path
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import hipbind
from .pipeline import pcm16_encode

ANALYSIS_WINDOW = 0.05          # auditok DEFAULT_ANALYSIS_WINDOW
_EPS = 1e-10


# ---- data contracts: the reference's own classes when importable, else field-for-field mirrors -----------------
try:  # pragma: no cover - depends on the host installation
    from whisperjav.modules.scene_detection_backends.base import (SceneDetectionError, SceneDetectionResult,  # type: ignore
                                                                   SceneInfo)
except Exception:  # noqa: BLE001
    class SceneDetectionError(Exception):
        """Detection itself failed (distinct from an empty result)."""

    @dataclass
    class SceneInfo:                                    # base.py:38-98
        start_sec: float
        end_sec: float
        scene_path: Optional[Path] = None
        detection_pass: int = 0
        metadata: Dict[str, Any] = field(default_factory=dict)

        @property
        def duration_sec(self) -> float:
            return self.end_sec - self.start_sec

        def to_dict(self) -> Dict[str, Any]:
            out = {"start_time_seconds": round(self.start_sec, 3), "end_time_seconds": round(self.end_sec, 3),
                   "duration_seconds": round(self.duration_sec, 3), "detection_pass": self.detection_pass}
            if self.scene_path is not None:
                out["filename"] = self.scene_path.name
                out["path"] = str(self.scene_path)
            return out

        def to_legacy_tuple(self) -> Tuple[Path, float, float, float]:
            if self.scene_path is None:
                raise ValueError("scene_path is not set")
            return (self.scene_path, self.start_sec, self.end_sec, self.duration_sec)

    @dataclass
    class SceneDetectionResult:                         # base.py:100-180
        scenes: List[SceneInfo]
        method: str
        audio_duration_sec: float = 0.0
        parameters: Dict[str, Any] = field(default_factory=dict)
        processing_time_sec: float = 0.0
        coarse_boundaries: Optional[List[Dict[str, Any]]] = None
        vad_segments: Optional[List[Dict[str, Any]]] = None

        @property
        def num_scenes(self) -> int:
            return len(self.scenes)

        @property
        def total_scene_duration_sec(self) -> float:
            return sum(s.duration_sec for s in self.scenes)

        def to_legacy_tuples(self):
            return [s.to_legacy_tuple() for s in self.scenes]


@dataclass
class AuditokSceneConfig:
    """Field for field ``AuditokSceneConfig`` (auditok_backend.py:35-93) minus the assistive-processing block
    (bandpass + DRC through pydub/scipy: off by default and outside the hot path)."""
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
    assist_processing: bool = False
    brute_force_fallback: bool = True
    brute_force_chunk_s: Optional[float] = None
    pad_edges_s: float = 0.0
    verbose_summary: bool = True
    force_mono: bool = True

    def __post_init__(self):
        if self.pass2_max_duration is None:
            self.pass2_max_duration = max(self.max_duration - 1.0, self.min_duration)
        if self.brute_force_chunk_s is None:
            self.brute_force_chunk_s = self.max_duration


def _nb_windows(duration: float, round_fn, eps: float = 0.0) -> int:
    return 0 if duration == 0 else int(round_fn(duration / ANALYSIS_WINDOW + eps))


def tokenize_flags(valid: np.ndarray, min_length: int, max_length: int, max_silence: int) -> List[Tuple[int, int]]:
    """auditok ``StreamTokenizer`` (mode DROP_TRAILING_SILENCE, ``init_min`` 0) over frame validity flags, frame by
    frame with upstream's own state variables (including its quirk that the "contiguous with a truncated token"
    exemption from ``min_length`` survives an all-silent remainder).  Returns inclusive frame ranges."""
    SILENCE, POSSIBLE_SILENCE, NOISE = 0, 1, 3
    tokens: List[Tuple[int, int]] = []
    state, n_data, n_sil, start, contiguous = SILENCE, 0, 0, 0, False

    def deliver(cur: int, truncated: bool) -> None:
        nonlocal n_data, start, contiguous
        if not truncated and n_sil > 0:
            n_data -= n_sil                              # drop_trailing_silence
        if n_data >= min_length or (n_data > 0 and contiguous):
            tokens.append((start, start + n_data - 1))
            if truncated:
                start = cur + 1
            contiguous = truncated
        else:
            contiguous = False
        n_data = 0

    flags = np.asarray(valid, dtype=bool).tolist()
    for cur, ok in enumerate(flags):
        if state == SILENCE:
            if ok:
                n_sil, start, n_data, state = 0, cur, n_data + 1, NOISE
                if n_data >= max_length:
                    deliver(cur, True)
        elif state == NOISE:
            if ok:
                n_data += 1
                if n_data >= max_length:
                    deliver(cur, True)
            elif max_silence <= 0:
                state = SILENCE
                deliver(cur, False)
            else:
                n_sil, n_data, state = 1, n_data + 1, POSSIBLE_SILENCE
                if n_data == max_length:
                    deliver(cur, True)
        else:
            if ok:
                n_data, n_sil, state = n_data + 1, 0, NOISE
                if n_data >= max_length:
                    deliver(cur, True)
            elif n_sil >= max_silence:
                state = SILENCE
                if n_sil < n_data:
                    deliver(cur, False)
                else:
                    n_data, n_sil = 0, 0
            else:
                n_data += 1
                n_sil += 1
                if n_data >= max_length:
                    deliver(cur, True)
    if state in (NOISE, POSSIBLE_SILENCE) and n_data > 0 and n_data > n_sil:
        deliver(len(flags), False)
    return tokens


class HipAuditokSceneDetector:
    """Drop-in for ``AuditokSceneDetector``: same constructor keywords (``config=`` or legacy ``*_s`` aliases), same
    ``detect_scenes(audio_path, output_dir, media_basename) -> SceneDetectionResult`` and ``cleanup()``."""

    name = "auditok-hip"
    display_name = "Auditok energy gate (MI355X)"

    _ALIASES = {"max_duration_s": "max_duration", "min_duration_s": "min_duration", "pass1_max_silence_s": "pass1_max_silence",
                "pass2_max_silence_s": "pass2_max_silence", "pass1_min_duration_s": "pass1_min_duration",
                "pass2_min_duration_s": "pass2_min_duration", "pass1_max_duration_s": "pass1_max_duration",
                "pass2_max_duration_s": "pass2_max_duration", "brute_force_chunk_s": "brute_force_chunk_s",
                "pad_edges_s": "pad_edges_s"}

    def __init__(self, config: Optional[AuditokSceneConfig] = None, device: int = 0, **kwargs):
        if config is None:
            fields = set(AuditokSceneConfig.__dataclass_fields__)
            params = {}
            for k, v in kwargs.items():
                k = self._ALIASES.get(k, k)
                if k in fields and v is not None:
                    params[k] = v
            config = AuditokSceneConfig(**params)
        if config.assist_processing:
            raise ValueError("assist_processing (bandpass + DRC) is not available on the HIP path")
        self._config = config
        self._device = int(device)
        self._ctx = hipbind.context(self._device)      # raises without the library / a gfx950 device
        self._lib = hipbind.lib()
        self._last_result = None

    # ---- device front end --------------------------------------------------------------------------------
    def _frame_flags(self, pcm_dev, n_total: int, regions: Sequence[Tuple[int, int]], block: int,
                     thresholds: Sequence[float]) -> List[np.ndarray]:
        """Per region (offset, length in samples): boolean validity of its analysis frames."""
        import ctypes as C
        offs, lens, counts = [], [], []
        for off, n in regions:
            k = (n + block - 1) // block
            counts.append(k)
            starts = off + block * np.arange(k, dtype=np.int64)
            offs.append(starts)
            lens.append(np.minimum(block, off + n - starts).astype(np.int32))
        if not offs or sum(counts) == 0:
            return [np.zeros(0, dtype=bool) for _ in regions]
        f_off = np.ascontiguousarray(np.concatenate(offs))
        f_len = np.ascontiguousarray(np.concatenate(lens))
        sums = np.empty(len(f_off), dtype=np.int64)
        hipbind.check(self._lib.wj_frame_sumsq(self._ctx.handle, pcm_dev.data_ptr(), int(n_total),
                                               f_off.ctypes.data_as(C.POINTER(C.c_int64)),
                                               f_len.ctypes.data_as(C.POINTER(C.c_int32)), len(f_off),
                                               sums.ctypes.data_as(C.POINTER(C.c_int64)), None), "wj_frame_sumsq")
        # auditok: 20 * log10(clip(sqrt(mean(x ** 2)), 1e-10)) in float64; the integer sums are exact in float64
        energy = 20.0 * np.log10(np.maximum(np.sqrt(sums.astype(np.float64) / f_len.astype(np.float64)), _EPS))
        out, p = [], 0
        for k, thr in zip(counts, thresholds):
            out.append(energy[p:p + k] >= thr)
            p += k
        return out

    @staticmethod
    def _split_params(min_dur: float, max_dur: float, max_silence: float) -> Tuple[int, int, int]:
        min_length = _nb_windows(min_dur, math.ceil)
        max_length = _nb_windows(max_dur, math.floor, _EPS)
        max_cs = _nb_windows(max_silence, math.floor, _EPS)
        if min_length > max_length:
            raise ValueError("'min_dur' is higher than 'max_dur' in analysis windows")
        if max_cs >= max_length:
            raise ValueError("'max_silence' is higher than or equal to 'max_dur' in analysis windows")
        return min_length, max_length, max_cs

    @staticmethod
    def _regions(tokens, block: int, sr: int, n: int) -> List[Tuple[float, float]]:
        block_dur = block / sr
        out = []
        for f0, f1 in tokens:
            start = f0 * block_dur
            out.append((start, start + (min((f1 + 1) * block, n) - f0 * block) / sr))
        return out

    # ---- the two passes ------------------------------------------------------------------------------------
    def split_clip(self, audio: np.ndarray, sample_rate: int = 16000) -> Tuple[List[Tuple[float, float, int, Dict[str, Any]]], List[Tuple[float, float]]]:
        """In-memory scene detection of one clip (float32 mono).  Returns (scenes, story_lines) with scenes =
        [(start_s, end_s, detection_pass, metadata)] in the order the reference produces them."""
        import torch
        cfg = self._config
        audio = np.ascontiguousarray(audio, dtype=np.float32).reshape(-1)
        n, sr = len(audio), int(sample_rate)
        total = n / sr
        block = int(sr * ANALYSIS_WINDOW)
        if n == 0:
            return [], []
        pcm_dev = torch.from_numpy(audio).to(f"cuda:{self._device}")
        p1 = self._split_params(cfg.pass1_min_duration, cfg.pass1_max_duration, min(total * 0.95, cfg.pass1_max_silence))
        flags = self._frame_flags(pcm_dev, n, [(0, n)], block, [cfg.pass1_energy_threshold])[0]
        story = self._regions(tokenize_flags(flags, *p1), block, sr, n)
        # pass 2 for every oversized story line, one launch
        big = [(i, r) for i, r in enumerate(story) if not (cfg.min_duration <= r[1] - r[0] <= cfg.max_duration)]
        sub: Dict[int, List[Tuple[float, float]]] = {}
        if big:
            regs = [(int(r[0] * sr), int(r[1] * sr) - int(r[0] * sr)) for _, r in big]
            flags2 = self._frame_flags(pcm_dev, n, regs, block, [cfg.pass2_energy_threshold] * len(regs))
            for (i, r), (off, ln), fl in zip(big, regs, flags2):
                p2 = self._split_params(cfg.pass2_min_duration, cfg.pass2_max_duration,
                                        min((r[1] - r[0]) * 0.95, cfg.pass2_max_silence))
                sub[i] = self._regions(tokenize_flags(fl, *p2), block, sr, ln)

        def clamp(s: float, e: float) -> Tuple[float, float]:
            s2, e2 = max(0.0, s - cfg.pad_edges_s), min(total, e + cfg.pad_edges_s)
            return s2, max(e2, s2)

        scenes: List[Tuple[float, float, int, Dict[str, Any]]] = []
        for i, (r0, r1) in enumerate(story):
            if i not in sub:
                scenes.append((*clamp(r0, r1), 1, {}))
                continue
            if sub[i]:
                for s0, s1 in sub[i]:
                    a, b = r0 + s0, r0 + s1
                    if b - a < cfg.min_duration:
                        continue
                    scenes.append((*clamp(a, b), 2, {}))
            elif cfg.brute_force_fallback:
                dur = r1 - r0
                for k in range(int(np.ceil(dur / max(cfg.brute_force_chunk_s, cfg.min_duration)))):
                    a = r0 + k * cfg.brute_force_chunk_s
                    b = min(r0 + (k + 1) * cfg.brute_force_chunk_s, r1)
                    if b - a < cfg.min_duration:
                        continue
                    scenes.append((*clamp(a, b), 2, {"split_method": "brute_force"}))
        return scenes, story

    def detect_scenes(self, audio_path: Path, output_dir: Path, media_basename: str, **kwargs) -> "SceneDetectionResult":
        t0 = time.time()
        try:
            from .asr import read_audio
            audio, sr = read_audio(Path(audio_path))
        except Exception as e:  # noqa: BLE001
            raise SceneDetectionError(f"Failed to load audio file {audio_path}: {e}") from e
        total = len(audio) / sr if sr else 0.0
        found, story = self.split_clip(audio, sr)
        coarse = [{"scene_index": i, "start_time_seconds": round(a, 3), "end_time_seconds": round(b, 3),
                   "duration_seconds": round(b - a, 3)} for i, (a, b) in enumerate(story)]
        scenes: List[SceneInfo] = []
        if found:
            output_dir = Path(output_dir)
            output_dir.mkdir(parents=True, exist_ok=True)
        for idx, (s, e, p, meta) in enumerate(found):
            chunk = audio[int(s * sr): int(e * sr)]
            if len(chunk) == 0:
                raise ValueError(f"Empty audio data for scene {idx}")
            path = output_dir / f"{media_basename}_scene_{idx:04d}.wav"
            with wave.open(str(path), "wb") as wf:      # PCM_16, as save_scene_wav (utils.py:107-150)
                wf.setnchannels(1)
                wf.setsampwidth(2)
                wf.setframerate(sr)
                wf.writeframes(pcm16_encode(chunk).astype("<i2").tobytes())       # libsndfile's float -> PCM_16 conversion
            scenes.append(SceneInfo(start_sec=s, end_sec=e, scene_path=path, detection_pass=p, metadata=dict(meta)))
        self._last_result = SceneDetectionResult(scenes=scenes, method=self.name, audio_duration_sec=total,
                                                 parameters=dict(self._config.__dict__), processing_time_sec=time.time() - t0,
                                                 coarse_boundaries=coarse)
        return self._last_result

    def cleanup(self) -> None:
        self._last_result = None
