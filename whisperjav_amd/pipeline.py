"""Pipeline-level seam: the scene loop of ``balanced`` mode with the work of ALL scenes pooled.

The reference's ``BalancedPipeline.process`` (/root/reference/whisperjav/pipelines/balanced_pipeline.py:281-514)
detects the scenes, then calls ``asr.transcribe_to_srt(scene_path, scene_srt_path, task=...)`` scene by scene
(:476-486) and stitches the per-scene SRT files (``SRTStitcher.stitch``, modules/srt_stitching.py:18-84).  Scene by
scene an MI355X sees ~5 windows per launch; pooled it sees hundreds (45x real-time at batch 1, ~1900x at 384).

Two forms of the same procedure live here:

* ``hip_balanced_pipeline_class()`` -> ``HipBalancedPipeline``: a subclass of the REFERENCE's own ``BalancedPipeline``
  (audio extraction, speech enhancement, stitching, post-processing, metadata, progress reporting all inherited
  unchanged).  It swaps the ASR module for ``asr.HipFasterWhisperProASR`` and wraps the scene detector so that the
  scene list is announced to the ASR module (``prime_scenes``) the moment it exists; the inherited loop then still
  calls ``transcribe_to_srt`` per scene and still finds one SRT per scene, but the first call transcribes the whole
  list in one pooled pass.  Registration: INTEGRATION.md section 2d (``main.py`` dispatch + ``PIPELINE_CLASSES``).
* ``hip_fidelity_pipeline_class()`` -> ``HipFidelityPipeline``: the same for ``FidelityPipeline`` / ``WhisperProASR``.
* ``RecordingTranscriber``: the same steps 2-4 + stitch for one in-memory recording without the ``whisperjav``
  package (bench.py, ``sharded_transcribe``, the GPU tests): scenes -> PCM16 round trip (the reference writes the
  scenes as PCM_16 WAVs, utils.py:107-150, and reads them back; that quantisation is part of its numerics) -> pooled
  ``transcribe_scenes`` -> scene-offset merge in scene order.
"""
from __future__ import annotations

import time
from pathlib import Path
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np

SR = 16000


def pcm16_encode(chunk: np.ndarray) -> np.ndarray:
    """The int16 samples ``soundfile.write(path, chunk, sr, subtype="PCM_16")`` stores (``save_scene_wav``,
    /root/reference/whisperjav/modules/scene_detection_backends/utils.py:140).

    libsndfile (un-vendored; the routine is ``src/pcm.c``: ``f2s_array`` for float32 input, ``d2s_array`` for float64,
    reached through ``sf_writef_float`` / ``sf_writef_double``) converts with ``normfact = 1.0 * 0x7FFF`` -- NOT 0x8000 --
    and ``dest[i] = lrintf(src[i] * normfact)``: the product is formed in the INPUT's precision, ``lrint`` rounds half
    to even (default FP environment), and without ``SFC_SET_CLIPPING`` (python-soundfile never sets it) the ``int`` is
    stored into a ``short``, i.e. samples beyond +-1.0 wrap.  Parity unpinned against the library itself (absent offline):
    ``tests/test_pooling_host.py::test_pcm16_round_trip_matches_soundfile`` lights up when ``soundfile`` is installed."""
    x = np.asarray(chunk)
    if x.dtype != np.float32:
        x = x.astype(np.float64)
    q = np.rint(x * x.dtype.type(32767.0)).astype(np.int64)            # float32 product for float32 input, as f2s_array
    return ((q + 32768) % 65536 - 32768).astype(np.int16)              # int -> short


def pcm16_roundtrip(chunk: np.ndarray) -> np.ndarray:
    """Scene audio as the reference's ASR module sees it: written as PCM_16 (``pcm16_encode``) and read back with
    ``sf.read(..., dtype="float32")`` (faster_whisper_pro_asr.py:477), where libsndfile's ``s2f_array`` multiplies by
    ``1.0 / 0x8000``: ``lrint(x * 32767) / 32768`` -- the write and read scales differ, so full scale comes back as
    0.99997, and the quantisation is part of the reference's numerics."""
    return pcm16_encode(chunk).astype(np.float32) / np.float32(32768.0)


def to_16k(audio: np.ndarray, sr: int) -> np.ndarray:
    """Mono float32 at 16 kHz (the pipeline extracts at 16 kHz, modules/audio_extraction.py; a stand-alone driver may be
    handed anything)."""
    audio = np.asarray(audio, dtype=np.float32)
    if audio.ndim > 1:
        audio = audio.mean(axis=1)
    if sr == SR:
        return np.ascontiguousarray(audio)
    from math import gcd
    from scipy.signal import resample_poly
    g = gcd(int(sr), SR)
    return np.ascontiguousarray(resample_poly(audio, SR // g, int(sr) // g).astype(np.float32))


class RecordingTranscriber:
    """Steps 2-4 of ``BalancedPipeline.process`` + stitch for one recording held in memory, scenes pooled."""

    def __init__(self, asr, scene_detector, pcm16_scenes: bool = True, device_resident: bool = True):
        """``device_resident``: upload the recording to HBM once and hand the ASR module scene clips that are views of
        it (the segmenter, the feature extractor and the engine gather them on the device); off = numpy clips on the
        host, the reference's call contract.  Both give the same numbers (the PCM16 round trip is the same IEEE arithmetic either way)."""
        self.asr, self.scene_detector, self.pcm16_scenes = asr, scene_detector, bool(pcm16_scenes)
        self.device_resident = bool(device_resident)
        self.timing: Dict[str, float] = {}

    def detect(self, audio: np.ndarray, sr: int = SR) -> List[Tuple[float, float]]:
        t0 = time.perf_counter()
        found, _ = self.scene_detector.split_clip(audio, sr)
        self.timing["scene_detection_s"] = time.perf_counter() - t0
        return [(float(a), float(b)) for a, b, _, _ in found if int(b * sr) - int(a * sr) > 0]

    def scene_audio(self, audio: np.ndarray, sr: int, scene: Tuple[float, float]) -> np.ndarray:
        chunk = audio[int(scene[0] * sr): int(scene[1] * sr)]
        return pcm16_roundtrip(chunk) if self.pcm16_scenes else np.ascontiguousarray(chunk, dtype=np.float32)

    def transcribe_scenes(self, audio: np.ndarray, sr: int, scenes: Sequence[Tuple[float, float]],
                          pooled: bool = True) -> List[Dict[str, Any]]:
        """Per-scene result dicts (scene-relative times) for ``scenes``; ``pooled=False`` is the reference's call
        pattern (one ``transcribe`` per scene) kept for the equivalence tests and the A/B figure of the bench."""
        clips = self._device_clips(audio, sr, scenes) if self.device_resident else None
        if clips is None:
            clips = [(self.scene_audio(audio, sr, sc), sr) for sc in scenes]
        t0 = time.perf_counter()
        if pooled:
            out = self.asr.transcribe_scenes(clips)
        else:
            out = [self.asr.transcribe_scenes([c])[0] for c in clips]
        self.timing["asr_s"] = time.perf_counter() - t0
        return out

    def _device_clips(self, audio: np.ndarray, sr: int, scenes):
        try:
            import torch
            if not torch.cuda.is_available() or sr != SR:
                return None
        except ImportError:
            return None
        dev = torch.device("cuda", int(getattr(self.scene_detector, "_device", 0)))
        rec = torch.from_numpy(np.ascontiguousarray(audio, dtype=np.float32)).to(dev)
        if self.pcm16_scenes:       # pcm16_roundtrip on the device: float32 product with 0x7FFF, round half to even, int -> short
            q = rec.mul_(32767.0).round_().to(torch.int32)       # wrap, / 0x8000 (every step exact or IEEE-identical to NumPy's)
            rec = (((q + 32768) & 0xFFFF) - 32768).to(torch.float32).div_(32768.0)
        return [(rec[int(a * sr): int(b * sr)], sr) for a, b in scenes]

    @staticmethod
    def stitch(scenes: Sequence[Tuple[float, float]], per_scene: Sequence[Dict[str, Any]]) -> List[Dict[str, Any]]:
        """``SRTStitcher.stitch``: every scene's segments shifted by the scene start, scenes in start order."""
        merged: List[Dict[str, Any]] = []
        for (start, _), res in sorted(zip(scenes, per_scene), key=lambda p: p[0][0]):
            for seg in res.get("segments", []):
                merged.append(dict(seg, start=seg["start"] + start, end=seg["end"] + start))
        return merged

    def transcribe(self, audio: np.ndarray, sr: int = SR, pooled: bool = True) -> Dict[str, Any]:
        audio = to_16k(audio, sr)
        scenes = self.detect(audio, SR)
        per_scene = self.transcribe_scenes(audio, SR, scenes, pooled=pooled)
        return {"scenes": scenes, "per_scene": per_scene, "segments": self.stitch(scenes, per_scene),
                "vad_segments": [[dict(v, start_sec=round(v["start_sec"] + a, 3), end_sec=round(v["end_sec"] + a, 3)) for v in vs]
                                 for (a, _), vs in zip(scenes, self.asr.get_vad_segments_per_scene())] if pooled else None,
                "timing": dict(self.timing)}


class PrimingSceneDetector:
    """Wraps a scene detector of the reference's protocol (scene_detection_backends/base.py:100-182): every attribute
    is forwarded; ``detect_scenes`` additionally hands the list of scene files to ``on_scenes`` before returning."""

    def __init__(self, inner, on_scenes):
        self._inner, self._on_scenes = inner, on_scenes

    def __getattr__(self, name):
        return getattr(self._inner, name)

    def detect_scenes(self, *args, **kwargs):
        result = self._inner.detect_scenes(*args, **kwargs)
        try:
            self._on_scenes([Path(t[0]) for t in result.to_legacy_tuples()])
        except Exception:       # priming is an optimisation: the per-scene path stays correct without it
            pass
        return result


def hip_balanced_pipeline_class():
    """Build ``HipBalancedPipeline`` against the installed ``whisperjav`` package (see the module docstring)."""
    from whisperjav.pipelines import balanced_pipeline as ref  # type: ignore

    from .asr import HipFasterWhisperProASR

    class HipBalancedPipeline(ref.BalancedPipeline):
        """``--mode balanced`` on the MI355X: the reference pipeline with the ASR scene loop pooled."""

        def __init__(self, *args, **kwargs):
            super().__init__(*args, **kwargs)
            self._pending_scene_paths = None
            self.scene_detector = PrimingSceneDetector(self.scene_detector, self._announce_scenes)

        def _announce_scenes(self, scene_paths) -> None:
            # kept until the ASR module is asked for (process() does that after the speech-enhancement phase, so the
            # reference's "one model in VRAM at a time" order is preserved); an enhancer that rewrites the scene files
            # under new names simply leaves the per-scene path in charge
            self._pending_scene_paths = list(scene_paths)

        def _ensure_asr(self):
            if self._asr is None:
                self._asr = HipFasterWhisperProASR(**self._asr_config)
                ref._IMMORTAL_ASR_REFERENCE = self._asr     # same lifetime rule as the reference (:214-222)
            if self._pending_scene_paths is not None:
                self._asr.prime_scenes(self._pending_scene_paths)
                self._pending_scene_paths = None
            return self._asr

        def get_mode_name(self) -> str:
            return "balanced"

    return HipBalancedPipeline


def hip_fidelity_pipeline_class():
    """``HipFidelityPipeline``: the reference's ``FidelityPipeline`` (pipelines/fidelity_pipeline.py:32-490) with the ASR
    module swapped for ``asr.HipWhisperProASR`` (openai-whisper semantics) and the scenes pooled the same way.  Fidelity
    mode creates its ASR module as a LOCAL of ``process()`` (``asr = WhisperProASR(**self._asr_config)``, :270), so the
    subclass binds that module-level name to a factory for the duration of the call -- the one-line alternative for a
    maintainer is the import swap of INTEGRATION.md section 2b."""
    from whisperjav.pipelines import fidelity_pipeline as ref  # type: ignore

    from . import asr as hip_asr

    class HipFidelityPipeline(ref.FidelityPipeline):
        def __init__(self, *args, **kwargs):
            super().__init__(*args, **kwargs)
            self._pending_scene_paths = None
            self.scene_detector = PrimingSceneDetector(self.scene_detector, self._announce_scenes)

        def _announce_scenes(self, scene_paths) -> None:
            self._pending_scene_paths = list(scene_paths)

        def _make_asr(self, **config):
            module = hip_asr.HipWhisperProASR(**config)
            if self._pending_scene_paths is not None:
                module.prime_scenes(self._pending_scene_paths)
                self._pending_scene_paths = None
            return module

        def process(self, media_info):
            saved = ref.WhisperProASR
            ref.WhisperProASR = self._make_asr
            try:
                return super().process(media_info)
            finally:
                ref.WhisperProASR = saved

        def get_mode_name(self) -> str:
            return "fidelity"

    return HipFidelityPipeline


def __getattr__(name):
    if name == "HipBalancedPipeline":
        return hip_balanced_pipeline_class()
    if name == "HipFidelityPipeline":
        return hip_fidelity_pipeline_class()
    raise AttributeError(name)
