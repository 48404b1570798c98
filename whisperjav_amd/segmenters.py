"""Speech-segmenter plugins for WhisperJAV's ``SpeechSegmenterFactory`` backed by the HIP VAD scorer.

Mirrors the reference's plugin surface (same constructor arguments, attribute names, result types
and error behaviour) so the classes register in ``_BACKEND_REGISTRY``
(/root/reference/whisperjav/modules/speech_segmentation/factory.py:17-33) unchanged:

  * ``HipSileroV6SpeechSegmenter``  <->  ``SileroV6SpeechSegmenter``
    (backends/silero_v6.py:28-338): library-side padding / max-speech splitting, then grouping;
    swallows scorer errors and returns an empty result (silero_v6.py:228-237).
  * ``HipSileroSpeechSegmenter``    <->  ``SileroSpeechSegmenter`` (backends/silero.py:38-459):
    the v3.1/v4.0-API flavour -- no max-speech argument, WhisperJAV's own sample padding
    (``start_pad_samples`` / ``end_pad_samples``, clamp to ``len - 16``, overlap fix,
    silero.py:287-297) and grouping (silero.py:325-361); errors propagate.

Both keep the reference's test seam: ``seg._model`` / ``seg._get_speech_timestamps`` can be replaced
with fakes (tests/test_vad_threshold_padding_e2e.py:400-575 in the reference does exactly that).
When the ``whisperjav`` package is importable its own ``SpeechSegment`` / ``SegmentationResult``
dataclasses are used, otherwise the structurally identical mirrors below.
"""
from __future__ import annotations

import logging
import time
from dataclasses import dataclass, field
from pathlib import Path
from typing import Any, Dict, List, Optional, Tuple, Union

import numpy as np

logger = logging.getLogger("whisperjav")

try:  # inside WhisperJAV: use the reference's own contracts
    from whisperjav.modules.speech_segmentation.base import SegmentationResult, SpeechSegment  # type: ignore
except Exception:  # standalone: mirrors of speech_segmentation/base.py:14-141
    @dataclass
    class SpeechSegment:
        start_sec: float
        end_sec: float
        start_sample: int = 0
        end_sample: int = 0
        confidence: float = 1.0
        metadata: Dict[str, Any] = field(default_factory=dict)

        @property
        def duration_sec(self) -> float:
            return self.end_sec - self.start_sec

        def to_dict(self) -> Dict[str, Any]:
            return {"start_sec": round(self.start_sec, 3), "end_sec": round(self.end_sec, 3),
                    "duration_sec": round(self.duration_sec, 3), "confidence": round(self.confidence, 3)}

    @dataclass
    class SegmentationResult:
        segments: List[SpeechSegment]
        groups: List[List[SpeechSegment]]
        method: str
        audio_duration_sec: float
        parameters: Dict[str, Any]
        processing_time_sec: float = 0.0

        @property
        def speech_coverage_sec(self) -> float:
            return sum(s.duration_sec for s in self.segments)

        @property
        def speech_coverage_ratio(self) -> float:
            return self.speech_coverage_sec / self.audio_duration_sec if self.audio_duration_sec > 0 else 0.0

        @property
        def num_segments(self) -> int:
            return len(self.segments)

        @property
        def num_groups(self) -> int:
            return len(self.groups)

        def to_legacy_format(self) -> List[List[Dict]]:
            return [[{"start": s.start_sample, "end": s.end_sample, "start_sec": s.start_sec, "end_sec": s.end_sec,
                      "metadata": s.metadata} for s in g] for g in self.groups]

        def to_flat_legacy_format(self) -> List[Dict]:
            return [{"start_sec": round(s.start_sec, 3), "end_sec": round(s.end_sec, 3)} for s in self.segments]


VAD_SR = 16000


def group_segments(segments: List[SpeechSegment], max_group_duration_s: float = 29.0,
                   chunk_threshold_s: float = 1.0) -> List[List[SpeechSegment]]:
    """Pack consecutive segments into ASR work units: a new group starts when the gap to the previous
    segment exceeds ``chunk_threshold_s`` or the group would span more than ``max_group_duration_s``
    (same rule as backends/ten.py:31-73 and silero.py:325-361)."""
    groups: List[List[SpeechSegment]] = []
    for seg in segments:
        if groups:
            cur = groups[-1]
            gap = seg.start_sec - cur[-1].end_sec
            too_long = (seg.end_sec - cur[0].start_sec) > max_group_duration_s
            if gap > chunk_threshold_s or too_long:
                groups.append([seg])
            else:
                cur.append(seg)
        else:
            groups.append([seg])
    return groups


def _load_audio(audio: Union[np.ndarray, Path, str], sample_rate: int) -> Tuple[np.ndarray, int]:
    if isinstance(audio, np.ndarray):
        return audio, sample_rate
    try:
        import soundfile as sf
    except ImportError:
        raise ImportError("soundfile is required for loading audio files")
    data, sr = sf.read(str(audio), dtype="float32")
    if data.ndim > 1:
        data = np.mean(data, axis=1)
    return data, sr


def _resample(audio: np.ndarray, orig_sr: int, target_sr: int) -> np.ndarray:
    if orig_sr == target_sr:
        return audio
    from scipy import signal
    return signal.resample(audio, int(len(audio) * target_sr / orig_sr)).astype(audio.dtype)


class _HipSileroBase:
    def _ensure_model(self) -> None:
        if self._model is not None:
            return
        from . import vad
        self._model = vad.HipSileroScorer(self._weights, device=self._device)
        if self._get_speech_timestamps is None:
            self._get_speech_timestamps = vad.get_speech_timestamps

    def cleanup(self) -> None:
        model, self._model = self._model, None
        if model is not None and hasattr(model, "close"):
            model.close()

    def get_supported_sample_rates(self) -> List[int]:
        return [16000]


class HipSileroV6SpeechSegmenter(_HipSileroBase):
    """Drop-in for ``SileroV6SpeechSegmenter`` with the window scorer on the MI355X."""

    def __init__(self, threshold: float = 0.35, min_speech_duration_ms: int = 100,
                 max_speech_duration_s: Optional[float] = None, min_silence_duration_ms: int = 100,
                 speech_pad_ms: int = 350, min_silence_at_max_speech: int = 98,
                 use_max_poss_sil_at_max_speech: bool = True, chunk_threshold_s: Optional[float] = 1.0,
                 max_group_duration_s: Optional[float] = None, weights: Optional[Dict[str, np.ndarray]] = None,
                 device: int = 0, **kwargs):
        self.threshold = float(threshold)
        self.min_speech_duration_ms = int(min_speech_duration_ms)
        self.min_silence_duration_ms = int(min_silence_duration_ms)
        self.speech_pad_ms = int(speech_pad_ms)
        self.min_silence_at_max_speech = int(min_silence_at_max_speech)
        self.use_max_poss_sil_at_max_speech = bool(use_max_poss_sil_at_max_speech)
        if chunk_threshold_s is not None:
            self.chunk_threshold_s = float(chunk_threshold_s)
        elif "chunk_threshold" in kwargs:
            self.chunk_threshold_s = float(kwargs["chunk_threshold"])
        else:
            self.chunk_threshold_s = 1.0
        self.max_group_duration_s = float(max_group_duration_s) if max_group_duration_s is not None else 29.0
        self.max_speech_duration_s = (float(max_speech_duration_s) if max_speech_duration_s is not None
                                      else self.max_group_duration_s)
        self._weights, self._device = weights, int(device)
        self._model = None
        self._get_speech_timestamps = None

    @property
    def name(self) -> str:
        return "silero-v6.2-hip"

    @property
    def display_name(self) -> str:
        return "Silero VAD v6.2 (MI355X HIP)"

    def _get_parameters(self) -> Dict[str, Any]:
        return {k: getattr(self, k) for k in (
            "threshold", "min_speech_duration_ms", "max_speech_duration_s", "min_silence_duration_ms",
            "speech_pad_ms", "min_silence_at_max_speech", "use_max_poss_sil_at_max_speech", "chunk_threshold_s",
            "max_group_duration_s")}

    def segment(self, audio: Union[np.ndarray, Path, str], sample_rate: int = 16000, **kwargs) -> SegmentationResult:
        t0 = time.time()
        self._ensure_model()
        data, sr = _load_audio(audio, sample_rate)
        duration = len(data) / sr
        if sr != VAD_SR:
            data, sr = _resample(data, sr, VAD_SR), VAD_SR
        if data.dtype != np.float32:
            data = data.astype(np.float32)
        try:
            stamps = self._get_speech_timestamps(
                data, self._model, sampling_rate=sr, threshold=self.threshold,
                min_speech_duration_ms=self.min_speech_duration_ms, max_speech_duration_s=self.max_speech_duration_s,
                min_silence_duration_ms=self.min_silence_duration_ms, speech_pad_ms=self.speech_pad_ms,
                return_seconds=False, min_silence_at_max_speech=self.min_silence_at_max_speech,
                use_max_poss_sil_at_max_speech=self.use_max_poss_sil_at_max_speech)
            segments = [SpeechSegment(start_sec=ts["start"] / sr, end_sec=ts["end"] / sr, start_sample=ts["start"],
                                      end_sample=ts["end"], confidence=1.0) for ts in stamps]
        except Exception as e:  # same policy as the reference backend: log, return an empty result
            logger.error(f"Silero VAD (HIP) segmentation failed: {e}", exc_info=True)
            return SegmentationResult(segments=[], groups=[], method=self.name, audio_duration_sec=duration,
                                      parameters=self._get_parameters(), processing_time_sec=time.time() - t0)
        groups = group_segments(segments, self.max_group_duration_s, self.chunk_threshold_s)
        return SegmentationResult(segments=segments, groups=groups, method=self.name, audio_duration_sec=duration,
                                  parameters=self._get_parameters(), processing_time_sec=time.time() - t0)


class HipSileroSpeechSegmenter(_HipSileroBase):
    """Drop-in for ``SileroSpeechSegmenter`` (torch.hub v3.1 / v4.0 API flavour).

    NOTE: the v3.1 / v4.0 network weights and architecture are not obtainable offline; this class runs
    the v5/v6-architecture HIP scorer behind the v3.1/v4.0 *call contract* (no ``max_speech_duration_s``
    forwarded, WhisperJAV-side sample padding).  See DESIGN.md "open items"."""

    VERSION_DEFAULTS = {
        "v4.0": {"threshold": 0.25, "min_speech_duration_ms": 150, "min_silence_duration_ms": 300,
                 "speech_pad_ms": 700, "max_speech_duration_s": float("inf"), "max_group_duration_s": 29.0},
        "v3.1": {"threshold": 0.125, "min_speech_duration_ms": 90, "min_silence_duration_ms": 300,
                 "speech_pad_ms": 700, "max_speech_duration_s": float("inf"), "max_group_duration_s": 29.0},
    }

    def __init__(self, version: str = "v4.0", threshold: Optional[float] = None,
                 min_speech_duration_ms: Optional[int] = None, min_silence_duration_ms: Optional[int] = None,
                 speech_pad_ms: Optional[int] = None, chunk_threshold_s: Optional[float] = None,
                 max_group_duration_s: Optional[float] = None, max_speech_duration_s: Optional[float] = None,
                 start_pad_samples: int = 11200, end_pad_samples: int = 20800,
                 weights: Optional[Dict[str, np.ndarray]] = None, device: int = 0, **kwargs):
        self.version = version if version in self.VERSION_DEFAULTS else "v4.0"
        dflt = self.VERSION_DEFAULTS[self.version]
        self.threshold = float(threshold) if threshold is not None else dflt["threshold"]
        self.min_speech_duration_ms = (int(min_speech_duration_ms) if min_speech_duration_ms is not None
                                       else dflt["min_speech_duration_ms"])
        self.min_silence_duration_ms = (int(min_silence_duration_ms) if min_silence_duration_ms is not None
                                        else dflt["min_silence_duration_ms"])
        self.speech_pad_ms = int(speech_pad_ms) if speech_pad_ms is not None else dflt["speech_pad_ms"]
        self.max_speech_duration_s = (float(max_speech_duration_s) if max_speech_duration_s is not None
                                      else dflt["max_speech_duration_s"])
        if chunk_threshold_s is not None:
            self.chunk_threshold_s = float(chunk_threshold_s)
        elif "chunk_threshold" in kwargs:
            self.chunk_threshold_s = float(kwargs["chunk_threshold"])
        else:
            self.chunk_threshold_s = 4.0
        self.max_group_duration_s = (float(max_group_duration_s) if max_group_duration_s is not None
                                     else dflt["max_group_duration_s"])
        self.start_pad_samples = int(start_pad_samples)
        self.end_pad_samples = int(end_pad_samples)
        self._weights, self._device = weights, int(device)
        self._model = None
        self._utils = None
        self._get_speech_timestamps = None

    @property
    def name(self) -> str:
        return f"silero-{self.version}-hip"

    @property
    def display_name(self) -> str:
        return f"Silero VAD {self.version} (MI355X HIP)"

    def _get_parameters(self) -> Dict[str, Any]:
        return {k: getattr(self, k) for k in (
            "version", "threshold", "min_speech_duration_ms", "min_silence_duration_ms", "speech_pad_ms",
            "max_speech_duration_s", "chunk_threshold_s", "max_group_duration_s", "start_pad_samples",
            "end_pad_samples")}

    def segment(self, audio: Union[np.ndarray, Path, str], sample_rate: int = 16000, **kwargs) -> SegmentationResult:
        t0 = time.time()
        self._ensure_model()
        data, sr = _load_audio(audio, sample_rate)
        duration = len(data) / sr
        threshold = kwargs.get("threshold", self.threshold)
        min_speech = kwargs.get("min_speech_duration_ms", self.min_speech_duration_ms)
        min_silence = kwargs.get("min_silence_duration_ms", self.min_silence_duration_ms)
        pad_ms = kwargs.get("speech_pad_ms", self.speech_pad_ms)
        audio16 = np.asarray(_resample(data, sr, VAD_SR), dtype=np.float32)
        stamps = self._get_speech_timestamps(audio16, self._model, sampling_rate=VAD_SR, threshold=threshold,
                                             min_speech_duration_ms=min_speech, min_silence_duration_ms=min_silence,
                                             speech_pad_ms=pad_ms)
        if not stamps:
            return SegmentationResult(segments=[], groups=[], method=self.name, audio_duration_sec=duration,
                                      parameters=self._get_parameters(), processing_time_sec=time.time() - t0)
        n = len(audio16)
        prev_end = None
        segments: List[SpeechSegment] = []
        for ts in stamps:
            start = max(0, int(ts["start"]) - self.start_pad_samples)
            end = min(n - 16, int(ts["end"]) + self.end_pad_samples)
            if prev_end is not None and start < prev_end:
                start = prev_end
            prev_end = end
            segments.append(SpeechSegment(start_sec=start / VAD_SR, end_sec=end / VAD_SR, start_sample=start,
                                          end_sample=end, confidence=1.0, metadata={}))
        groups = group_segments(segments, self.max_group_duration_s, self.chunk_threshold_s)
        return SegmentationResult(segments=segments, groups=groups, method=self.name, audio_duration_sec=duration,
                                  parameters=self._get_parameters(), processing_time_sec=time.time() - t0)


REGISTRY_ENTRIES = {
    # add these to whisperjav/modules/speech_segmentation/factory.py:_BACKEND_REGISTRY
    "silero-hip": "whisperjav_amd.segmenters.HipSileroV6SpeechSegmenter",
    "silero-v6.2-hip": "whisperjav_amd.segmenters.HipSileroV6SpeechSegmenter",
    "silero-v4.0-hip": "whisperjav_amd.segmenters.HipSileroSpeechSegmenter",
    "silero-v3.1-hip": "whisperjav_amd.segmenters.HipSileroSpeechSegmenter",
}
