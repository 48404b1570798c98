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


def _on_device(audio) -> bool:
    return hasattr(audio, "is_cuda") and bool(audio.is_cuda)


def _load_audio(audio: Union[np.ndarray, Path, str], sample_rate: int) -> Tuple[np.ndarray, int]:
    if isinstance(audio, np.ndarray) or _on_device(audio):      # device tensors: clips of a recording resident in HBM
        return audio, sample_rate
    if hasattr(audio, "detach"):
        return audio.detach().cpu().numpy(), sample_rate
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
    _weights_path: Optional[str] = None

    def _ensure_model(self) -> None:
        if self._model is not None:
            return
        from . import vad
        # trained Silero parameters by default (weights_path / the silero_vad package); weights="synthetic" is an
        # explicit opt-in for tests and benchmarks -- see vad.HipSileroScorer
        self._model = vad.HipSileroScorer(self._weights, device=self._device, weights_path=self._weights_path)
        if self._get_speech_timestamps is None:
            self._get_speech_timestamps = vad.get_speech_timestamps

    def segment_many(self, audios, sample_rates) -> List["SegmentationResult"]:
        """``segment`` for many clips (scenes) with ONE scorer launch: every clip is a stream of the batched HIP
        scorer (``wj_vad_scores`` scores them concurrently, state reset per stream), then the per-clip region logic
        runs exactly as in ``segment``.  Falls back to a loop when the scorer seam has been replaced (the test
        doubles of the reference's suite) or a clip needs resampling / loading."""
        from . import vad
        self._ensure_model()
        if isinstance(sample_rates, int):
            sample_rates = [sample_rates] * len(audios)
        batched = (self._get_speech_timestamps is vad.get_speech_timestamps and hasattr(self._model, "scores")
                   and all((isinstance(a, np.ndarray) or _on_device(a)) and sr == VAD_SR for a, sr in zip(audios, sample_rates)))
        if not batched:
            return [self.segment(a, sample_rate=sr) for a, sr in zip(audios, sample_rates)]
        try:
            probs = self._model.scores(list(audios))
        except Exception as e:      # per-clip calls reproduce each class's own error policy (swallow / propagate)
            logger.error(f"batched VAD scoring failed ({e}); falling back to per-clip calls")
            return [self.segment(a, sample_rate=sr) for a, sr in zip(audios, sample_rates)]
        return [self.segment(a, sample_rate=sr, _probs=p) for a, sr, p in zip(audios, sample_rates, probs)]

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
                 max_group_duration_s: Optional[float] = None, weights: Union[Dict[str, np.ndarray], str, None] = None,
                 device: int = 0, weights_path: Optional[str] = None, **kwargs):
        self.threshold = float(threshold)
        self.min_speech_duration_ms = int(min_speech_duration_ms)
        self.min_silence_duration_ms = int(min_silence_duration_ms)
        self.speech_pad_ms = int(speech_pad_ms)
        self._weights_path = weights_path
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
        if _on_device(data) and sr != VAD_SR:
            data = data.detach().cpu().numpy()
        if sr != VAD_SR:
            data, sr = _resample(data, sr, VAD_SR), VAD_SR
        if isinstance(data, np.ndarray) and data.dtype != np.float32:
            data = data.astype(np.float32)
        try:
            stamps = self._get_speech_timestamps(
                data, self._model, sampling_rate=sr, threshold=self.threshold,
                min_speech_duration_ms=self.min_speech_duration_ms, max_speech_duration_s=self.max_speech_duration_s,
                min_silence_duration_ms=self.min_silence_duration_ms, speech_pad_ms=self.speech_pad_ms,
                return_seconds=False, min_silence_at_max_speech=self.min_silence_at_max_speech,
                use_max_poss_sil_at_max_speech=self.use_max_poss_sil_at_max_speech,
                **({"probs": kwargs["_probs"]} if kwargs.get("_probs") is not None else {}))
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
    """``SileroSpeechSegmenter``'s call contract (torch.hub v3.1 / v4.0 flavour, backends/silero.py:214-323: no
    ``max_speech_duration_s`` forwarded, WhisperJAV-side sample padding and overlap fix, 4 s chunk threshold).

    The v3.1 / v4.0 NETWORK (the reference's balanced default, main.py:1867-1876; 1536-sample windows, TorchScript
    archives from ``torch.hub``, backends/silero.py:68-72,199-206) has no HIP kernel: neither the archives nor their source
    are reachable offline and the graph cannot be reconstructed from parameter names.  This class therefore REFUSES
    to score (``WjError`` from ``segment``) unless the caller asks for the substitution by name: ``network="v6"`` runs
    the v5/v6 HIP scorer behind this contract -- a different network, different probabilities, NOT the reference's VAD
    frames -- and says so in ``name`` / ``display_name``.  A scorer seam injected by a test double (``_model`` /
    ``_get_speech_timestamps``) bypasses the check, as in the reference's own suite.

    ``scorer`` (round 4) keeps the reference's OWN network when its archive is on the box: the reference's JIT model scores on
    the host, on its 1536-sample grid, through the archive's own ``get_speech_timestamps`` -- exactly the call of
    backends/silero.py:258-273 -- and everything downstream (padding, grouping, log-mel, encoder, search) stays on the device.
    Accepted: ``"torch.hub"`` (``torch.hub.load("snakers4/silero-vad:<version>", "silero_vad", onnx=False)``, the reference's
    loader :199-206, served from the local hub cache), a ``(model, get_speech_timestamps)`` pair, the ``(model, utils)`` pair
    ``torch.hub.load`` returns, or a zero-argument callable returning one of these (loaded on first use).  ``--mode balanced``
    with the reference's default segmenter then runs end to end: ``HipSileroSpeechSegmenter(version="v3.1",
    scorer="torch.hub")``."""

    REPOS = {"v4.0": "snakers4/silero-vad:v4.0", "v3.1": "snakers4/silero-vad:v3.1"}      # backends/silero.py:68-72

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
                 weights: Union[Dict[str, np.ndarray], str, None] = None, device: int = 0,
                 weights_path: Optional[str] = None, network: Optional[str] = None, scorer: Any = None,
                 device_scoring: Optional[bool] = None, window_size_samples: Optional[int] = None, region_route: str = "certified",
                 **kwargs):
        self._weights_path = weights_path
        # round 6: how device probabilities become regions when the caller handed over the archive's own get_speech_timestamps
        # (scorer=): "archive" = always through that function (a replay model answers its model(chunk, sr) calls: ~6 us of host
        # Python per window, 0.5 s per 120 min); "certified" = the function is first run against vad.regions_from_probs (the same
        # state machine restated, vectorised) on a battery of probability tracks AT THE CALL'S PARAMETERS -- equal on every track:
        # the restated machine serves the scenes and one scene per call is still replayed through the archive's function as a
        # spot check; any difference, at certification or later: the archive's function serves everything from then on
        if region_route not in ("certified", "archive"):
            raise ValueError("region_route must be 'certified' or 'archive'")
        self._region_route = region_route
        self._certified: Dict[Tuple, bool] = {}
        self.region_stats = {"certified": 0, "replayed": 0, "spot_checks": 0, "mismatches": 0}
        # round 5: a TorchScript archive (weights_path=<model.jit>, or the model of scorer=) is LOWERED onto the device
        # (vad_graph.py).  device_scoring: None = lower whenever the model is a TorchScript module (a graph outside the loader's
        # op table raises LoweringError), True = require it, False = round 4's host scoring through the archive's own loop
        self._device_scoring = device_scoring
        self._window = int(window_size_samples) if window_size_samples else None
        self._graph_scorer = None
        self._gst_takes_window: Optional[bool] = None
        if network not in (None, "v6", "v5/v6"):
            raise ValueError("network must be None (the version's own network: no HIP kernel, refuses) or 'v6'")
        if scorer is not None and network:
            raise ValueError("scorer= (the version's own network on the host) and network='v6' (the v6 HIP network) are alternatives")
        self.network = "v6" if network else None
        self._scorer = scorer
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
        if self._graph_scorer is not None or (self._scorer is None and self._is_archive_path()):
            return f"silero-{self.version}-hip+graph"
        return f"silero-{self.version}-hip" + ("+v6net" if self.network else "+hostnet" if self._scorer is not None else "")

    @property
    def display_name(self) -> str:
        if self.network:
            return f"Silero VAD {self.version} call contract over the v6 network (MI355X HIP)"
        if self._graph_scorer is not None or (self._scorer is None and self._is_archive_path()):
            return f"Silero VAD {self.version} (the archive's own network lowered onto the MI355X: TorchScript graph -> HIP)"
        if self._scorer is not None:
            return f"Silero VAD {self.version} (the reference's own network scoring on the host; MI355X HIP downstream)"
        return f"Silero VAD {self.version} (MI355X HIP; no kernel for this network: refuses to score)"

    @property
    def can_score(self) -> bool:
        """False = ``segment`` would refuse (no HIP kernel for this network, no scorer seam, no v6 substitution asked for)."""
        return self.network is not None or self._scorer is not None or self._model is not None or self._is_archive_path()

    def _is_archive_path(self) -> bool:
        p = self._weights_path or (self._weights if isinstance(self._weights, str) and self._weights != "synthetic" else None)
        return bool(p) and self.network is None and not str(p).lower().endswith((".npz", ".safetensors"))

    def _resolve_scorer(self) -> None:
        sc = self._scorer
        if sc == "torch.hub":
            import torch
            model, utils = torch.hub.load(repo_or_dir=self.REPOS[self.version], model="silero_vad", onnx=False, trust_repo=True)
            sc = (model, utils)
        elif callable(sc) and not isinstance(sc, (tuple, list)):
            sc = sc()
        if not isinstance(sc, (tuple, list)) or len(sc) != 2:
            raise TypeError("scorer must be 'torch.hub', (model, get_speech_timestamps), torch.hub.load's (model, utils), or a callable returning one")
        model, second = sc
        gst = second[0] if isinstance(second, (tuple, list)) else second         # utils = (get_speech_timestamps, save_audio, ...)
        if not callable(gst):
            raise TypeError("scorer: no callable get_speech_timestamps found")
        self._model, self._get_speech_timestamps = model, gst
        self._host_scorer = True
        self._lower_if_scripted(model, gst)

    def _lower_if_scripted(self, model: Any, gst: Any) -> None:
        """The archive's network on the device: lower its graph (vad_graph.lower) and keep the archive's own
        ``get_speech_timestamps`` for the regions -- it is handed a replay model that answers its ``model(chunk, sr)`` calls
        from the probabilities the device computed for the same window grid."""
        import torch
        if self._device_scoring is False:
            return
        scripted = isinstance(model, (torch.jit.ScriptModule, torch.jit.RecursiveScriptModule))
        if not scripted:
            if self._device_scoring:
                from .hipbind import WjError
                raise WjError(f"silero-{self.version}-hip: device_scoring=True needs a TorchScript model to lower, got {type(model).__name__}")
            return
        from . import vad_graph
        window = self._window
        if window is None and gst is not None:
            try:
                import inspect
                window = int(inspect.signature(gst).parameters["window_size_samples"].default)
            except Exception:
                window = None
        self._window = int(window or 1536)       # utils_vad.get_speech_timestamps' default grid for the v3.1 / v4.0 hub archives
        self._graph_scorer = vad_graph.HipGraphVadScorer(model, window=self._window, sample_rate=VAD_SR, device=self._device)
        self._host_scorer = False
        self._archive_gst = gst

    def _ensure_model(self) -> None:
        if self._model is None and self._scorer is not None:
            self._resolve_scorer()
            return
        if self._model is None and self._is_archive_path():
            from . import vad, vad_graph, vad_weights
            path = str(self._weights_path or self._weights)
            archive = vad_graph.load_archive(path)
            try:
                generation = vad_weights.classify_state_dict(archive.state_dict())
            except ValueError:
                generation = "unknown"
            if generation == "v5/v6":
                from .hipbind import WjError
                raise WjError(f"{path} is a Silero v5/v6 archive (512-sample windows): use 'silero-v6.2-hip' for it, or network='v6' to "
                              f"run it behind the {self.version} call contract knowingly")
            self._model = archive
            self._lower_if_scripted(archive, None)
            if self._graph_scorer is None:
                from .hipbind import WjError
                raise WjError(f"silero-{self.version}-hip: weights_path={path!r} with device_scoring=False: host scoring needs scorer=(model, "
                              "get_speech_timestamps)")
            self._get_speech_timestamps = vad.get_speech_timestamps
            return
        if self._model is None and self.network is None:
            from .hipbind import WjError
            raise WjError(
                f"silero-{self.version}-hip: the Silero {self.version} network (1536-sample windows, torch.hub "
                "snakers4/silero-vad) has no HIP kernel and is never replaced silently.  Use 'silero-v6.2-hip' (the "
                "v5/v6 network, its own thresholds), or pass network='v6' to run the v6 scorer behind the "
                f"{self.version} call contract knowingly; vad_weights.from_torchscript(path) identifies an archive.")
        super()._ensure_model()

    def _archive_regions(self, gst, probs, n: int, key: Tuple) -> List[Dict[str, int]]:
        """Regions of one clip through the archive's OWN get_speech_timestamps, its ``model(chunk, sr)`` calls answered from ``probs``."""
        import torch
        threshold, min_speech, min_silence, pad_ms, window = key
        kw = dict(sampling_rate=VAD_SR, threshold=threshold, min_speech_duration_ms=min_speech, min_silence_duration_ms=min_silence,
                  speech_pad_ms=pad_ms)
        if self._gst_takes_window is None:
            try:
                import inspect
                self._gst_takes_window = "window_size_samples" in inspect.signature(gst).parameters
            except (TypeError, ValueError):
                self._gst_takes_window = False
        if self._gst_takes_window:
            kw["window_size_samples"] = window
        # the loop only needs the clip's LENGTH (it slices windows and hands them to the model): zeros of the same length
        return [dict(ts) for ts in gst(torch.zeros(n, dtype=torch.float32), _ReplayModel(probs, window), **kw)]

    def _restated_regions(self, probs, n: int, key: Tuple) -> List[Dict[str, int]]:
        from . import vad
        threshold, min_speech, min_silence, pad_ms, window = key
        return vad.regions_from_probs(probs, n, threshold=threshold, sampling_rate=VAD_SR, min_speech_duration_ms=min_speech,
                                      max_speech_duration_s=float("inf"), min_silence_duration_ms=min_silence, speech_pad_ms=pad_ms,
                                      neg_threshold=threshold - 0.15, window=window)

    def _certify_regions(self, gst, key: Tuple) -> bool:
        """The archive's get_speech_timestamps against the restated state machine, at these parameters, on probability tracks built to
        visit every branch: plateaus around both thresholds (values exactly on them included), silences shorter and longer than
        min_silence, speech shorter and longer than min_speech, tracks that end inside speech, clip lengths on and off the window
        grid, neighbours closer than twice the padding."""
        threshold, _, _, _, window = key
        rng = np.random.default_rng(20240)
        levels = np.asarray([0.0, threshold - 0.15, np.nextafter(np.float32(threshold - 0.15), np.float32(0)), threshold - 0.07, threshold,
                             np.nextafter(np.float32(threshold), np.float32(0)), min(1.0, threshold + 0.2), 1.0], dtype=np.float32).clip(0, 1)
        try:
            for trial in range(64):
                n_win = int(rng.integers(1, 90)) if trial else 1
                if trial % 3 == 0:
                    track = rng.random(n_win).astype(np.float32)
                else:
                    runs = rng.integers(1, 9 if trial % 3 == 1 else 30, size=n_win)
                    track = np.repeat(levels[rng.integers(0, len(levels), size=n_win)], runs)[:n_win].astype(np.float32)
                n = (n_win - 1) * window + (window if trial % 4 == 0 else int(rng.integers(1, window + 1)))
                if self._archive_regions(gst, track, n, key) != self._restated_regions(track, n, key):
                    logger.warning("silero-%s-hip: this archive's get_speech_timestamps is not the v3.1 / v4.0 state machine restated in "
                                   "vad.regions_from_probs (track %d): regions will come from the archive's function", self.version, trial)
                    return False
        except Exception as e:       # a signature or a check this route does not know: the archive's function decides by itself
            logger.warning("silero-%s-hip: certification of the archive's get_speech_timestamps failed (%s)", self.version, e)
            return False
        return True

    def _graph_regions(self, audio16, probs, threshold, min_speech, min_silence, pad_ms, spot_check: bool = True) -> List[Dict[str, int]]:
        """Window probabilities from the lowered archive (one launch group; ``probs`` when ``segment_many`` scored the pool
        already) -> regions: through the archive's own ``get_speech_timestamps`` when the caller handed it over (scorer=), else
        through the v3.1 / v4.0 state machine restated in ``vad.regions_from_probs`` (no speech cap, lower threshold =
        threshold - 0.15 as utils_vad has it for these versions)."""
        from . import vad
        n = int(audio16.numel()) if _on_device(audio16) else int(len(audio16))
        if n == 0:
            return []
        if probs is None:
            probs = self._graph_scorer.scores([audio16])[0]
        gst = getattr(self, "_archive_gst", None)
        if gst is not None:
            key = (float(threshold), int(min_speech), int(min_silence), int(pad_ms), int(self._window))
            if self._region_route == "certified" and key not in self._certified:
                self._certified[key] = self._certify_regions(gst, key)
            if self._region_route == "certified" and self._certified[key]:
                got = self._restated_regions(probs, n, key)
                if spot_check:
                    self.region_stats["spot_checks"] += 1
                    want = self._archive_regions(gst, probs, n, key)
                    if want != got:       # never seen; if it happens the archive's own function takes over for good
                        logger.warning("silero-%s-hip: the archive's get_speech_timestamps and the restated state machine differ on a scene; "
                                       "regions come from the archive's function from now on", self.version)
                        self.region_stats["mismatches"] += 1
                        self._certified[key] = False
                        return want
                self.region_stats["certified"] += 1
                return got
            self.region_stats["replayed"] += 1
            return self._archive_regions(gst, probs, n, key)
        return self._restated_regions(probs, n, (float(threshold), int(min_speech), int(min_silence), int(pad_ms), int(self._window)))

    def segment_many(self, audios, sample_rates) -> List["SegmentationResult"]:
        self._ensure_model()
        if self._graph_scorer is None:
            return super().segment_many(audios, sample_rates)
        if isinstance(sample_rates, int):
            sample_rates = [sample_rates] * len(audios)
        if not all((isinstance(a, np.ndarray) or _on_device(a)) and sr == VAD_SR for a, sr in zip(audios, sample_rates)):
            return [self.segment(a, sample_rate=sr) for a, sr in zip(audios, sample_rates)]
        probs = self._graph_scorer.scores(list(audios))       # every scene a stream of ONE launch group
        # certified region route: the longest scene of the call is still replayed through the archive's own function
        check = int(np.argmax([len(p) for p in probs])) if len(probs) else -1
        return [self.segment(a, sample_rate=sr, _probs=p, _spot_check=(i == check)) for i, (a, sr, p) in enumerate(zip(audios, sample_rates, probs))]

    def cleanup(self) -> None:
        g, self._graph_scorer = self._graph_scorer, None
        if g is not None:
            g.close()
        if self._model is not None and not hasattr(self._model, "close"):
            self._model = None
        super().cleanup()

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
        if _on_device(data) and (sr != VAD_SR or getattr(self, "_host_scorer", False)):
            data = data.detach().cpu().numpy()
        audio16 = data if _on_device(data) else np.asarray(_resample(data, sr, VAD_SR), dtype=np.float32)
        if self._graph_scorer is not None:
            stamps = self._graph_regions(audio16, kwargs.get("_probs"), threshold, min_speech, min_silence, pad_ms,
                                         spot_check=kwargs.get("_spot_check", True))
        elif getattr(self, "_host_scorer", False):
            # the reference's own call (backends/silero.py:258-273): a host FloatTensor, the archive's get_speech_timestamps,
            # the four keyword arguments the v3.1 / v4.0 API takes -- the network's 1536-sample windows and state live in there
            import torch
            stamps = self._get_speech_timestamps(torch.from_numpy(np.ascontiguousarray(audio16, dtype=np.float32)), self._model,
                                                 sampling_rate=VAD_SR, threshold=threshold, min_speech_duration_ms=min_speech,
                                                 min_silence_duration_ms=min_silence, speech_pad_ms=pad_ms)
            stamps = [dict(ts) for ts in stamps]
        else:
            stamps = self._get_speech_timestamps(audio16, self._model, sampling_rate=VAD_SR, threshold=threshold,
                                                 min_speech_duration_ms=min_speech, min_silence_duration_ms=min_silence,
                                                 speech_pad_ms=pad_ms,
                                                 **({"probs": kwargs["_probs"]} if kwargs.get("_probs") is not None else {}))
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


class _ReplayModel:
    """What the archive's ``get_speech_timestamps`` sees in place of the JIT model when the network ran on the device: its
    ``model(chunk, sr)`` calls are answered, in order, from the device's probabilities for the same window grid."""

    def __init__(self, probs: np.ndarray, window: int):
        import torch
        self._probs, self._window, self._i = np.ascontiguousarray(probs, dtype=np.float32), int(window), 0
        self._t = torch.from_numpy(self._probs)        # 0-d views of it answer the calls: no tensor is built per window

    def reset_states(self) -> None:
        self._i = 0

    def __call__(self, chunk, sr=VAD_SR):
        if int(chunk.shape[-1]) != self._window:
            raise ValueError(f"the archive's get_speech_timestamps scores {int(chunk.shape[-1])}-sample windows, the device scored "
                             f"{self._window}-sample ones: pass window_size_samples={int(chunk.shape[-1])}")
        if self._i >= len(self._probs):
            raise IndexError(f"the archive's get_speech_timestamps asks for window {self._i} of {len(self._probs)}")
        out = self._t[self._i]
        self._i += 1
        return out


class NullSpeechSegmenter:
    """Stand-alone mirror of the reference's passthrough segmenter (backends/none.py:21-92): the whole clip is one
    segment / one group, ``name == "none"`` (the ASR modules then transcribe the full scene).  Inside WhisperJAV the
    reference's own class is created by its factory; this one serves ``--speech-segmenter none`` without it."""

    def __init__(self, **kwargs):
        pass

    @property
    def name(self) -> str:
        return "none"

    @property
    def display_name(self) -> str:
        return "No Segmentation"

    def segment(self, audio: Union[np.ndarray, Path, str], sample_rate: int = 16000, **kwargs) -> SegmentationResult:
        t0 = time.time()
        data, sr = _load_audio(audio, sample_rate)
        duration = len(data) / sr
        seg = SpeechSegment(start_sec=0.0, end_sec=duration, start_sample=0, end_sample=len(data), confidence=1.0,
                            metadata={"bypass": True})
        return SegmentationResult(segments=[seg], groups=[[seg]], method=self.name, audio_duration_sec=duration,
                                  parameters={"mode": "passthrough"}, processing_time_sec=time.time() - t0)

    def cleanup(self) -> None:
        pass

    def get_supported_sample_rates(self) -> List[int]:
        return []


class BatchFrameScorer:
    """Adapter that gives a whole-clip scorer the frame API of ``ten_vad.TenVad`` (``process(frame)`` then
    ``out_flags.value`` / ``out_probability.value``, backends/ten.py:232-239), so the reference's TEN post-ops can
    run unchanged on top of it.  ``clip_probs(int16 clip, hop) -> float array`` (one probability per hop) is called
    once per clip -- on the first frame -- and the frames are then served from the result."""

    class _Box:
        value = 0

    def __init__(self, clip_probs, threshold: float, hop_size: int):
        self._clip_probs, self.threshold, self.hop_size = clip_probs, float(threshold), int(hop_size)
        self.out_flags, self.out_probability = self._Box(), self._Box()
        self._probs: Optional[np.ndarray] = None
        self._i = 0

    def begin_clip(self, audio_int16: np.ndarray) -> None:
        self._probs = np.asarray(self._clip_probs(audio_int16, self.hop_size), dtype=np.float64)
        self._i = 0

    def process(self, frame) -> None:
        p = float(self._probs[self._i]) if self._probs is not None and self._i < len(self._probs) else 0.0
        self._i += 1
        self.out_probability.value = p
        self.out_flags.value = int(p >= self.threshold)


def hip_ten_segmenter_class():
    """``HipTenSpeechSegmenter``: the reference's ``TenSpeechSegmenter`` (backends/ten.py:75-520) with every post-op
    -- flags -> segments with the max-speech cut, silence merge, padding, split at probability minima, grouping --
    inherited UNCHANGED and only the frame scorer pluggable:

      * ``scorer="ten"``    the ``ten_vad`` package's own model (closed native library, CPU) -- the reference's result;
      * ``scorer="silero"`` the HIP Silero scorer's window probabilities resampled onto the TEN hop grid (one launch
                            per clip on the MI355X; a different network, so different probabilities -- opt-in);
      * a callable ``(int16 clip, hop) -> probabilities`` for anything else (tests use this).

    Needs the ``whisperjav`` package for the post-ops (they are the reference's code, not restated here)."""
    from whisperjav.modules.speech_segmentation.backends.ten import TenSpeechSegmenter  # type: ignore

    class HipTenSpeechSegmenter(TenSpeechSegmenter):
        def __init__(self, *args, scorer="ten", weights=None, weights_path=None, device: int = 0, **kwargs):
            super().__init__(*args, **kwargs)
            self._scorer_kind, self._weights, self._weights_path, self._device = scorer, weights, weights_path, int(device)
            self._silero = None

        @property
        def name(self) -> str:
            return "ten" if self._scorer_kind == "ten" else "ten-hip"

        @property
        def display_name(self) -> str:
            return "TEN VAD" if self._scorer_kind == "ten" else "TEN post-processing over a pluggable scorer (MI355X)"

        def _silero_probs(self, audio_int16: np.ndarray, hop: int) -> np.ndarray:
            from . import vad
            if self._silero is None:
                self._silero = vad.HipSileroScorer(self._weights, device=self._device, weights_path=self._weights_path)
            win = self._silero.scores([audio_int16.astype(np.float32) / 32768.0])[0]
            n_frames = (len(audio_int16) + hop - 1) // hop
            idx = np.minimum((np.arange(n_frames) * hop) // vad.WINDOW, max(len(win) - 1, 0))
            return win[idx] if len(win) else np.zeros(n_frames)

        def _ensure_model(self) -> None:
            if self._model is not None:
                return
            if self._scorer_kind == "ten":
                return super()._ensure_model()
            fn = self._silero_probs if self._scorer_kind == "silero" else self._scorer_kind
            if not callable(fn):
                raise ValueError("scorer must be 'ten', 'silero' or a callable (int16 clip, hop) -> probabilities")
            self._model = BatchFrameScorer(fn, self.threshold, self.hop_size)

        def _convert_to_int16(self, audio_data):
            out = super()._convert_to_int16(audio_data)
            if isinstance(self._model, BatchFrameScorer):
                self._model.begin_clip(out)        # the one scorer launch of this clip
            return out

        def cleanup(self) -> None:
            if self._silero is not None:
                self._silero.close()
                self._silero = None
            if hasattr(super(), "cleanup"):
                super().cleanup()

    return HipTenSpeechSegmenter


def __getattr__(name):      # ``segmenters.HipTenSpeechSegmenter`` resolves lazily (it needs the whisperjav package)
    if name == "HipTenSpeechSegmenter":
        return hip_ten_segmenter_class()
    raise AttributeError(name)


REGISTRY_ENTRIES = {
    # add these to whisperjav/modules/speech_segmentation/factory.py:_BACKEND_REGISTRY
    "silero-hip": "whisperjav_amd.segmenters.HipSileroV6SpeechSegmenter",
    "silero-v6.2-hip": "whisperjav_amd.segmenters.HipSileroV6SpeechSegmenter",
    "silero-v4.0-hip": "whisperjav_amd.segmenters.HipSileroSpeechSegmenter",
    "silero-v3.1-hip": "whisperjav_amd.segmenters.HipSileroSpeechSegmenter",
    "ten-hip": "whisperjav_amd.segmenters.HipTenSpeechSegmenter",
}
