"""``FasterWhisperProASR``-shaped ASR module running on the MI355X engine.

Duck-type of /root/reference/whisperjav/modules/faster_whisper_pro_asr.py (class at :31): the
pipelines only use ``transcribe_to_srt(audio_path, output_srt_path, task=...)``,
``reset_statistics()``, ``get_last_vad_segments()``, ``get_filter_statistics()``, ``cleanup()`` and the
attribute ``model_name`` (pipelines/balanced_pipeline.py:398-403,483,497,580), and construct it as
``(model_config, params, task, tracer=None)`` (:34).  This class keeps that surface and the per-scene
semantics -- speech segmentation, per-group transcription, high/low suppress lists, optional
post-model log-prob gate, timestamps shifted by the group start -- but hands ALL VAD groups of a
scene to the engine in one batched call instead of a Python loop of batch-1 upstream calls.

Cross-scene pooling (round 2): ``transcribe_scenes([...])`` runs the SAME per-scene procedure for many scenes with
ONE speech-segmentation launch and ONE pooled ``transcribe_many`` over the groups of all of them -- a scene only
yields ~5 groups, the MI355X wants hundreds of windows per launch (45x real-time at batch 1, ~1900x at batch 384).
The reference's pipeline loop calls ``transcribe_to_srt`` scene by scene (pipelines/balanced_pipeline.py:476-486);
``prime_scenes(paths)`` lets whoever knows the scene list (``whisperjav_amd.pipeline``: a priming wrapper around the
scene detector) announce it, the first per-scene call then transcribes the whole list pooled and the following calls
are served from the cache -- same per-scene results, same SRT files, pooled throughput behind the unchanged loop.
"""
from __future__ import annotations

import datetime
import logging
import wave
from pathlib import Path
from typing import Any, Dict, List, Optional, Union

import numpy as np

logger = logging.getLogger("whisperjav")

try:  # inside WhisperJAV the reference's own helpers are used unchanged
    from whisperjav.modules.segment_filters import SegmentFilterConfig, SegmentFilterHelper  # type: ignore
    from whisperjav.modules.vad_failover import should_force_full_transcribe  # type: ignore
except Exception:
    from dataclasses import dataclass

    @dataclass
    class SegmentFilterConfig:  # stand-alone mirror of modules/segment_filters.py:10-34 (same fields, same defaults)
        enabled: bool = True
        logprob_threshold: Optional[float] = None
        logprob_margin: float = 0.0
        drop_nonverbal_vocals: bool = False
        short_segment_window: float = 1.6

    class SegmentFilterHelper:
        def __init__(self, config: SegmentFilterConfig):
            self.enabled = bool(config.enabled)
            self.threshold = config.logprob_threshold
            self.margin = max(0.0, config.logprob_margin or 0.0)
            self.short_window = max(0.4, float(config.short_segment_window or 1.6))
            self.drop_nonverbal = bool(config.drop_nonverbal_vocals)

        # what the reference treats as "not speech" (segment_filters.py:40-72): descriptor words, music notes, and
        # up-to-six-character strings made only of vocalisation kana / letters once punctuation is removed
        _WORDS = ("music applause laugh laughs laughter sfx fx noise silence ambient moan moans moaning groan groans sigh "
                  "sighs breath breathing 喘 喘ぎ 喘ぎ声 うめき うめき声").split()
        _NOTES = frozenset("♪♫")
        _VOCAL = frozenset("ahmnou" "ぁあァアんンっッふフぅゥうウおオえエはハほホ")
        _IGNORED = frozenset("!！?？。、,.・~〜～ー… \u3000")

        @classmethod
        def _looks_nonverbal(cls, text: str) -> bool:
            body = (text or "").strip()
            if not body:
                return False
            if all(ch in cls._NOTES or ch in cls._IGNORED for ch in body):
                return True
            body = body.lower().strip().lstrip("[](){}<>").rstrip("[](){}<>").strip()
            if not body:
                return False
            if any(w in body for w in cls._WORDS):
                return True
            core = "".join(ch for ch in body if ch not in cls._IGNORED)
            return bool(core) and len(core) <= 6 and all(ch in cls._VOCAL for ch in core)

        def should_filter(self, avg_logprob: float, duration: float, text: str):
            if not self.enabled:
                return False, None, None
            thr = self.threshold
            if thr is not None and self.margin > 0 and duration <= self.short_window:
                thr = thr - self.margin
            if thr is not None and avg_logprob < thr:
                return True, "logprob", thr
            if self.drop_nonverbal and self._looks_nonverbal(text):
                return True, "nonverbal", thr
            return False, None, thr

    def should_force_full_transcribe(vad_segments, audio_duration, min_duration_for_fallback=120.0,
                                     min_coverage_ratio=0.01) -> bool:
        if audio_duration <= 0 or audio_duration < min_duration_for_fallback:
            return False
        flat = [s for g in (vad_segments or []) for s in (g or [])]
        if not flat:
            return True
        speech = sum(max(0.0, float(s.get("end_sec", 0.0)) - float(s.get("start_sec", 0.0))) for s in flat)
        if speech / audio_duration < min_coverage_ratio:
            return True
        return len(flat) <= 2 and audio_duration >= 4 * min_duration_for_fallback


class _NullTracer:
    """utils/parameter_tracer.py NullTracer: every emit_* is a no-op."""

    def __getattr__(self, name):
        if name.startswith("emit"):
            return lambda *a, **k: None
        raise AttributeError(name)


def read_audio(path: Union[str, Path]):
    """float32 mono samples + rate; soundfile when present, stdlib ``wave`` for the PCM16 scene files."""
    try:
        import soundfile as sf
        data, sr = sf.read(str(path), dtype="float32")
        if data.ndim > 1:
            data = np.mean(data, axis=1)
        return data, sr
    except ImportError:
        with wave.open(str(path), "rb") as wf:
            if wf.getsampwidth() != 2:
                raise ValueError("without soundfile only PCM16 WAV files can be read")
            sr, ch = wf.getframerate(), wf.getnchannels()
            pcm = np.frombuffer(wf.readframes(wf.getnframes()), dtype="<i2").astype(np.float32) / 32768.0
        if ch > 1:
            pcm = pcm.reshape(-1, ch).mean(axis=1)
        return pcm, sr


def _srt_time(td: datetime.timedelta) -> str:
    """``srt.timedelta_to_srt_timestamp``: milliseconds are the FLOOR of the timedelta's microseconds."""
    hrs, rem = divmod(td.seconds, 3600)
    hrs += td.days * 24
    mins, secs = divmod(rem, 60)
    return "%02d:%02d:%02d,%03d" % (hrs, mins, secs, td.microseconds // 1000)


def compose_srt(segments: List[Dict[str, Any]]) -> str:
    """``srt.compose`` of the reference's ``transcribe_to_srt`` (faster_whisper_pro_asr.py:1044-1057); without the
    ``srt`` package the same rules are applied here: sort by (start, end, index), skip subtitles without content, with
    a negative start or with start >= end, re-index from 1, collapse blank lines inside the content."""
    try:
        import srt
        make, compose = srt.Subtitle, srt.compose
    except (ImportError, AttributeError):
        make = None
    if make is not None:
        subs = [make(index=i, start=datetime.timedelta(seconds=s["start"]),
                     end=datetime.timedelta(seconds=s["end"]), content=s["text"])
                for i, s in enumerate(segments, 1)]
        return compose(subs)
    else:
        import re
        rows = sorted(((datetime.timedelta(seconds=s["start"]), datetime.timedelta(seconds=s["end"]), i, s["text"])
                       for i, s in enumerate(segments, 1)), key=lambda r: r[:3])
        out, idx = [], 1
        for start, end, _, content in rows:
            if not content.strip() or start < datetime.timedelta(0) or start >= end:
                continue
            if content[0] == "\n" or "\n\n" in content:
                content = re.sub(r"\n\n+", "\n", content.strip("\n"))
            out.append(f"{idx}\n{_srt_time(start)} --> {_srt_time(end)}\n{content}\n\n")
            idx += 1
        return "".join(out)


class HipFasterWhisperProASR:
    """Speech segmenter + batched Whisper transcription of one scene file on the MI355X."""

    def __init__(self, model_config: Dict, params: Dict, task: str, tracer=None, *, whisper_model=None,
                 segmenter=None):
        self.tracer = tracer if tracer is not None else _NullTracer()
        self.model_name = model_config.get("model_name", "large-v2")
        self.device = model_config.get("device", "cuda")
        self.compute_type = model_config.get("compute_type", "auto")
        decoder_params = params["decoder"]
        vad_params = params.get("vad", {}) or {}
        provider_params = params.get("provider", {}) or {}
        seg_cfg = params.get("speech_segmenter", {}) or {}
        backend = seg_cfg.get("backend", "silero-v6.2-hip")
        if not backend.startswith("silero"):
            vad_params = {}
        self.vad_threshold = vad_params.get("threshold", 0.28)
        self.min_speech_duration_ms = vad_params.get("min_speech_duration_ms", 100)
        # faster-whisper VadOptions dict the reference hands to transcribe() although vad_filter stays False
        # (faster_whisper_pro_asr.py:319-338, :410-411): carried through for call compatibility
        keys = ("threshold", "neg_threshold", "min_speech_duration_ms", "max_speech_duration_s", "min_silence_duration_ms",
                "speech_pad_ms")
        self._vad_parameters = {k: vad_params[k] for k in keys if vad_params.get(k) is not None} or None
        merged = {**vad_params, **seg_cfg} if backend.startswith("silero") else dict(seg_cfg)
        merged.pop("backend", None)
        if segmenter is not None:
            self._external_segmenter = segmenter
        else:
            try:
                self._external_segmenter = self._create_segmenter(backend, merged)
            except Exception as e:
                logger.error(f"Failed to create Speech Segmenter '{backend}': {e}")
                raise ValueError(f"Speech Segmenter not configured - this is an architecture violation: {e}")
        self.whisper_params: Dict[str, Any] = {}
        self.whisper_params.update(decoder_params)
        self.whisper_params.update(provider_params)
        raw_thr = self.whisper_params.get("logprob_threshold", -1.0)
        self.logprob_threshold = float(raw_thr) if raw_thr is not None else None
        self.logprob_margin = float(self.whisper_params.get("logprob_margin", 0.0) or 0.0)
        self.drop_nonverbal_vocals = bool(self.whisper_params.get("drop_nonverbal_vocals", False))
        enabled = self.whisper_params.get("post_model_filter_enabled")
        self.post_model_filter_enabled = False if enabled is None else bool(enabled)
        self._segment_filter = SegmentFilterHelper(SegmentFilterConfig(
            enabled=self.post_model_filter_enabled, logprob_threshold=self.logprob_threshold,
            logprob_margin=self.logprob_margin, drop_nonverbal_vocals=self.drop_nonverbal_vocals))
        for key in ("logprob_margin", "drop_nonverbal_vocals", "post_model_filter_enabled"):
            self.whisper_params.pop(key, None)
        self.whisper_params["task"] = task
        self.task = task
        self.suppress_low = ["Thank you", "視聴", "Thanks for"]
        self.suppress_high = ["視聴ありがとうございました", "ご視聴ありがとうございました", "字幕作成者", "提供", "スポンサー"]
        self._reset_runtime_statistics()
        self._last_full_results: List[Dict] = []
        self._primed: List[str] = []               # scene paths announced by prime_scenes()
        self._scene_cache: Dict[str, Dict[str, Any]] = {}
        self.pooled_calls = 0                      # diagnostics: engine calls made for primed scene lists
        if whisper_model is not None:
            self.whisper_model = whisper_model
        else:
            from .whisper_model import HipWhisperModel
            self.whisper_model = HipWhisperModel(self.model_name, device="cuda", compute_type=self.compute_type)

    @staticmethod
    def _create_segmenter(backend: str, config: Dict[str, Any]):
        from . import segmenters
        try:  # inside WhisperJAV: go through the reference's factory (registry entry in INTEGRATION.md)
            from whisperjav.modules.speech_segmentation import SpeechSegmenterFactory  # type: ignore
            return SpeechSegmenterFactory.create(backend, config=config)
        except ImportError:
            # stand-alone (no whisperjav package): the reference's registry names map onto the HIP back ends
            # (factory.py:17-33); "none" keeps the reference's semantics -- no gating, the whole scene is transcribed
            # (backends/none.py:51-92) -- and anything unknown is an error, as in SpeechSegmenterFactory.create
            if backend in ("none", ""):
                return segmenters.NullSpeechSegmenter()
            alias = {"silero": "silero-v4.0-hip", "silero-v4.0": "silero-v4.0-hip", "silero-v3.1": "silero-v3.1-hip",
                     "silero-v6.2": "silero-v6.2-hip", "ten": "ten-hip"}
            cls_path = segmenters.REGISTRY_ENTRIES.get(alias.get(backend, backend))
            if cls_path is None:
                raise ValueError(f"Unknown speech segmenter backend: {backend!r}. Available: "
                                 f"{sorted(set(segmenters.REGISTRY_ENTRIES) | {'none'})}")
            cls = getattr(segmenters, cls_path.rsplit(".", 1)[1])
            if "v3.1" in backend or "v4.0" in backend:
                config = dict(config, version="v3.1" if "v3.1" in backend else "v4.0")
            if not backend.startswith("silero"):        # silero-only keys of the resolver's VAD presets
                import inspect
                accepted = set(inspect.signature(cls.__init__).parameters)
                config = {k: v for k, v in config.items() if k in accepted}
            seg = cls(**config)
            if getattr(seg, "can_score", True) is False:
                # the v3.1 / v4.0 class used to be constructed here and to refuse only at its first segment() call --
                # after the model load and the scene split.  Fail before any audio is processed, and say what works.
                from .hipbind import WjError
                raise WjError(f"speech segmenter {backend!r}: the Silero {seg.version} network has no HIP kernel and is never replaced "
                              "silently.  Options: 'silero-v6.2-hip' (the v5/v6 network on the device, its own thresholds); "
                              f"HipSileroSpeechSegmenter(version={seg.version!r}, scorer='torch.hub') when the torch.hub archive is on "
                              "this box (the reference's own network scores on the host); or network='v6' to run the v6 scorer behind "
                              f"the {seg.version} call contract knowingly")
            return seg

    # ---- statistics hooks used by the pipelines -------------------------------------------------
    def _reset_runtime_statistics(self) -> None:
        self._filter_statistics = {"logprob_filtered": 0, "nonverbal_filtered": 0}
        # every candidate segment BEFORE the suppress lists and the log-prob / nonverbal gate: count and order-independent digest
        # (a regression check that still works when the gate drops everything, as it does on synthetic weights in fidelity mode)
        self._pregate = {"segments": 0, "digest": 0}
        self._last_vad_segments: List[Dict] = []
        self._vad_segments_per_scene: List[List[Dict]] = []

    def reset_statistics(self) -> None:
        self._reset_runtime_statistics()

    def get_filter_statistics(self) -> Dict[str, int]:
        return dict(self._filter_statistics)

    def get_last_vad_segments(self) -> List[Dict]:
        return list(self._last_vad_segments)

    # ---- parameters -------------------------------------------------------------------------------
    def _prepare_whisper_params(self) -> Dict[str, Any]:
        """Same normalisation as the reference (faster_whisper_pro_asr.py:340-436)."""
        p = dict(self.whisper_params)
        if "logprob_threshold" in p:
            p["log_prob_threshold"] = p.pop("logprob_threshold")
        if "suppress_tokens" in p:
            tok = p["suppress_tokens"]
            if isinstance(tok, tuple):
                p["suppress_tokens"] = list(tok)
            elif isinstance(tok, int):
                p["suppress_tokens"] = [tok]
            elif tok is not None and not isinstance(tok, list):
                del p["suppress_tokens"]
        if "no_repeat_ngram_size" in p and p["no_repeat_ngram_size"] is not None:
            try:
                p["no_repeat_ngram_size"] = int(p["no_repeat_ngram_size"])
            except (ValueError, TypeError):
                del p["no_repeat_ngram_size"]
        if isinstance(p.get("temperature"), list):
            t = p["temperature"]
            p["temperature"] = tuple(t) if len(t) > 1 else t[0]
        for key in ("fp16", "verbose", "vad", "vad_threshold", "hallucination_silence_threshold"):
            p.pop(key, None)
        p.setdefault("log_progress", False)
        p["vad_filter"] = p.get("vad_filter", False)
        if self._vad_parameters and "vad_parameters" not in p:
            p["vad_parameters"] = self._vad_parameters
        return {k: v for k, v in p.items() if v is not None}

    # ---- transcription ----------------------------------------------------------------------------
    def transcribe(self, audio_path: Union[str, Path], **kwargs) -> Dict:
        """One scene file (the reference's call contract, faster_whisper_pro_asr.py:438-557).  A path announced through
        ``prime_scenes`` is answered from the pooled pass over the whole announced list."""
        self._last_full_results = []
        audio_path = Path(audio_path)
        self._apply_runtime_task(kwargs)
        key = str(audio_path)
        if key not in self._scene_cache and key in self._primed:
            # an unreadable announced scene must fail on ITS OWN call, not on the first scene's: skip it here
            batch, loaded = [], []
            for p in self._primed:
                if p in self._scene_cache or not Path(p).exists():
                    continue
                try:
                    loaded.append(read_audio(Path(p)))
                    batch.append(p)
                except Exception as e:
                    logger.error(f"announced scene {p} is unreadable ({type(e).__name__}: {e}); left to its own call")
            self._primed = []
            try:
                results = self._transcribe_loaded(loaded) if key in batch else None
            except Exception as e:      # e.g. the segmenter failing on another scene: fall through to the per-scene path
                logger.error(f"pooled pass over {len(batch)} announced scenes failed ({type(e).__name__}: {e}); "
                             "continuing scene by scene")
                results = None
            if results is not None:
                self.pooled_calls += 1
                for p, res, vad in zip(batch, results, self._vad_segments_per_scene):
                    self._scene_cache[p] = {"result": res, "vad": vad}
        hit = self._scene_cache.pop(key, None)
        if hit is not None:
            self._last_vad_segments = hit["vad"]
            return hit["result"]
        return self._transcribe_loaded([read_audio(audio_path)])[0]

    def prime_scenes(self, scene_paths) -> None:
        """Announce the scene files the pipeline is about to hand over one by one (see the module docstring)."""
        self._primed = [str(Path(p)) for p in scene_paths]
        self._scene_cache = {}

    def transcribe_scenes(self, scenes, **kwargs) -> List[Dict]:
        """Pooled form: ``scenes`` = scene file paths or ``(audio float32, sample_rate)`` pairs -> the per-scene result
        dicts ``transcribe`` would return for each, computed with one segmentation launch and one pooled engine call.
        ``get_vad_segments_per_scene()`` holds every scene's VAD segments afterwards."""
        self._apply_runtime_task(kwargs)
        loaded = [s if isinstance(s, tuple) else read_audio(Path(s)) for s in scenes]
        return self._transcribe_loaded(loaded)

    def get_vad_segments_per_scene(self) -> List[List[Dict]]:
        return [list(v) for v in self._vad_segments_per_scene]

    def _apply_runtime_task(self, kwargs: Dict[str, Any]) -> None:
        if "task" in kwargs:
            runtime_task = kwargs.pop("task")
            if runtime_task != self.task:
                self.whisper_params["task"] = runtime_task
                self.task = runtime_task

    def _segment_all(self, loaded):
        seg = self._external_segmenter
        if len(loaded) > 1 and hasattr(seg, "segment_many"):
            return seg.segment_many([a for a, _ in loaded], [sr for _, sr in loaded])
        return [seg.segment(a, sample_rate=sr) for a, sr in loaded]

    def _transcribe_loaded(self, loaded) -> List[Dict]:
        language = self.whisper_params.get("language", "ja")
        results: List[Optional[Dict]] = [None] * len(loaded)
        self._vad_segments_per_scene = []
        clips: List[np.ndarray] = []
        owner: List[tuple] = []                       # (scene index, group start in seconds)
        for i, ((audio, sr), seg_result) in enumerate(zip(loaded, self._segment_all(loaded))):
            vad_groups = seg_result.to_legacy_format()
            duration = len(audio) / sr if sr else 0.0
            vad = [{"start_sec": round(s["start_sec"], 3), "end_sec": round(s["end_sec"], 3)} for g in vad_groups for s in g]
            self._vad_segments_per_scene.append(vad)
            self._last_vad_segments = vad
            if not vad_groups:
                if self._external_segmenter.name == "none" or should_force_full_transcribe(vad_groups, duration):
                    spans = [(0.0, duration)]
                else:
                    results[i] = {"segments": [], "text": "", "language": language}
                    continue
            elif should_force_full_transcribe(vad_groups, duration):
                spans = [(0.0, duration)]
            else:
                spans = [(g[0]["start_sec"], g[-1]["end_sec"]) for g in vad_groups if g]
            n_before = len(clips)
            for start, end in spans:
                clip = audio[int(start * sr): int(end * sr)]
                if len(clip) > 200:
                    clips.append(clip)
                    owner.append((i, start))
                    self.tracer.emit_transcribe_params(
                        params=self.whisper_params,
                        audio_info={"duration_sec": end - start, "start_sec": start, "end_sec": end, "sample_rate": sr,
                                    "shape": str(clip.shape), "dtype": str(clip.dtype)},
                        context=("full_audio" if len(spans) == 1 and start == 0.0 and end == duration
                                 else f"vad_group_{start:.2f}s-{end:.2f}s"))
            if len(clips) == n_before:
                results[i] = {"segments": [], "text": "", "language": language}
        per_scene: List[List[Dict]] = [[] for _ in loaded]
        if clips:
            for (i, start), segs in zip(owner, self._run_model(clips)):
                per_scene[i].extend(self._filter_group(segs, start))
        for i in range(len(loaded)):
            if results[i] is None:
                segs = per_scene[i]
                results[i] = {"segments": segs, "text": " ".join(s["text"] for s in segs), "language": language}
        return results  # type: ignore[return-value]

    def _run_model(self, clips: List[np.ndarray]):
        """Every clip through the engine in one call.  On an engine error the pooled list is BISECTED with the full
        parameters (one bad clip or a transient out-of-memory must not change the transcript of the whole recording);
        only a clip that still fails alone goes down the reference's retry ladder, which is per VAD group
        (faster_whisper_pro_asr.py:925-980): that clip with minimal parameters, then an empty result."""
        params = self._prepare_whisper_params()
        out: List[Any] = [None] * len(clips)

        def run(lo: int, hi: int) -> None:
            try:
                per_clip, _ = self.whisper_model.transcribe_many(clips[lo:hi], **params)
                out[lo:hi] = list(per_clip)
                return
            except Exception as e:
                if hi - lo > 1:
                    logger.error(f"pooled transcription of {hi - lo} clips failed ({type(e).__name__}: {e}); "
                                 "retrying the two halves with the same parameters")
                    mid = (lo + hi) // 2
                    run(lo, mid)
                    run(mid, hi)
                    return
                logger.error(f"transcription of one clip failed ({type(e).__name__}: {e}); retrying it with minimal parameters")
            minimal = {"task": self.whisper_params.get("task", "transcribe"),
                       "language": self.whisper_params.get("language", "ja"), "temperature": 0.0, "beam_size": 3,
                       "log_progress": False}
            try:
                segs, _ = self.whisper_model.transcribe_many(clips[lo:hi], **minimal)
                out[lo] = segs[0]
            except Exception as e2:
                logger.error(f"minimal-parameter retry failed too ({type(e2).__name__}: {e2}); clip dropped")
                out[lo] = []

        run(0, len(clips))
        return out

    def _filter_group(self, segs, start_sec: float) -> List[Dict]:
        out = []
        for seg in segs:
            text = (seg.text or "").strip()
            if not text:
                continue
            pg = getattr(self, "_pregate", None)
            if pg is not None:
                import zlib
                pg["segments"] += 1
                pg["digest"] = (pg["digest"] + zlib.crc32(f"{seg.start + start_sec:.2f},{seg.end + start_sec:.2f},{text}".encode())) % (1 << 32)
            if any(s in text for s in self.suppress_high):
                continue
            avg_lp = seg.avg_logprob
            for word in self.suppress_low:
                if word in text:
                    avg_lp -= 0.15
            drop, reason, _ = self._segment_filter.should_filter(
                avg_logprob=avg_lp, duration=max(0.0, float(seg.end - seg.start)), text=text)
            if drop:
                key = "logprob_filtered" if reason == "logprob" else "nonverbal_filtered"
                self._filter_statistics[key] += 1
                continue
            out.append({"start": seg.start + start_sec, "end": seg.end + start_sec, "text": text, "avg_logprob": avg_lp})
        return out

    def transcribe_to_srt(self, audio_path: Union[str, Path], output_srt_path: Union[str, Path], **kwargs) -> Path:
        output_srt_path = Path(output_srt_path)
        result = self.transcribe(Path(audio_path), **kwargs)
        output_srt_path.parent.mkdir(parents=True, exist_ok=True)
        with open(output_srt_path, "w", encoding="utf-8") as f:
            f.write(compose_srt(result.get("segments", [])))
        return output_srt_path

    def cleanup(self) -> None:
        """Idempotent; the reference may skip it entirely (os._exit, main.py:2490-2494)."""
        model, self.whisper_model = getattr(self, "whisper_model", None), None
        if model is not None and hasattr(model, "close"):
            model.close()
        seg = getattr(self, "_external_segmenter", None)
        if seg is not None and hasattr(seg, "cleanup"):
            seg.cleanup()


class HipWhisperProASR(HipFasterWhisperProASR):
    """``WhisperProASR``-shaped module for the fidelity pipeline
    (/root/reference/whisperjav/modules/whisper_pro_asr.py:31-576): openai-whisper call contract
    (``model.transcribe(chunk, **params) -> dict``), post-model log-prob gate ON by default (:123-127),
    ``fp16`` / ``verbose`` accepted."""

    def __init__(self, model_config: Dict, params: Dict, task: str, tracer=None, *, whisper_model=None, segmenter=None):
        params = dict(params)
        decoder = dict(params.get("decoder", {}))
        if decoder.get("post_model_filter_enabled") is None and \
                (params.get("provider") or {}).get("post_model_filter_enabled") is None:
            decoder["post_model_filter_enabled"] = True
        params["decoder"] = decoder
        if whisper_model is None:
            from .whisper_model import HipOpenAIWhisperModel
            whisper_model = HipOpenAIWhisperModel(model_config.get("model_name", "large-v2"), device="cuda",
                                                  compute_type="float16" if decoder.get("fp16", True) else "float32")
        super().__init__(model_config, params, task, tracer, whisper_model=whisper_model, segmenter=segmenter)

    def _prepare_whisper_params(self) -> Dict[str, Any]:
        # whisper_pro_asr.py:201-218: the tuner's parameters go through as they are (None values included --
        # whisper.transcribe accepts them), the temperature list becomes a tuple, verbose defaults to None
        p = dict(self.whisper_params)
        if isinstance(p.get("temperature"), list):
            p["temperature"] = tuple(p["temperature"])
        p.setdefault("verbose", None)
        return p

    def transcribe(self, audio_path, **kwargs):
        # the dict-returning model is adapted to the (segments, info) batch interface of the base class
        model = self.whisper_model
        if not hasattr(model, "transcribe_many"):
            class _Adapter:
                def __init__(self, inner):
                    self.inner = inner

                def transcribe_many(self, clips, **params):
                    from .whisper_model import Segment
                    out = []
                    for c in clips:
                        res = self.inner.transcribe(c, **params)
                        out.append([Segment(id=s.get("id", 0), seek=s.get("seek", 0), start=s["start"], end=s["end"],
                                            text=s["text"], tokens=s.get("tokens", []),
                                            avg_logprob=s.get("avg_logprob", 0.0),
                                            compression_ratio=s.get("compression_ratio", 0.0),
                                            no_speech_prob=s.get("no_speech_prob", 0.0)) for s in res["segments"]])
                    return out, [None] * len(clips)

                def close(self):
                    if hasattr(self.inner, "close"):
                        self.inner.close()
            self.whisper_model = _Adapter(model)
        return super().transcribe(audio_path, **kwargs)
