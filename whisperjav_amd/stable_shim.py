"""``stable_whisper.load_faster_whisper`` shaped entry over the HIP Whisper path (BASELINE cfg1: the reference's
``faster`` / ``fast`` modes reach Whisper through ``StableTSASR``, whisperjav/modules/stable_ts_asr.py).

What the reference does there (stable_ts_asr.py:296-305, 363-420, 477-509, 645-655):

    self.model = stable_whisper.load_faster_whisper(model_name, device=..., compute_type=...)
    result = self.model.transcribe(audio_numpy_or_path, **final_params)     # a stable_whisper.WhisperResult
    JapanesePostProcessor().process(result, ...)                            # stable-ts regrouping algebra on the result
    result.to_srt_vtt(path, word_level=False, segment_level=True, strip=True)

``final_params`` are the tuner's decoder + provider sections filtered by ``FASTER_WHISPER_PARAMS`` plus what stable-ts itself
consumes (``regroup``, ``vad``, ``vad_threshold``, ``batch_size``, ``verbose``).  The drop-in is therefore one symbol:
``load_faster_whisper`` below returns an object whose ``transcribe`` takes exactly those keywords, runs the clip through
:class:`whisperjav_amd.whisper_model.HipWhisperModel` (log-mel -> encoder -> search on the MI355X, word timestamps from the
device alignment pass -- stable-ts always asks faster-whisper for them) and hands back

* a real ``stable_whisper.WhisperResult`` built from the segment dictionaries when stable-ts is importable -- the regrouping
  (``regroup=`` here, ``JapanesePostProcessor`` afterwards) and the SRT writer then ARE the reference's own, unchanged;
* otherwise :class:`HipWhisperResult`: ``segments`` / ``to_dict`` / ``to_srt_vtt`` with the call contract the reference uses,
  so the pipeline still produces its SRT.  The reference's ``_postprocess`` catches and logs the missing regrouping methods
  (stable_ts_asr.py:623-643); nothing here re-implements stable-ts.

Not reproduced: stable-ts's silence suppression (``vad=True`` nudges word edges with its own Silero pass).  It is applied
through stable-ts when that is installed (``WhisperResult.adjust_by_silence``) and skipped, with one warning, when it is not.
Nothing in this module computes on the CPU: without the HIP extension the constructor of ``HipWhisperModel`` raises.
"""
from __future__ import annotations

import logging
import warnings
from pathlib import Path
from typing import Any, Dict, List, Optional, Union

import numpy as np

log = logging.getLogger("whisperjav_amd")

# keywords stable-ts consumes itself before calling faster-whisper (stable_whisper/whisper_word_level/faster_whisper.py),
# as far as the reference can send them (STANDARD / FASTER_WHISPER_PARAMS, stable_ts_asr.py:103-132, + its own additions)
_STABLE_TS_KEYS = ("regroup", "vad", "vad_threshold", "verbose", "batch_size", "suppress_silence", "suppress_word_ts",
                   "use_word_position", "q_levels", "k_size", "denoiser", "denoiser_options", "min_word_dur", "nonspeech_error",
                   "only_voice_freq", "check_sorted", "progress_callback", "nonspeech_skip", "min_silence_dur", "ignore_compatibility",
                   "extra_models", "stream", "only_ffmpeg", "dynamic_heads", "gap_padding", "time_scale", "demucs", "demucs_options")


# ... of which these act here (or are pure UI); every other key that is SET to a non-default value is warned about once
_HONOURED = frozenset(("regroup", "vad", "vad_threshold", "suppress_silence", "verbose", "batch_size", "progress_callback",
                       "check_sorted", "ignore_compatibility"))
# stable-ts's own defaults (transcribe_stable signature): passing them changes nothing there either
_STABLE_TS_DEFAULTS = {"suppress_word_ts": True, "use_word_position": True, "q_levels": 20, "k_size": 5, "min_word_dur": None,
                       "nonspeech_error": 0.1, "only_voice_freq": False, "nonspeech_skip": 5.0, "min_silence_dur": None,
                       "stream": None, "only_ffmpeg": False, "dynamic_heads": None, "gap_padding": " ...", "time_scale": None}


def _srt_time(t: float) -> str:
    ms = int(round(max(0.0, float(t)) * 1000.0))
    h, ms = divmod(ms, 3600000)
    m, ms = divmod(ms, 60000)
    s, ms = divmod(ms, 1000)
    return f"{h:02d}:{m:02d}:{s:02d},{ms:03d}"


class HipWord:
    __slots__ = ("word", "start", "end", "probability")

    def __init__(self, word: str, start: float, end: float, probability: float):
        self.word, self.start, self.end, self.probability = word, float(start), float(end), float(probability)

    def to_dict(self) -> Dict[str, Any]:
        return {"word": self.word, "start": self.start, "end": self.end, "probability": self.probability}


class HipSegment:
    """One segment with the attributes the reference reads from stable-ts segments (``text``, ``start``, ``end``, ``words``)."""

    def __init__(self, d: Dict[str, Any]):
        self.id = int(d.get("id", 0))
        self.seek = d.get("seek", 0)
        self.start, self.end, self.text = float(d["start"]), float(d["end"]), str(d["text"])
        self.tokens = list(d.get("tokens") or [])
        self.temperature = d.get("temperature")
        self.avg_logprob = d.get("avg_logprob")
        self.compression_ratio = d.get("compression_ratio")
        self.no_speech_prob = d.get("no_speech_prob")
        self.words: List[HipWord] = [HipWord(w["word"], w["start"], w["end"], w.get("probability", 0.0)) for w in (d.get("words") or [])]

    @property
    def has_words(self) -> bool:
        return bool(self.words)

    def to_dict(self) -> Dict[str, Any]:
        out = {"id": self.id, "seek": self.seek, "start": self.start, "end": self.end, "text": self.text, "tokens": self.tokens,
               "temperature": self.temperature, "avg_logprob": self.avg_logprob, "compression_ratio": self.compression_ratio,
               "no_speech_prob": self.no_speech_prob}
        if self.words:
            out["words"] = [w.to_dict() for w in self.words]
        return out


class HipWhisperResult:
    """The part of ``stable_whisper.WhisperResult`` the reference touches when stable-ts's regrouping is not available:
    ``segments``, ``text``, ``language``, ``to_dict()``, ``to_srt_vtt(path, segment_level=True, word_level=False, strip=True)``
    (stable_ts_asr.py:509, 577-587, 645-655)."""

    def __init__(self, result: Dict[str, Any]):
        self.language = result.get("language")
        self.segments: List[HipSegment] = [HipSegment(s) for s in result.get("segments", [])]
        self.ori_dict = result

    @property
    def text(self) -> str:
        return "".join(s.text for s in self.segments)

    def has_words(self) -> bool:
        return all(s.has_words for s in self.segments) if self.segments else False

    def to_dict(self) -> Dict[str, Any]:
        return {"text": self.text, "segments": [s.to_dict() for s in self.segments], "language": self.language}

    def to_srt_vtt(self, filepath: Optional[str] = None, segment_level: bool = True, word_level: bool = True, min_dur: float = 0.02,
                   tag=None, vtt: Optional[bool] = None, strip: bool = True, reverse_text: Union[bool, tuple] = False) -> str:
        """SubRip (or WebVTT when ``vtt`` / a ``.vtt`` path) at segment level.  Word-level karaoke tags are stable-ts's own
        rendering and need stable-ts: asking for them here raises instead of writing something else."""
        if word_level:
            raise NotImplementedError("word-level SRT/VTT rendering is stable-ts's; install stable-ts (the result is then a "
                                      "stable_whisper.WhisperResult) or call with word_level=False as the reference does")
        if not segment_level:
            raise ValueError("segment_level and word_level cannot both be False")
        if vtt is None:
            vtt = bool(filepath) and str(filepath).lower().endswith(".vtt")
        blocks = []
        n = 0
        for s in self.segments:
            text = s.text.strip() if strip else s.text
            if not text:
                continue
            start, end = s.start, max(s.end, s.start + min_dur)
            n += 1
            if vtt:
                blocks.append(f"{_srt_time(start).replace(',', '.')} --> {_srt_time(end).replace(',', '.')}\n{text}")
            else:
                blocks.append(f"{n}\n{_srt_time(start)} --> {_srt_time(end)}\n{text}")
        body = ("WEBVTT\n\n" if vtt else "") + "\n\n".join(blocks) + ("\n" if blocks else "")
        if filepath:
            path = Path(str(filepath))
            if not path.suffix:
                path = path.with_suffix(".vtt" if vtt else ".srt")
            path.parent.mkdir(parents=True, exist_ok=True)
            path.write_text(body, encoding="utf-8")
        return body

    to_srt = to_srt_vtt


class HipStableWhisperModel:
    """What ``stable_whisper.load_faster_whisper`` returns, over the HIP path: ``transcribe(audio, **params)`` with the keywords
    ``StableTSASR._prepare_transcribe_parameters`` produces in turbo mode (stable_ts_asr.py:363-420)."""

    def __init__(self, model_size_or_path: str, device: str = "cuda", compute_type: str = "float16", *, model: Any = None,
                 **model_kwargs: Any):
        if model is None:
            from .whisper_model import HipWhisperModel
            model = HipWhisperModel(model_size_or_path, device=device, compute_type=compute_type, **model_kwargs)
        self.model = model
        self.model_size_or_path = model_size_or_path
        self._warned: set = set()

    def _warn_once(self, key: str, msg: str) -> None:
        if key not in self._warned:
            self._warned.add(key)
            log.warning(msg)

    @staticmethod
    def _load(audio: Union[str, Path, np.ndarray]) -> np.ndarray:
        if isinstance(audio, (str, Path)):
            from .asr import read_audio
            data, sr = read_audio(audio)
            if sr != 16000:
                raise ValueError(f"{audio}: {sr} Hz; the reference hands 16 kHz scene files (or arrays it resampled) to stable-ts")
            return np.asarray(data, dtype=np.float32)
        if hasattr(audio, "detach"):            # a torch tensor: stable-ts accepts them
            audio = audio.detach().cpu().numpy()
        return np.asarray(audio, dtype=np.float32).reshape(-1)

    def transcribe(self, audio: Union[str, Path, np.ndarray], **params: Any):
        from .whisper_model import _KNOWN
        p = dict(params)
        stable = {k: p.pop(k) for k in _STABLE_TS_KEYS if k in p}
        for k, v in stable.items():     # stable-ts features that are not reproduced: say so once instead of changing behaviour silently
            if k not in _HONOURED and v not in (None, False, {}, [], ()) and v != _STABLE_TS_DEFAULTS.get(k, None):
                self._warn_once("ignored:" + k, f"transcribe({k}={v!r}): a stable-ts option this shim does not implement; it has no effect here")
        # names stable-ts (and faster-whisper's older signatures) accept for the same thing
        if "logprob_threshold" in p:
            p["log_prob_threshold"] = p.pop("logprob_threshold")
        unknown = set(p) - _KNOWN
        if unknown:      # the reference's own retry path keys on TypeError (stable_ts_asr.py:517-548)
            raise TypeError(f"transcribe() got unexpected keyword argument(s): {sorted(unknown)}")
        temp = p.get("temperature")
        if isinstance(temp, (list, tuple)) and len(temp) == 1:
            p["temperature"] = float(temp[0])
        p["word_timestamps"] = True              # stable-ts always asks for word timings: its regrouping works on words
        p.setdefault("vad_filter", False)
        samples = self._load(audio)
        segments, info = self.model.transcribe(samples, **p)
        seg_dicts = []
        for i, s in enumerate(segments):
            d = {"id": i, "seek": s.seek, "start": s.start, "end": s.end, "text": s.text, "tokens": list(s.tokens),
                 "temperature": s.temperature, "avg_logprob": s.avg_logprob, "compression_ratio": s.compression_ratio,
                 "no_speech_prob": s.no_speech_prob}
            if s.words is not None:
                d["words"] = [{"word": w.word, "start": w.start, "end": w.end, "probability": w.probability} for w in s.words]
            seg_dicts.append(d)
        result_dict = {"language": getattr(info, "language", p.get("language")), "segments": seg_dicts}
        return self._wrap(result_dict, samples, stable)

    def _wrap(self, result_dict: Dict[str, Any], samples: np.ndarray, stable: Dict[str, Any]):
        regroup = stable.get("regroup", True)
        want_silence = bool(stable.get("vad", False)) or bool(stable.get("suppress_silence", False))
        try:
            import stable_whisper
            WhisperResult = stable_whisper.WhisperResult
        except Exception:
            WhisperResult = None
        if WhisperResult is None or WhisperResult is object:
            if regroup not in (False, None):
                self._warn_once("regroup", "stable-ts is not importable: segments are returned as decoded (no regrouping); "
                                           "the reference's JapanesePostProcessor needs stable-ts as well")
            if want_silence:
                self._warn_once("silence", "stable-ts is not importable: its silence suppression (vad=True) is skipped")
            return HipWhisperResult(result_dict)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", UserWarning)
            result = WhisperResult(result_dict)
            if want_silence and hasattr(result, "adjust_by_silence"):
                try:
                    result.adjust_by_silence(samples, vad=bool(stable.get("vad", False)),
                                             vad_threshold=float(stable.get("vad_threshold", 0.35)))
                except Exception as e:      # stable-ts loads Silero through torch.hub: offline boxes end here
                    self._warn_once("silence", f"stable-ts silence suppression skipped: {e}")
            if regroup not in (False, None) and hasattr(result, "regroup"):
                result.regroup(regroup)
        return result

    # stable-ts leaves faster-whisper's own method reachable under this name
    def transcribe_original(self, audio, **params):
        return self.model.transcribe(self._load(audio), **params)


def load_faster_whisper(model_size_or_path: str, **model_init_options: Any) -> HipStableWhisperModel:
    """``stable_whisper.load_faster_whisper(model_size_or_path, **model_init_options)`` (called at stable_ts_asr.py:301-305
    with ``device=`` and ``compute_type=``).  ``compute_type`` int8 -- the reference's default for this backend -- has no
    counterpart here; HipWhisperModel maps it to float16 and says so."""
    return HipStableWhisperModel(model_size_or_path, **model_init_options)
