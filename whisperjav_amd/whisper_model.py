"""``faster_whisper.WhisperModel``-shaped shim over the HIP engine.

WhisperJAV enters the upstream model through exactly two calls
(/root/reference/whisperjav/modules/faster_whisper_pro_asr.py):

    WhisperModel(model_size_or_path, device, compute_type, cpu_threads, num_workers)      # :247-253
    segments, info = model.transcribe(audio_float32_16k, **params)                        # :819-822

with ``params`` produced by ``_prepare_whisper_params`` (:340-436).  ``HipWhisperModel`` accepts
the same arguments and yields ``Segment`` objects with the fields the reference consumes
(``start, end, text, avg_logprob, words``; ``dataclasses.asdict`` works, :883-887).

The long-form logic below restates faster-whisper 1.2.1 ``transcribe.py`` (``generate_segments``,
``generate_with_fallback``, ``get_prompt``, ``_split_segments_by_timestamps``), which is not vendored
in the reference: 30 s windows advanced by the last timestamp token, no-speech / log-prob gates,
segments cut at consecutive timestamp pairs.  Beyond the drop-in call, ``transcribe_many`` runs the
SAME per-clip procedure for many clips at once (one encoder batch + one batched decode per round),
which is how the MI355X is kept busy: every VAD group of a scene (or file) becomes one row.

``word_timestamps=True`` runs the device alignment pass (``wj_whisper_align``) window by window while the
cross K/V are resident, then faster-whisper's word heuristics; ``hallucination_silence_threshold`` is not implemented.
"""
from __future__ import annotations

import json
import logging
import os
import time
import zlib
from dataclasses import dataclass, field
from typing import Any, Dict, Iterable, Iterator, List, Optional, Sequence, Tuple, Union

import numpy as np

from . import dims as pdims

logger = logging.getLogger("whisperjav")

SAMPLE_RATE = 16000
HOP = 160
N_FRAMES = 3000
TIME_PRECISION = 0.02
INPUT_STRIDE = 2           # mel frames per encoder position
FRAMES_PER_SECOND = 100


@dataclass
class Word:
    start: float
    end: float
    word: str
    probability: float


@dataclass
class Segment:
    id: int
    seek: int
    start: float
    end: float
    text: str
    tokens: List[int]
    avg_logprob: float
    compression_ratio: float
    no_speech_prob: float
    words: Optional[List[Word]] = None
    temperature: Optional[float] = None


@dataclass
class TranscriptionInfo:
    language: str
    language_probability: float
    duration: float
    duration_after_vad: float
    all_language_probs: Optional[List[Tuple[str, float]]] = None
    transcription_options: Dict[str, Any] = field(default_factory=dict)
    vad_options: Optional[Dict[str, Any]] = None


class IdTokenizer:
    """Stand-in used when no ``tokenizer.json`` is available (synthetic weights): renders token ids."""

    real = False

    def decode(self, tokens: Sequence[int]) -> str:
        return "".join(f"<{t}>" for t in tokens)

    def encode(self, text: str) -> List[int]:
        raise ValueError("this model was loaded without a tokenizer.json; text prompts cannot be encoded")

    def non_speech_tokens(self) -> List[int]:
        return []

    def split_to_word_tokens(self, tokens: Sequence[int], language: str = "ja"):
        """Without a vocabulary every token renders as one ``<id>`` "word"."""
        return [f"<{t}>" for t in tokens], [[int(t)] for t in tokens]


def _bytes_to_unicode() -> Dict[int, str]:
    """The byte <-> printable-character table of byte-level BPE vocabularies (GPT-2's ``bytes_to_unicode``, which Whisper's
    multilingual vocabulary uses): printable latin-1 bytes map to themselves, the rest to code points from 256 up."""
    keep = list(range(ord("!"), ord("~") + 1)) + list(range(0xA1, 0xAC + 1)) + list(range(0xAE, 0xFF + 1))
    table, extra = {}, 0
    for b in range(256):
        if b in keep:
            table[b] = chr(b)
        else:
            table[b] = chr(256 + extra)
            extra += 1
    return table


class VocabTokenizer:
    """Decode-only tokenizer over a CTranslate2 directory's ``vocabulary.json`` / ``vocabulary.txt`` (the token strings of the
    byte-level BPE vocabulary, one per id) for model directories that ship no ``tokenizer.json``: transcripts come out as TEXT
    (faster-whisper would fetch ``openai/whisper-tiny``'s tokenizer from the hub in that case; nothing is downloaded here).
    Encoding needs the merge table, which the vocabulary file does not hold: ``initial_prompt`` / ``hotwords`` / ``prefix`` as
    text raise with that explanation (token-id lists are accepted where the shim's callers may pass them)."""

    real = True

    def __init__(self, tokens: Sequence[str], non_speech: Optional[Sequence[int]] = None):
        self._tokens = list(tokens)
        self._byte_of = {c: b for b, c in _bytes_to_unicode().items()}
        self._ids = {t: i for i, t in enumerate(self._tokens)}
        self._non_speech = sorted(set(int(t) for t in non_speech)) if non_speech else None

    @classmethod
    def from_directory(cls, path: str, non_speech: Optional[Sequence[int]] = None) -> Optional["VocabTokenizer"]:
        import json
        js, txt = os.path.join(path, "vocabulary.json"), os.path.join(path, "vocabulary.txt")
        if os.path.exists(js):
            with open(js, encoding="utf-8") as f:
                return cls(json.load(f), non_speech)
        if os.path.exists(txt):
            with open(txt, encoding="utf-8") as f:
                return cls([line.rstrip("\n") for line in f], non_speech)
        return None

    def _token_bytes(self, t: int) -> bytes:
        tok = self._tokens[t] if 0 <= t < len(self._tokens) else ""
        if tok.startswith("<|") and tok.endswith("|>"):
            return tok.encode("utf-8")                       # special tokens render as written, like tokenizers does
        out = bytearray()
        for ch in tok:
            b = self._byte_of.get(ch)
            out.extend(bytes([b]) if b is not None else ch.encode("utf-8"))
        return bytes(out)

    def decode(self, tokens: Sequence[int]) -> str:
        return b"".join(self._token_bytes(int(t)) for t in tokens).decode("utf-8", errors="replace")

    def encode(self, text: str) -> List[int]:
        raise ValueError("this model directory has a vocabulary file but no tokenizer.json (no BPE merge table): text prompts cannot be "
                         "encoded; copy tokenizer.json of the matching openai/whisper-* repository into the directory")

    def token_to_id(self, token: str) -> Optional[int]:
        return self._ids.get(token)

    def non_speech_tokens(self) -> List[int]:
        """``suppress_tokens=[-1]``: the model's own list when the directory's ``config.json`` carries one (``suppress_ids``: what
        the converter copied from the checkpoint's generation config = whisper's ``non_speech_tokens``), else the symbols of
        whisper/tokenizer.py ``non_speech_tokens`` that ARE single vocabulary entries (no merge table is needed to know that)."""
        if self._non_speech is not None:
            return list(self._non_speech)
        table = _bytes_to_unicode()
        enc = lambda t: "".join(table[b] for b in t.encode("utf-8"))      # noqa: E731
        symbols = list('"#()*+/:;<=>@[\\]^_`{|}~「」『』')
        symbols += "<< >> <<< >>> -- --- -( -[ (' (\" (( )) ((( ))) [[ ]] {{ }} ♪♪ ♪♪♪".split()
        out = set()
        for sym in symbols + [" -", " '"]:
            for cand in (sym, " " + sym) if not sym.startswith(" ") else (sym,):
                i = self._ids.get(enc(cand))
                if i is not None:
                    out.add(i)
        return sorted(out)

    split_tokens_on_unicode = None      # bound below (shared with HfTokenizer: they only need decode())
    split_to_word_tokens = None


_NO_SPACE_LANGUAGES = {"zh", "ja", "th", "lo", "my", "yue"}


class HfTokenizer:
    """``tokenizer.json`` of a CTranslate2 / HF Whisper model directory via the ``tokenizers`` package."""

    real = True

    def __init__(self, path: str):
        import tokenizers
        self._tok = tokenizers.Tokenizer.from_file(path)

    def decode(self, tokens: Sequence[int]) -> str:
        return self._tok.decode(list(tokens), skip_special_tokens=False)

    def encode(self, text: str) -> List[int]:
        return self._tok.encode(text, add_special_tokens=False).ids

    def split_tokens_on_unicode(self, tokens: Sequence[int]):
        """whisper/tokenizer.py ``split_tokens_on_unicode``: a word closes when its tokens decode to valid unicode."""
        full = self.decode(tokens)
        bad = "\ufffd"
        words, word_tokens, current, offset = [], [], [], 0
        for t in tokens:
            current.append(int(t))
            dec = self.decode(current)
            idx = dec.find(bad)
            if idx < 0 or (offset + idx < len(full) and full[offset + idx] == bad):
                words.append(dec)
                word_tokens.append(current)
                current = []
                offset += len(dec)
        return words, word_tokens

    def split_to_word_tokens(self, tokens: Sequence[int], language: str = "ja"):
        """whisper/tokenizer.py ``split_to_word_tokens`` (``tokens`` end with eot)."""
        import string
        words, word_tokens = self.split_tokens_on_unicode(tokens)
        if language in _NO_SPACE_LANGUAGES:
            return words, word_tokens
        eot = self._tok.token_to_id("<|endoftext|>")
        out_w, out_t = [], []
        for sub, sub_t in zip(words, word_tokens):
            special = eot is not None and sub_t[0] >= eot
            if special or sub.startswith(" ") or sub.strip() in string.punctuation or not out_w:
                out_w.append(sub)
                out_t.append(list(sub_t))
            else:
                out_w[-1] += sub
                out_t[-1].extend(sub_t)
        return out_w, out_t

    def non_speech_tokens(self) -> List[int]:
        symbols = list('"#()*+/:;<=>@[\\]^_`{|}~「」『』')
        symbols += "<< >> <<< >>> -- --- -( -[ (' (\" (( )) ((( ))) [[ ]] {{ }} ♪♪ ♪♪♪".split()
        miscellaneous = set("♩♪♫♬♭♮♯")
        result = {self.encode(" -")[0], self.encode(" '")[0]}
        for symbol in symbols + list(miscellaneous):
            for toks in (self.encode(symbol), self.encode(" " + symbol)):
                if len(toks) == 1 or symbol in miscellaneous:
                    result.add(toks[0])
        return sorted(result)


def compression_ratio(text: str) -> float:
    raw = text.encode("utf-8")
    return len(raw) / len(zlib.compress(raw)) if raw else 0.0


def _vocab_split_to_word_tokens(self, tokens: Sequence[int], language: str = "ja"):
    import string
    words, word_tokens = HfTokenizer.split_tokens_on_unicode(self, tokens)
    if language in _NO_SPACE_LANGUAGES:
        return words, word_tokens
    eot = self.token_to_id("<|endoftext|>")
    out_w, out_t = [], []
    for sub, sub_t in zip(words, word_tokens):
        special = eot is not None and sub_t[0] >= eot
        if special or sub.startswith(" ") or sub.strip() in string.punctuation or not out_w:
            out_w.append(sub)
            out_t.append(list(sub_t))
        else:
            out_w[-1] += sub
            out_t[-1].extend(sub_t)
    return out_w, out_t


VocabTokenizer.split_tokens_on_unicode = HfTokenizer.split_tokens_on_unicode
VocabTokenizer.split_to_word_tokens = _vocab_split_to_word_tokens


@dataclass
class TranscribeOptions:
    task: str = "transcribe"
    language: Optional[str] = "ja"
    beam_size: int = 5
    best_of: int = 5
    patience: float = 1.0
    length_penalty: float = 1.0
    repetition_penalty: float = 1.0
    no_repeat_ngram_size: int = 0
    temperature: Union[float, Sequence[float]] = (0.0, 0.2, 0.4, 0.6, 0.8, 1.0)
    compression_ratio_threshold: Optional[float] = 2.4
    log_prob_threshold: Optional[float] = -1.0
    no_speech_threshold: Optional[float] = 0.6
    condition_on_previous_text: bool = True
    prompt_reset_on_temperature: float = 0.5
    initial_prompt: Optional[Union[str, Iterable[int]]] = None
    prefix: Optional[str] = None
    suppress_blank: bool = True
    suppress_tokens: Optional[Sequence[int]] = (-1,)
    without_timestamps: bool = False
    max_initial_timestamp: Optional[float] = 1.0
    word_timestamps: bool = False
    max_new_tokens: Optional[int] = None
    hotwords: Optional[str] = None
    chunk_length: Optional[int] = None
    multilingual: bool = False
    log_progress: bool = False
    vad_filter: bool = False
    vad_parameters: Optional[dict] = None
    clip_timestamps: Union[str, List[float]] = "0"
    hallucination_silence_threshold: Optional[float] = None
    language_detection_threshold: Optional[float] = 0.5
    language_detection_segments: int = 1
    prepend_punctuations: Optional[str] = None
    append_punctuations: Optional[str] = None


_KNOWN = set(TranscribeOptions.__dataclass_fields__)


def split_segments_by_timestamps(tokens: Sequence[int], time_offset: float, segment_size: int,
                                 segment_duration: float, seek: int, timestamp_begin: int):
    """faster-whisper ``_split_segments_by_timestamps``: cut a window's tokens at consecutive timestamp
    pairs; return (segments, new_seek, single_timestamp_ending)."""
    tokens = list(tokens)
    out = []
    single_ending = len(tokens) >= 2 and tokens[-2] < timestamp_begin <= tokens[-1]
    pairs = [i for i in range(1, len(tokens)) if tokens[i] >= timestamp_begin and tokens[i - 1] >= timestamp_begin]
    if pairs:
        cuts = list(pairs)
        if single_ending:
            cuts.append(len(tokens))
        last = 0
        for cut in cuts:
            piece = tokens[last:cut]
            t0 = piece[0] - timestamp_begin
            t1 = piece[-1] - timestamp_begin
            out.append({"seek": seek, "start": time_offset + t0 * TIME_PRECISION,
                        "end": time_offset + t1 * TIME_PRECISION, "tokens": piece})
            last = cut
        if single_ending:
            seek += segment_size
        else:
            seek += (tokens[last - 1] - timestamp_begin) * INPUT_STRIDE
    else:
        duration = segment_duration
        stamps = [t for t in tokens if t >= timestamp_begin]
        if stamps and stamps[-1] != timestamp_begin:
            duration = (stamps[-1] - timestamp_begin) * TIME_PRECISION
        out.append({"seek": seek, "start": time_offset, "end": time_offset + duration, "tokens": tokens})
        seek += segment_size
    return out, seek, single_ending


@dataclass
class _ClipState:
    index: int
    n_frames_content: int
    duration: float
    seek: int = 0
    all_tokens: List[int] = field(default_factory=list)
    prompt_reset_since: int = 0
    segments: List[Segment] = field(default_factory=list)
    next_id: int = 0
    last_speech: float = 0.0
    language: str = "ja"
    language_probability: float = 1.0
    all_language_probs: Optional[List[Tuple[str, float]]] = None

    @property
    def active(self) -> bool:
        return self.seek < self.n_frames_content


class HipWhisperModel:
    """Drop-in for ``faster_whisper.WhisperModel`` (constructor + ``transcribe``)."""

    MEL_MODE = "fw"      # faster-whisper feature semantics
    FLAVOR = "fw"
    # faster-whisper moves ``seek`` to the end of the last aligned word (transcribe.py generate_segments).  On random
    # weights the alignment is noise and that re-seek multiplies the windows, so bench.py switches it off for its
    # word-timestamp figure (and says so); everything else keeps upstream's behaviour
    word_reseek = True
    bucket_by_length = True      # transcribe_many: batches are formed from windows of similar content length
    # Measured and NOT the default (profiles/r03_bench_overlap_*.json): encoding the next chunk on a second stream while the
    # current one decodes -- 8.66 s vs 8.70 s for the sequential 768-window schedule (the 256-tile GEMM workgroups take a
    # CU's whole register file and LDS: the two streams alternate instead of sharing the chip); on disjoint compute units
    # (encoder_cus > 0, CU-masked streams) it is slower -- 9.9 / 13.0 / 22.5 s with 144 / 176 / 208 CUs for the encoder: both
    # sides scale linearly with their CU count (the cross-attention's 6 TB/s needs all 256 CUs issuing), so no split wins.
    overlap_encode = False       # the next chunk is encoded on a second stream while the current one decodes
    encoder_cus = 0              # > 0: ... on that many compute units, the decode loop on the others (wj_stream_create)
    _side_stream = None
    _overlapped = False
    decode_stats = None          # counters of the decode calls, see reset_decode_stats()

    def __init__(self, model_size_or_path: str = "large-v3", device: str = "cuda", device_index: int = 0,
                 compute_type: str = "float16", cpu_threads: int = 0, num_workers: int = 1,
                 weights: Optional[Dict[str, np.ndarray]] = None, dims: Optional[pdims.WhisperDims] = None,
                 max_batch: int = 32, max_beam: int = 5, blob=None, offsets=None, kv_len: Optional[int] = None,
                 enc_batch: Optional[int] = None, **_unused):
        from . import engine, weights as W
        if device not in ("cuda", "auto", "hip"):
            raise ValueError(f"HipWhisperModel runs on the MI355X only (device={device!r}); there is no CPU path")
        # ctranslate2 compute types -> engine arithmetic.  "float16" is the reference's own GPU arithmetic
        # (faster-whisper float16 / whisper fp16=True) and what "auto" resolves to on a GPU there
        # (whisperjav/config/resolver_v3.py:163-168); the int8 flavours have no MI355X counterpart here and run
        # in the 16-bit type they dequantise to.
        ct = {"auto": "float16", "default": "float16", "bfloat16": "bfloat16", "float16": "float16",
              "float32": "float32", "int8": "float16", "int8_float16": "float16", "int8_bfloat16": "bfloat16"}
        if compute_type not in ct:
            raise ValueError(f"unsupported compute_type {compute_type!r}")
        self.compute_type = ct[compute_type]
        self.tokenizer: Any = IdTokenizer()
        if blob is None and weights is None:
            dims, weights = self._load_checkpoint(model_size_or_path)
        elif dims is None:
            dims = pdims.dims_for(model_size_or_path)
        self.dims = dims
        self.model = engine.HipWhisper(dims, weights, blob=blob, offsets=offsets, dtype=self.compute_type,
                                       device=device_index, max_batch=max_batch, max_beam=max_beam, kv_len=kv_len,
                                       enc_batch=enc_batch)
        self.fe = engine.HipLogMel(dims.n_mels, self.MEL_MODE, device=device_index)
        self.tokens = self.model.tokens
        self.max_batch, self.max_beam = max_batch, max_beam
        # decoder positions a window may use: n_text_ctx, or the KV cache the engine was sized for (kv_len: a deployment
        # that caps max_new_tokens trades cache positions for resident windows, see engine.HipWhisper)
        self.max_length = self.model.kv_len
        self.n_text_ctx = dims.n_text_ctx
        self._warned = set()
        self.seed = 0               # base seed of the device sampler's counter-based generator
        self.device_beam = True     # beam search on the device (False: host-driven search.py over the step API)
        self._sample_calls = 0

    # ---- loading ---------------------------------------------------------------------------
    def _load_checkpoint(self, path: str):
        from . import weights as W
        if os.path.isdir(path):
            tok = os.path.join(path, "tokenizer.json")
            if os.path.exists(tok):
                self.tokenizer = HfTokenizer(tok)
            if os.path.exists(os.path.join(path, "model.bin")):       # a CTranslate2 conversion: what faster_whisper.WhisperModel opens
                from . import ct2_format                                # (faster_whisper_pro_asr.py:246-253)
                dims, sd, extras = ct2_format.load_ct2_whisper(path)
                if not os.path.exists(tok):          # no tokenizer.json: decode through the directory's vocabulary file
                    vt = VocabTokenizer.from_directory(path, extras.get("suppress_ids"))
                    if vt is not None and not all(t.startswith("<") and t.endswith(">") and t[1:-1].isdigit() for t in vt._tokens[:16]):
                        self.tokenizer = vt
                if extras.get("alignment_heads"):
                    self._alignment_heads = list(extras["alignment_heads"])
                self._ct2_extras = extras
                return dims, sd
            for name in ("model.pt", "whisper.pt"):
                cand = os.path.join(path, name)
                if os.path.exists(cand):
                    return W.load_openai_checkpoint(cand)
            if os.path.exists(os.path.join(path, "config.json")) and (
                    os.path.exists(os.path.join(path, "model.safetensors"))
                    or os.path.exists(os.path.join(path, "model.safetensors.index.json"))):
                dims, sd, extras = W.load_hf_checkpoint(path)      # a Hugging Face WhisperForConditionalGeneration directory
                if extras.get("alignment_heads"):
                    self._alignment_heads = list(extras["alignment_heads"])
                return dims, sd
            raise FileNotFoundError(f"{path} holds neither a CTranslate2 conversion (model.bin), an openai-format checkpoint "
                                    "(model.pt) nor a Hugging Face one (config.json + model.safetensors); see INTEGRATION.md")
        if os.path.isfile(path):
            return W.load_openai_checkpoint(path)
        from . import ct2_format
        cached = ct2_format.resolve_cached_model(path)      # "large-v3" -> Systran/faster-whisper-large-v3 in the local HF cache
        if cached is not None and os.path.exists(os.path.join(cached, "model.bin")):
            return self._load_checkpoint(cached)
        raise FileNotFoundError(
            f"model {path!r}: no local checkpoint found and this environment has no network; pass weights=/dims= "
            "(e.g. whisperjav_amd.weights.synth_weights) or a directory with model.pt + tokenizer.json")

    def close(self) -> None:
        self.model.close()

    def _warn_once(self, key: str, msg: str) -> None:
        if key not in self._warned:
            self._warned.add(key)
            logger.warning(msg)

    # ---- option plumbing -------------------------------------------------------------------
    def _options(self, kw: Dict[str, Any]) -> TranscribeOptions:
        unknown = set(kw) - _KNOWN
        if unknown:
            raise TypeError(f"transcribe() got unexpected keyword argument(s): {sorted(unknown)}")
        o = TranscribeOptions(**kw)
        if o.length_penalty is None and self.FLAVOR == "fw":
            o.length_penalty = 1.0
        if o.prepend_punctuations is None:
            o.prepend_punctuations = "\"'“¿([{-"
        if o.append_punctuations is None:
            o.append_punctuations = "\"'.。,，!！?？:：”)]}、"
        if o.word_timestamps and o.hallucination_silence_threshold is not None:
            self._warn_once("hst", "hallucination_silence_threshold is not implemented on the HIP path; ignored")
        if o.vad_filter:
            raise ValueError("vad_filter=True is not supported: WhisperJAV runs its own speech segmenter "
                             "(faster_whisper_pro_asr.py passes vad_filter=False)")
        return o

    def _suppressed(self, o: TranscribeOptions) -> Tuple[int, ...]:
        t = self.tokens
        raw = o.suppress_tokens
        if isinstance(raw, str):          # whisper.DecodingOptions: "-1" or a comma-separated id list
            raw = [int(x) for x in raw.split(",") if x.strip()]
        sup = list(raw or [])
        if -1 in sup:
            sup = [x for x in sup if x >= 0] + list(self.tokenizer.non_speech_tokens())
        sup += [t.transcribe, t.translate, t.sot, t.sot_prev, t.sot_lm]
        if self.FLAVOR == "ow":
            sup.append(t.no_speech)      # whisper/decoding.py _get_suppress_tokens adds <|nospeech|> as well
        return tuple(sorted(set(sup)))

    def _prompt(self, o: TranscribeOptions, previous: Sequence[int], first_window: bool,
                language: Optional[str] = None) -> List[int]:
        t = self.tokens
        prompt: List[int] = []
        hot = o.hotwords if (o.hotwords and not o.prefix) else None
        if previous or hot:
            prompt.append(t.sot_prev)
            if hot:
                ht = self.tokenizer.encode(" " + hot.strip())
                prompt.extend(ht[: self._n_text_ctx() // 2 - 1])
            if previous:
                prompt.extend(list(previous)[-(self._n_text_ctx() // 2 - 1):])
        lang = language or o.language or "ja"
        prompt.extend([t.sot, t.language_token(pdims.language_index(lang)),
                       t.transcribe if o.task == "transcribe" else t.translate])
        if o.without_timestamps:
            prompt.append(t.no_timestamps)
        if o.prefix and first_window:
            pt = self.tokenizer.encode(" " + o.prefix.strip())
            if not o.without_timestamps:
                prompt.append(t.timestamp_begin)
            prompt.extend(pt[: self._n_text_ctx() // 2 - 1])
        return prompt

    # ---- decoding of one batch of windows ------------------------------------------------------
    def _max_new(self, o: TranscribeOptions, P: int) -> int:
        if self.FLAVOR == "ow":
            max_new = self._n_text_ctx() // 2                      # DecodingOptions.sample_len default
            if o.max_new_tokens is not None:
                max_new = min(max_new, int(o.max_new_tokens))
        else:
            max_new = (self.max_length - P) if o.max_new_tokens is None else int(o.max_new_tokens)
        return max(1, min(max_new, self.max_length - P))

    def _rank_score(self, sum_lp: float, n: int, o: TranscribeOptions) -> float:
        """Hypothesis ranking among ``best_of`` samples: ctranslate2 normalises the cumulative log-prob by
        ``len ** length_penalty``; whisper's MaximumLikelihoodRanker by the length (or the GNMT penalty)."""
        n = max(n, 1)
        lp = o.length_penalty
        if self.FLAVOR == "ow":
            return sum_lp / (n if lp is None else ((5 + n) / 6) ** float(lp))
        return sum_lp / (n ** float(1.0 if lp is None else lp))

    def _n_text_ctx(self) -> int:
        """The model's text context (prompt truncation, whisper's sample_len default) -- not the KV positions of this engine."""
        return int(getattr(self, "n_text_ctx", None) or self.max_length)

    def reset_decode_stats(self) -> None:
        """Counters over the decode calls since the last reset (bench.py reports them: realised tokens per window,
        decode iterations actually run -- searches end at EOT -- against the iterations allowed)."""
        self.decode_stats = {"calls": 0, "windows": 0, "tokens": 0, "max_tokens": 0, "steps_run": 0, "steps_allowed": 0,
                             "window_steps_run": 0, "at_length_limit": 0}
        # wall time of the phases of transcribe_many as the HOST sees them (seconds; the encoder call returns before its kernels end,
        # the decode and alignment calls return with their results): where a step goes outside the kernels
        self.phase_s = {"features": 0.0, "encode_call": 0.0, "decode": 0.0, "finish_host": 0.0, "align": 0.0}

    def _phase(self, name: str, t0: float) -> None:
        ph = getattr(self, "phase_s", None)
        if ph is not None:
            ph[name] = ph.get(name, 0.0) + (time.perf_counter() - t0)

    def _count_decode(self, decoded, max_new: int) -> None:
        st = getattr(self, "decode_stats", None)
        if st is None:
            return
        lens = [len(t) for t, _, _ in decoded]
        st["calls"] += 1
        st["windows"] += len(lens)
        st["tokens"] += int(sum(lens))
        st["max_tokens"] = max(st["max_tokens"], max(lens) if lens else 0)
        st["at_length_limit"] += sum(1 for n in lens if n >= max_new)
        try:
            info = self.model.last_decode_info()
            st["steps_run"] += info["steps"]
            st["steps_allowed"] += info["max_new_tokens"]
            st["window_steps_run"] += info["window_steps"]
            st["compactions"] = st.get("compactions", 0) + info["compactions"]
        except Exception:
            pass

    def _decode_once(self, prompts: List[List[int]], slots: List[int], temperature: float, o: TranscribeOptions,
                     suppress: Tuple[int, ...]) -> List[Tuple[List[int], float, float]]:
        """One rung of the ladder for the resident windows ``slots``: (tokens, avg_logprob, no_speech_prob) each."""
        out = self._decode_once_impl(prompts, slots, temperature, o, suppress)
        self._count_decode(out, self._max_new(o, len(prompts[0])))
        return out

    def _decode_once_impl(self, prompts: List[List[int]], slots: List[int], temperature: float, o: TranscribeOptions,
                          suppress: Tuple[int, ...]) -> List[Tuple[List[int], float, float]]:
        from . import engine, search
        P = len(prompts[0])
        max_new = self._max_new(o, P)
        mit = None if o.max_initial_timestamp is None else int(round(float(o.max_initial_timestamp) / TIME_PRECISION))
        mit_s = None if mit is None else mit * TIME_PRECISION        # None: any first timestamp (whisper's default for the option)
        n = len(prompts)
        if temperature > 0:
            # sampling rung: beam_size -> 1, best_of hypotheses per window drawn on the device
            best_of = max(1, int(o.best_of or 1))
            if best_of == 7 or best_of > 8:
                raise ValueError(f"best_of={best_of}: the device sampler groups 1..6 or 8 samples per window")
            do = engine.DecodeOptions(max_new_tokens=max_new, suppress_blank=o.suppress_blank,
                                      without_timestamps=o.without_timestamps, suppress_tokens=suppress,
                                      max_initial_timestamp=mit_s,
                                      repetition_penalty=float(o.repetition_penalty if self.FLAVOR == "fw" else 1.0),
                                      no_repeat_ngram_size=int(o.no_repeat_ngram_size if self.FLAVOR == "fw" else 0))
            cap = max(1, (self.max_batch * self.max_beam) // best_of)
            out: List[Tuple[List[int], float, float]] = []
            for c0 in range(0, n, cap):
                sub = slice(c0, min(n, c0 + cap))
                self._sample_calls += 1
                res = self.model.decode_sample(np.array(prompts[sub], dtype=np.int32), do, temperature=temperature,
                                               best_of=best_of, slots=slots[sub],
                                               seed=(self.seed + 0x9E3779B1 * self._sample_calls) & 0xFFFFFFFF)
                for w in range(sub.stop - sub.start):
                    rows = range(w * best_of, (w + 1) * best_of)
                    r = max(rows, key=lambda q: self._rank_score(float(res.sum_logprob[q]), int(res.n_tokens[q]), o))
                    toks = res.tokens[r, : res.n_tokens[r]].tolist()
                    out.append((toks, float(res.sum_logprob[r]) / (len(toks) + 1), float(res.no_speech_prob[r])))
            return out
        identity = list(slots) == list(range(n))
        beam = int(o.beam_size or 1)
        device_loop = beam == 1 and (self.FLAVOR == "fw" or (float(o.repetition_penalty) == 1.0
                                                             and int(o.no_repeat_ngram_size) == 0))
        n_fin = round(beam * float(o.patience or 1.0))       # finished hypotheses a window collects before it stops
        device_beam = (not device_loop and self.device_beam and beam in (2, 3, 4, 5, 6, 8) and n_fin + beam <= 24
                       and (self.FLAVOR == "fw" or n_fin >= beam))
        if not identity and not (device_loop or device_beam):
            # host-driven search over the step API addresses windows 0..n-1 only: decode the resident prefix, keep ours
            hi = max(slots) + 1
            full = self._decode_once_impl([prompts[slots.index(i)] if i in slots else prompts[0] for i in range(hi)],
                                          list(range(hi)), temperature, o, suppress)
            return [full[i] for i in slots]
        out = []
        if device_beam:
            # beam search, device resident: CTranslate2's rules (wj_whisper_decode_beam) for the faster-whisper flavour,
            # openai-whisper's BeamSearchDecoder (wj_whisper_decode_beam_openai) for the fidelity flavour; the host-driven
            # restatements in search.py stay available (device_beam = False) and are what the GPU tests cross-check with
            lp = o.length_penalty
            ow = self.FLAVOR == "ow"
            res = self.model.decode_beam(
                np.array(prompts, dtype=np.int32),
                engine.DecodeOptions(max_new_tokens=max_new, suppress_blank=o.suppress_blank,
                                     without_timestamps=o.without_timestamps, suppress_tokens=suppress,
                                     max_initial_timestamp=mit_s,
                                     repetition_penalty=1.0 if ow else float(o.repetition_penalty),
                                     no_repeat_ngram_size=0 if ow else int(o.no_repeat_ngram_size)),
                beam_size=beam, patience=float(o.patience or 1.0),
                length_penalty=(None if lp is None else float(lp)) if ow else (1.0 if lp is None else float(lp)),
                slots=None if identity else slots, flavor="openai" if ow else "ct2")
            for r in range(n):
                toks = res.tokens[r, : res.n_tokens[r]].tolist()
                out.append((toks, float(res.sum_logprob[r]) / (len(toks) + 1), float(res.no_speech_prob[r])))
        elif device_loop:
            do = engine.DecodeOptions(max_new_tokens=max_new, suppress_blank=o.suppress_blank,
                                      without_timestamps=o.without_timestamps, suppress_tokens=suppress,
                                      max_initial_timestamp=mit_s,
                                      repetition_penalty=float(o.repetition_penalty),
                                      no_repeat_ngram_size=int(o.no_repeat_ngram_size))
            if identity:
                res = self.model.decode_greedy(np.array(prompts, dtype=np.int32), do)
            else:
                res = self.model.decode_sample(np.array(prompts, dtype=np.int32), do, temperature=0.0, best_of=1, slots=slots)
            for r in range(n):
                toks = res.tokens[r, : res.n_tokens[r]].tolist()
                out.append((toks, float(res.sum_logprob[r]) / (len(toks) + 1), float(res.no_speech_prob[r])))
        else:
            lp = o.length_penalty
            so = search.SearchOptions(beam_size=beam, patience=float(o.patience or 1.0),
                                      length_penalty=(-1.0 if lp is None else float(lp)),
                                      repetition_penalty=float(o.repetition_penalty),
                                      no_repeat_ngram_size=int(o.no_repeat_ngram_size), suppress_blank=o.suppress_blank,
                                      suppress_tokens=suppress, without_timestamps=o.without_timestamps,
                                      max_initial_timestamp_index=mit, max_new_tokens=max_new)
            fn = search.beam_search_openai if self.FLAVOR == "ow" else search.beam_search
            results = fn(search.HipStepScorer(self.model, so), prompts, so, eot=self.tokens.eot,
                         timestamp_begin=self.tokens.timestamp_begin, n_text_ctx=self.max_length)
            for wr in results:
                out.append((wr.sequences[0], wr.avg_logprob(0), wr.no_speech_prob))
        return out

    def _decode_windows(self, prompts: List[List[int]], o: TranscribeOptions, suppress: Tuple[int, ...],
                        slots: Optional[List[int]] = None):
        """Temperature-fallback ladder over the resident windows (faster-whisper ``generate_with_fallback`` /
        whisper ``decode_with_fallback``): every rung re-decodes only the windows that still need a fallback.
        Returns per window (tokens, avg_logprob, no_speech_prob, temperature, compression_ratio)."""
        temps = [float(x) for x in (o.temperature if isinstance(o.temperature, (list, tuple)) else [o.temperature])]
        n = len(prompts)
        final: List[Any] = [None] * n
        tried: List[List[Any]] = [[] for _ in range(n)]        # every rung's result
        below_cr: List[List[Any]] = [[] for _ in range(n)]     # ... those under the compression-ratio threshold
        pending = list(range(n))
        where = list(range(n)) if slots is None else list(slots)      # resident window of each prompt row
        for T in temps:
            if not pending:
                break
            decoded = self._decode_once([prompts[i] for i in pending], [where[i] for i in pending], T, o, suppress)
            still = []
            for i, (toks, avg_lp, nsp) in zip(pending, decoded):
                text = self.tokenizer.decode([t for t in toks if t < self.tokens.eot]).strip()
                cr = compression_ratio(text)
                cand = (toks, avg_lp, nsp, T, cr)
                tried[i].append(cand)
                needs_fallback = False
                if o.compression_ratio_threshold is not None and cr > o.compression_ratio_threshold:
                    needs_fallback = True                                   # too repetitive
                else:
                    below_cr[i].append(cand)
                if o.log_prob_threshold is not None and avg_lp < o.log_prob_threshold:
                    needs_fallback = True                                   # average log probability too low
                if (o.no_speech_threshold is not None and nsp > o.no_speech_threshold
                        and o.log_prob_threshold is not None and avg_lp < o.log_prob_threshold):
                    needs_fallback = False                                  # silence
                if needs_fallback:
                    still.append(i)
                else:
                    final[i] = cand
            pending = still
        for i in pending:       # every rung failed
            if self.FLAVOR == "ow":
                final[i] = tried[i][-1]                                     # whisper keeps the last rung's result
            else:
                best = max(below_cr[i] or tried[i], key=lambda c: c[1])     # faster-whisper: best average log-prob
                final[i] = (best[0], best[1], best[2], temps[-1], best[4])  # last temperature drives the prompt reset
        return final

    # ---- word timestamps ---------------------------------------------------------------------------
    def alignment_heads(self) -> List[Tuple[int, int]]:
        """(layer, head) pairs whose cross-attention tracks time.  A model directory may carry them in
        ``config.json`` ("alignment_heads", CTranslate2 converter); else the published table entry of the model, else
        whisper's default (every head of the upper half of the decoder)."""
        if getattr(self, "_alignment_heads", None):
            return self._alignment_heads
        L, H = self.dims.n_text_layer, self.dims.n_text_head
        if (L, H, self.dims.n_mels) == (32, 20, 128):     # large-v3 (generation_config.json of openai/whisper-large-v3)
            return [(7, 0), (10, 17), (12, 18), (13, 12), (16, 1), (17, 14), (19, 11), (21, 4), (24, 1), (25, 6)]
        return [(l, h) for l in range(L // 2, L) for h in range(H)]

    def _find_alignment(self, windows: List[Tuple[int, List[int], int, str]], task: str = "transcribe") -> List[List[dict]]:
        """``windows`` = (resident slot, text tokens, content frames, language) -> per window the word dicts of
        faster-whisper ``find_alignment`` ({word, tokens, start, end, probability}; times relative to the window)."""
        t = self.tokens
        task_tok = t.transcribe if task == "transcribe" else t.translate
        rows, keep = [], []
        for i, (_, text_tokens, _, lang) in enumerate(windows):
            if text_tokens:
                rows.append([t.sot, t.language_token(pdims.language_index(lang)), task_tok, t.no_timestamps, *text_tokens, t.eot])
                keep.append(i)
        out: List[List[dict]] = [[] for _ in windows]
        if not rows:
            return out
        t_ph = time.perf_counter()
        res = self.model.align(rows, 4, self.alignment_heads(), [windows[i][2] for i in keep],
                               slots=[windows[i][0] for i in keep])
        self._phase("align", t_ph)               # (also inside "finish_host": the reader subtracts it)
        for i, (text_idx, time_idx, probs) in zip(keep, res):
            text_tokens = windows[i][1]
            words, word_tokens = self.tokenizer.split_to_word_tokens(list(text_tokens) + [t.eot], windows[i][3])
            if len(word_tokens) <= 1 or len(text_idx) == 0:
                continue
            bounds = np.pad(np.cumsum([len(w) for w in word_tokens[:-1]]), (1, 0))
            jumps = np.pad(np.diff(text_idx), (1, 0), constant_values=1).astype(bool)
            jump_times = time_idx[jumps] / (SAMPLE_RATE / HOP / INPUT_STRIDE)      # tokens_per_second = 50
            starts, ends = jump_times[bounds[:-1]], jump_times[bounds[1:]]
            # np.mean(probs[a:b]) per word, all words of one length at once (72 k tiny np.mean calls were 0.35 s of host time per 120-min
            # step): rows of equal length gathered into a matrix and averaged along the contiguous axis -- bit-identical to the
            # per-slice call (checked on random cuts; np.add.reduceat is NOT: it sums in another order)
            lens = np.diff(bounds)
            p32 = np.ascontiguousarray(probs, dtype=np.float32)
            word_p = np.full(len(lens), np.nan, dtype=np.float32)
            if len(p32) and bounds[-1] <= len(p32):
                for ln in np.unique(lens[lens > 0]):
                    idx = np.nonzero(lens == ln)[0]
                    word_p[idx] = np.mean(p32[bounds[idx][:, None] + np.arange(ln)[None, :]], axis=1)
            else:
                word_p = np.array([np.mean(p32[a:b]) for a, b in zip(bounds[:-1], bounds[1:])], dtype=np.float32)
            out[i] = [dict(word=w, tokens=wt, start=float(s0), end=float(e0), probability=float(pw))
                      for w, wt, s0, e0, pw in zip(words, word_tokens, starts, ends, word_p)]
        return out

    @staticmethod
    def _merge_punctuations(alignment: List[dict], prepended: str, appended: str) -> None:
        i, j = len(alignment) - 2, len(alignment) - 1
        while i >= 0:
            prev, foll = alignment[i], alignment[j]
            if prev["word"].startswith(" ") and prev["word"].strip() in prepended:
                foll["word"] = prev["word"] + foll["word"]
                foll["tokens"] = prev["tokens"] + foll["tokens"]
                prev["word"], prev["tokens"] = "", []
            else:
                j = i
            i -= 1
        i, j = 0, 1
        while j < len(alignment):
            prev, foll = alignment[i], alignment[j]
            if not prev["word"].endswith(" ") and foll["word"] in appended:
                prev["word"] = prev["word"] + foll["word"]
                prev["tokens"] = prev["tokens"] + foll["tokens"]
                foll["word"], foll["tokens"] = "", []
            else:
                i = j
            j += 1

    def _add_word_timestamps(self, windows: List[dict], o: TranscribeOptions) -> None:
        """faster-whisper ``add_word_timestamps`` (whisper/timing.py's, which it copies) for a batch of windows.
        Each entry: {"slot", "pieces" (the window's sub-segments), "frames", "seek", "last_speech"}; the pieces
        gain "words" and their start/end move to the word boundaries; "last_speech" is updated."""
        eot = self.tokens.eot
        per_piece = [[[tk for tk in pc["tokens"] if tk < eot] for pc in w["pieces"]] for w in windows]
        flat = [[tk for pc in pieces for tk in pc] for pieces in per_piece]
        alignments = self._find_alignment([(w["slot"], flat[i], w["frames"], w["st"].language) for i, w in enumerate(windows)],
                                          o.task)
        for w, alignment, piece_tokens in zip(windows, alignments, per_piece):
            durations = np.array([a["end"] - a["start"] for a in alignment])
            durations = durations[durations.nonzero()]
            median_duration = min(0.7, float(np.median(durations))) if len(durations) else 0.0
            max_duration = median_duration * 2
            if len(durations):      # truncate long words at sentence boundaries
                marks = ".。!！?？"
                for i in range(1, len(alignment)):
                    if alignment[i]["end"] - alignment[i]["start"] > max_duration:
                        if alignment[i]["word"] in marks:
                            alignment[i]["end"] = alignment[i]["start"] + max_duration
                        elif alignment[i - 1]["word"] in marks:
                            alignment[i]["start"] = alignment[i]["end"] - max_duration
            self._merge_punctuations(alignment, o.prepend_punctuations, o.append_punctuations)
            time_offset = w["seek"] / FRAMES_PER_SECOND
            last_speech = w["last_speech"]
            word_index = 0
            for pc, toks in zip(w["pieces"], piece_tokens):
                saved, words = 0, []
                while word_index < len(alignment) and saved < len(toks):
                    timing = alignment[word_index]
                    if timing["word"]:
                        words.append(dict(word=timing["word"], start=round(time_offset + timing["start"], 2),
                                          end=round(time_offset + timing["end"], 2), probability=timing["probability"]))
                    saved += len(timing["tokens"])
                    word_index += 1
                if words:
                    # the first (and second) word after a pause may not be longer than twice the median duration
                    if words[0]["end"] - last_speech > median_duration * 4 and (
                            words[0]["end"] - words[0]["start"] > max_duration
                            or (len(words) > 1 and words[1]["end"] - words[0]["start"] > max_duration * 2)):
                        if len(words) > 1 and words[1]["end"] - words[1]["start"] > max_duration:
                            boundary = max(words[1]["end"] / 2, words[1]["end"] - max_duration)
                            words[0]["end"] = words[1]["start"] = boundary
                        words[0]["start"] = max(0, words[0]["end"] - max_duration)
                    # prefer the segment-level start / end if the first / last word is too long
                    if pc["start"] < words[0]["end"] and pc["start"] - 0.5 > words[0]["start"]:
                        words[0]["start"] = max(0, min(words[0]["end"] - median_duration, pc["start"]))
                    else:
                        pc["start"] = words[0]["start"]
                    if pc["end"] > words[-1]["start"] and pc["end"] + 0.5 < words[-1]["end"]:
                        words[-1]["end"] = max(words[-1]["start"] + median_duration, pc["end"])
                    else:
                        pc["end"] = words[-1]["end"]
                    last_speech = pc["end"]
                pc["words"] = words
            w["last_speech"] = last_speech

    def _finish_windows(self, o: TranscribeOptions, batch: List["_ClipState"], sizes: List[int], decoded, slots: List[int],
                        tb: int) -> None:
        """Gates, timestamp slicing, word alignment (while the windows' cross K/V are resident at ``slots``), seek
        update and segment emission for one decoded group of windows (the body of ``generate_segments``)."""
        work = []
        for st, size, dec, slot in zip(batch, sizes, decoded, slots):
            toks, avg_lp, nsp, temp, cr = dec
            time_offset = st.seek * HOP / SAMPLE_RATE
            seg_duration = size * HOP / SAMPLE_RATE
            if o.no_speech_threshold is not None:
                skip = nsp > o.no_speech_threshold
                if o.log_prob_threshold is not None and avg_lp > o.log_prob_threshold:
                    skip = False
                if skip:
                    st.seek += size
                    continue
            prev_seek = st.seek
            pieces, st.seek, single_ending = split_segments_by_timestamps(toks, time_offset, size, seg_duration, st.seek, tb)
            work.append(dict(st=st, dec=dec, pieces=pieces, prev_seek=prev_seek, single_ending=single_ending,
                             time_offset=time_offset, slot=slot, frames=size, seek=prev_seek, last_speech=st.last_speech))
        if o.word_timestamps and work:
            self._add_word_timestamps(work, o)
            for w in work:
                st = w["st"]
                ends = [wd["end"] for pc in w["pieces"] for wd in pc.get("words", [])]
                last_word_end = ends[-1] if ends else None
                if self.word_reseek and not w["single_ending"] and last_word_end is not None and last_word_end > w["time_offset"]:
                    st.seek = round(last_word_end * FRAMES_PER_SECOND)
                if last_word_end is not None:
                    st.last_speech = last_word_end
        for w in work:
            st = w["st"]
            _, avg_lp, nsp, temp, cr = w["dec"]
            for piece in w["pieces"]:
                text = self.tokenizer.decode([t for t in piece["tokens"] if t < self.tokens.eot])
                words = [Word(start=x["start"], end=x["end"], word=x["word"], probability=x["probability"])
                         for x in piece["words"]] if "words" in piece else None
                if piece["start"] == piece["end"] or not text.strip():
                    if self.FLAVOR == "ow":   # whisper.transcribe keeps the slot with cleared text/tokens
                        st.next_id += 1
                        st.segments.append(Segment(id=st.next_id, seek=w["prev_seek"], start=piece["start"],
                                                   end=piece["end"], text="", tokens=[], avg_logprob=avg_lp,
                                                   compression_ratio=cr, no_speech_prob=nsp, temperature=temp,
                                                   words=[] if words is not None else None))
                    continue
                st.all_tokens.extend(piece["tokens"])
                st.next_id += 1
                st.segments.append(Segment(id=st.next_id, seek=w["prev_seek"], start=piece["start"], end=piece["end"],
                                           text=text, tokens=piece["tokens"], avg_logprob=avg_lp,
                                           compression_ratio=cr, no_speech_prob=nsp, words=words, temperature=temp))
            if not o.condition_on_previous_text or temp > o.prompt_reset_on_temperature:
                st.prompt_reset_since = len(st.all_tokens)

    def _detect_languages(self, states: List["_ClipState"], feats, threshold: float, n_segments: int) -> None:
        """``language=None``: faster-whisper's detection loop (whisper.transcribe uses the first 30 s only, i.e.
        ``n_segments`` = 1) -- per clip the language-token distribution of successive 30 s windows until one exceeds
        ``threshold``, else the majority vote of the per-window winners."""
        import torch
        codes = pdims.LANGUAGE_CODES[: self.tokens.num_languages]
        votes: List[List[Tuple[str, float]]] = [[] for _ in states]
        todo = list(range(len(states)))
        for seg in range(n_segments):
            todo = [i for i in todo if seg * N_FRAMES < states[i].n_frames_content]
            if not todo:
                break
            nxt = []
            for lo in range(0, len(todo), self.max_batch):
                part = todo[lo: lo + self.max_batch]
                mel = torch.zeros((len(part), self.dims.n_mels, N_FRAMES), dtype=torch.float32, device=feats.device)
                for j, i in enumerate(part):
                    size = min(N_FRAMES, states[i].n_frames_content - seg * N_FRAMES)
                    mel[j, :, :size] = feats[states[i].index, :, seg * N_FRAMES: seg * N_FRAMES + size]
                self.model.encode(mel)
                probs = self.model.language_probs(len(part))
                for j, i in enumerate(part):
                    order = np.argsort(-probs[j], kind="stable")
                    st = states[i]
                    st.all_language_probs = [(codes[k], float(probs[j, k])) for k in order]
                    top, p = st.all_language_probs[0]
                    votes[i].append((top, p))
                    if p > threshold:
                        st.language, st.language_probability = top, p
                    else:
                        nxt.append(i)
            todo = nxt
        for i in todo:      # no window was confident enough: majority vote over the windows seen
            seen: Dict[str, List[float]] = {}
            for name, p in votes[i]:
                seen.setdefault(name, []).append(p)
            best = max(seen, key=lambda name: len(seen[name]))          # first language wins ties (insertion order)
            states[i].language = best
            states[i].language_probability = max(seen[best])

    # ---- public API ------------------------------------------------------------------------------
    def transcribe(self, audio: np.ndarray, **kwargs) -> Tuple[Iterator[Segment], TranscriptionInfo]:
        """faster-whisper's call contract for ONE clip: ``(segment iterator, info)``."""
        segs, infos = self.transcribe_many([audio], **kwargs)
        return iter(segs[0]), infos[0]

    def transcribe_many(self, audios: Sequence[np.ndarray], **kwargs) -> Tuple[List[List[Segment]], List[TranscriptionInfo]]:
        """The same per-clip procedure for many clips, batched across clips window by window."""
        import torch
        o = self._options(dict(kwargs))
        suppress = self._suppressed(o)
        # numpy clips (the reference's call contract) or CUDA tensors (clips of a recording already resident in HBM)
        clips = [a.reshape(-1) if (hasattr(a, "is_cuda") and a.is_cuda) else np.ascontiguousarray(a, dtype=np.float32).reshape(-1)
                 for a in audios]
        if any(len(c) <= 200 for c in clips):
            raise ValueError("every clip must be longer than 200 samples (12.5 ms)")
        # features of every clip in ONE launch; frame axis padded so any 3000-frame window can be sliced
        frames = [self.fe.frames(len(c)) for c in clips]
        width = max(frames) + N_FRAMES
        t_ph = time.perf_counter()
        feats = self.fe(clips, out_frames=width)                   # [n, n_mels, width], zero padded (pad_or_trim)
        self._phase("features", t_ph)
        tail = 1 if self.FLAVOR == "fw" else N_FRAMES    # fw: features[:, :-1]; ow: mel minus the 30 s of padding
        states = [_ClipState(i, max(1, frames[i] - tail), len(clips[i]) / SAMPLE_RATE) for i in range(len(clips))]
        initial: List[int] = []
        if o.initial_prompt is not None:
            initial = (self.tokenizer.encode(" " + o.initial_prompt.strip()) if isinstance(o.initial_prompt, str)
                       else list(o.initial_prompt))
            for st in states:
                st.all_tokens.extend(initial)
        tb = self.tokens.timestamp_begin
        if o.language is None:
            self._detect_languages(states, feats, float(o.language_detection_threshold or 0.5),
                                   max(1, int(o.language_detection_segments or 1)))
        else:
            for st in states:
                st.language = o.language

        while True:
            active = [st for st in states if st.active]
            if not active:
                break
            half = self.max_batch // 2
            overlap = bool(self.overlap_encode and half >= 1 and len(active) > half and not o.word_timestamps)
            step = half if overlap else self.max_batch
            if self.bucket_by_length and len(active) > step:
                # length bucketing: windows holding similar amounts of audio decode together.  A search ends when its
                # window's text does, the number of tokens grows with the audio in the window, and a batch runs until its
                # slowest window is done -- so batches of similar windows end together instead of idling behind a few long
                # ones (results are stored per clip: the order of the work does not show in the output)
                active.sort(key=lambda st: -min(N_FRAMES, st.n_frames_content - st.seek))
            # Windows go through the engine in chunks.  Without overlap: full batches + a tail (even batches were measured
            # and are slower: 384 windows x 5 beams = 1920 rows fill the 128-row GEMM tiles exactly).  With overlap (the
            # default when there is more than half a batch of work and no word alignment is asked for): chunks of HALF the
            # resident slots, and while chunk i decodes out of one half, chunk i + 1 is ENCODED into the other half on a
            # second stream -- the encoder is matrix-core bound, the decode step HBM / latency bound, they share the chip
            # instead of taking turns (wj_whisper_encode_at).
            chunks = [active[lo: lo + step] for lo in range(0, len(active), step)]

            def make_mel(batch):
                mel = torch.empty((len(batch), self.dims.n_mels, N_FRAMES), dtype=torch.float32, device=feats.device)
                sizes = []
                for j, st in enumerate(batch):
                    size = min(N_FRAMES, st.n_frames_content - st.seek)
                    sizes.append(size)
                    mel[j] = feats[st.index, :, st.seek: st.seek + N_FRAMES]
                    if size < N_FRAMES:
                        mel[j, :, size:] = 0.0
                return mel, sizes

            side = None
            pending = None
            split = None
            if overlap:
                # encoder_cus > 0: the overlapped pair runs on disjoint compute units (engine.HipWhisper.cu_split); the first
                # chunk's encoder and the last chunk's decode have nothing beside them and take the whole chip
                split = self.model.cu_split(self.encoder_cus) if self.encoder_cus else None
                if split is None and self._side_stream is None:
                    self._side_stream = torch.cuda.Stream(device=feats.device)
                side = split[0] if split is not None else self._side_stream
                self._overlapped = True
                pending = make_mel(chunks[0])
                self.model.encode_at(pending[0], 0, None)
                self.model.ctx.sync()
            # if a decode of the loop raises while the next chunk's encoder is in flight on the side stream, the caller's
            # retry (asr._run_model bisects and calls transcribe_many again) must not start an encode on the shared workspaces
            # beside it: whatever the exit, the side stream is drained and the decode stream reset
            try:
                for ci, batch in enumerate(chunks):
                    base = (ci % 2) * half if overlap else 0
                    if overlap:
                        mel, sizes = pending
                        paired = ci + 1 < len(chunks)
                        if paired:                           # the next chunk's encoder runs beside this chunk's decode
                            pending = make_mel(chunks[ci + 1])
                            self.model.encode_at(pending[0], ((ci + 1) % 2) * half, side)
                        self.model.decode_stream = split[1] if (split is not None and paired) else None
                    else:
                        t_ph = time.perf_counter()
                        mel, sizes = make_mel(batch)
                        self.model.encode(mel)
                        self._phase("encode_call", t_ph)
                    # windows with equal prompt lengths decode together (the common case: no previous text)
                    groups: Dict[int, List[Tuple[_ClipState, List[int], int]]] = {}
                    prompts = [self._prompt(o, st.all_tokens[st.prompt_reset_since:], st.seek == 0, st.language) for st in batch]
                    for j, p in enumerate(prompts):
                        groups.setdefault(len(p), []).append((batch[j], p, j))
                    if len(groups) == 1:
                        slots = [base + j for j in range(len(batch))]
                        t_ph = time.perf_counter()
                        decoded = self._decode_windows(prompts, o, suppress, slots=None if base == 0 else slots)
                        self._phase("decode", t_ph)
                        t_ph = time.perf_counter()
                        self._finish_windows(o, batch, sizes, decoded, slots, tb)
                        self._phase("finish_host", t_ph)
                    else:   # heterogeneous prompt lengths: one decode per length, addressing the resident windows by slot
                        for _, members in groups.items():
                            idx = [j for _, _, j in members]
                            res = self._decode_windows([p for _, p, _ in members], o, suppress, slots=[base + j for j in idx])
                            self._finish_windows(o, [batch[j] for j in idx], [sizes[j] for j in idx], res, [base + j for j in idx], tb)
                    del mel
                    if overlap:
                        self.model.decode_stream = None
                        if paired:                           # chunk ci + 1 is encoded before its decode starts
                            if split is not None:
                                self.model.stream_sync(side)
                            else:
                                side.synchronize()
            finally:
                if overlap:
                    self.model.decode_stream = None
                    try:
                        if split is not None:
                            self.model.stream_sync(side)
                        elif side is not None:
                            side.synchronize()
                    except Exception:       # the device is already in an error state: the original exception is the one to report
                        pass
                    pending = None
        infos = [TranscriptionInfo(language=st.language, language_probability=st.language_probability, duration=st.duration,
                                   duration_after_vad=st.duration, all_language_probs=st.all_language_probs,
                                   transcription_options=dict(kwargs))
                 for st in states]
        return [st.segments for st in states], infos


class HipOpenAIWhisperModel(HipWhisperModel):
    """Drop-in for the object ``whisper.load_model(name, device)`` returns, as used by the fidelity pipeline
    (/root/reference/whisperjav/modules/whisper_pro_asr.py:182,433): ``transcribe(audio, **kw) -> dict`` with
    ``{"text", "segments": [{id, seek, start, end, text, tokens, temperature, avg_logprob, compression_ratio,
    no_speech_prob}], "language"}``.  Restates openai-whisper 20250625 ``whisper/transcribe.py``: log-mel with
    30 s of zero audio appended, windows zero-padded to 3000 frames, ``DecodingTask`` search (greedy or
    ``BeamSearchDecoder`` + ``MaximumLikelihoodRanker``), ``sample_len`` 224, the same timestamp slicing."""

    MEL_MODE = "ow"
    FLAVOR = "ow"

    _RENAMES = {"logprob_threshold": "log_prob_threshold"}
    _DROPPED = ("verbose", "fp16", "carry_initial_prompt", "sample_len", "prompt")   # prompt: whisper.transcribe overwrites it per window

    def _options(self, kw: Dict[str, Any]) -> TranscribeOptions:
        kw = dict(kw)
        sample_len = kw.get("sample_len")
        for k in self._DROPPED:
            kw.pop(k, None)
        for a, b in self._RENAMES.items():
            if a in kw:
                kw[b] = kw.pop(a)
        if isinstance(kw.get("suppress_tokens"), str):
            kw["suppress_tokens"] = [int(t) for t in kw["suppress_tokens"].split(",") if t.strip()]
        if kw.get("beam_size") is None:
            kw["beam_size"] = 1
        kw.setdefault("length_penalty", None)
        kw.setdefault("no_speech_threshold", 0.6)
        kw.setdefault("condition_on_previous_text", True)
        if sample_len is not None:
            kw["max_new_tokens"] = int(sample_len)
        return super()._options(kw)

    def transcribe(self, audio, **kwargs) -> Dict[str, Any]:      # type: ignore[override]
        segs, infos = self.transcribe_many([audio], **kwargs)
        segments = [{"id": i, "seek": s.seek, "start": s.start, "end": s.end, "text": s.text, "tokens": s.tokens,
                     "temperature": s.temperature, "avg_logprob": s.avg_logprob,
                     "compression_ratio": s.compression_ratio, "no_speech_prob": s.no_speech_prob}
                    for i, s in enumerate(segs[0])]
        for d, s in zip(segments, segs[0]):
            if s.words is not None:     # whisper.transcribe: segment["words"] = [{word, start, end, probability}]
                d["words"] = [{"word": w.word, "start": w.start, "end": w.end, "probability": w.probability} for w in s.words]
        tokens = [t for s in segs[0] for t in s.tokens if t < self.tokens.eot]
        return {"text": self.tokenizer.decode(tokens), "segments": segments, "language": infos[0].language}
