"""Python face of the HIP engine: device-resident log-mel, Whisper encode / decode, VAD scoring.

PyTorch is used for plumbing only (HBM allocation, host<->device copies, the RCCL weight
broadcast in ``sharding.py``); every FLOP of the path runs in libwjhip's hand-written gfx950
kernels through the C ABI (``include/wjhip.h``).  Nothing here falls back to PyTorch or NumPy math:
without the library or without an MI355X the constructors raise.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from . import hipbind
from .dims import SpecialTokens, WhisperDims, special_tokens
from .hipbind import DTYPES, WJ_MEL_FW, WJ_MEL_OW, WJ_MEL_RAW, check

N_FRAMES = 3000
MEL_MODES = {"fw": WJ_MEL_FW, "ow": WJ_MEL_OW, "raw": WJ_MEL_RAW}
TORCH_DTYPES = {"float32": torch.float32, "bfloat16": torch.bfloat16, "float16": torch.float16}


def _require_gpu(device: int) -> torch.device:
    if not torch.cuda.is_available():
        raise hipbind.WjError("no ROCm device visible: the HIP path has no CPU fallback")
    return torch.device("cuda", device)


def _ptr(t: torch.Tensor) -> C.c_void_p:
    return C.c_void_p(t.data_ptr())


def _torch_sync() -> None:
    torch.cuda.current_stream().synchronize()


class HipLogMel:
    """Batched log-mel extractor (``wj_logmel_f32``).

    ``mode='fw'`` reproduces faster-whisper's FeatureExtractor (+ zero ``pad_or_trim`` of the frame
    axis), ``mode='ow'`` openai-whisper's ``log_mel_spectrogram(padding=N_SAMPLES)``.
    """

    def __init__(self, n_mels: int = 128, mode: str = "fw", device: int = 0):
        if mode not in MEL_MODES:
            raise ValueError(f"mode must be one of {sorted(MEL_MODES)}")
        self.n_mels, self.mode, self.device = int(n_mels), mode, int(device)
        self.dev = _require_gpu(device)
        self.ctx = hipbind.context(device)
        self._lib = hipbind.lib()

    def frames(self, n_samples: int) -> int:
        return int(self._lib.wj_logmel_frames(int(n_samples), MEL_MODES[self.mode]))

    def from_device(self, pcm: torch.Tensor, offsets: Sequence[int], out_frames: int = N_FRAMES) -> torch.Tensor:
        """pcm: float32 CUDA tensor holding all clips back to back; offsets: n_clips+1 sample offsets."""
        if pcm.dtype != torch.float32 or not pcm.is_cuda or not pcm.is_contiguous():
            raise ValueError("pcm must be a contiguous float32 CUDA tensor")
        n = len(offsets) - 1
        off = (C.c_int64 * (n + 1))(*[int(o) for o in offsets])
        out = torch.empty((n, self.n_mels, out_frames), dtype=torch.float32, device=self.dev)
        _torch_sync()
        check(self._lib.wj_logmel_f32(self.ctx.handle, _ptr(pcm), off, n, self.n_mels, MEL_MODES[self.mode],
                                      int(out_frames), _ptr(out), None), "wj_logmel_f32")
        self.ctx.sync()
        return out

    def __call__(self, clips: Sequence[np.ndarray], out_frames: int = N_FRAMES) -> torch.Tensor:
        if len(clips) and all(isinstance(c, torch.Tensor) and c.is_cuda for c in clips):
            # clips already resident in HBM (views of one uploaded recording): gather on the device
            offsets = np.concatenate([[0], np.cumsum([int(c.numel()) for c in clips])]).astype(np.int64)
            pcm = torch.cat([c.reshape(-1).to(torch.float32) for c in clips])
            return self.from_device(pcm, offsets.tolist(), out_frames)
        clips = [c.detach().cpu().numpy() if isinstance(c, torch.Tensor) else c for c in clips]
        arrs = [np.ascontiguousarray(c, dtype=np.float32).reshape(-1) for c in clips]
        offsets = np.concatenate([[0], np.cumsum([a.shape[0] for a in arrs])]).astype(np.int64)
        pcm = torch.from_numpy(np.concatenate(arrs)).to(self.dev)
        return self.from_device(pcm, offsets.tolist(), out_frames)


@dataclass
class DecodeOptions:
    """Per-call decode options (a subset of faster-whisper's ``transcribe`` kwargs that reach the
    token loop; see ``FasterWhisperProASR._prepare_whisper_params``,
    /root/reference/whisperjav/modules/faster_whisper_pro_asr.py:340-436)."""
    max_new_tokens: int = 224
    suppress_blank: bool = True
    without_timestamps: bool = False
    max_initial_timestamp: Optional[float] = 1.0
    suppress_tokens: Sequence[int] = field(default_factory=tuple)
    repetition_penalty: float = 1.0
    no_repeat_ngram_size: int = 0


@dataclass
class GreedyResult:
    tokens: np.ndarray          # int32 [B, max_new] (eot padded)
    n_tokens: np.ndarray        # int32 [B]
    sum_logprob: np.ndarray     # float32 [B]
    no_speech_prob: np.ndarray  # float32 [B]
    token_logprob: np.ndarray   # float32 [B, max_new]
    beam_score: Optional[np.ndarray] = None   # beam search: the normalised score the winner was ranked by, float32 [B]

    def avg_logprob(self) -> np.ndarray:
        """``cum_logprob / (len + 1)`` as faster-whisper / openai-whisper report it."""
        return self.sum_logprob / (self.n_tokens.astype(np.float32) + 1.0)


class HipWhisper:
    """A Whisper model resident in HBM (``wj_whisper_*``)."""

    def __init__(self, dims: WhisperDims, weights: Union[Dict[str, np.ndarray], None] = None, *,
                 blob: Optional[torch.Tensor] = None, offsets: Optional[np.ndarray] = None,
                 dtype: str = "bfloat16", device: int = 0, max_batch: int = 8, max_beam: int = 1,
                 kv_len: Optional[int] = None, enc_batch: Optional[int] = None):
        """``kv_len``: positions of the self-attention KV cache per row (default ``n_text_ctx``): prompt + max_new_tokens of
        every later decode call must fit.  The cache is [layers][rows][heads][positions][64] x 2 -- 141 GB for 1920 rows at
        448 positions, 22 GB at 72 -- so a caller that knows its token budget can hold twice the windows.  ``enc_batch``:
        windows per encoder slice (default ``max_batch``): bounds the encoder workspaces (~50 MB per window)."""
        from . import weights as W
        if dtype not in DTYPES:
            raise ValueError(f"dtype must be one of {sorted(DTYPES)}")
        self.dims, self.dtype, self.device = dims, dtype, int(device)
        self.dev = _require_gpu(device)
        self.ctx = hipbind.context(device)
        self._lib = hipbind.lib()
        self.tokens: SpecialTokens = special_tokens(dims.n_vocab)
        if blob is None:
            if weights is None:
                raise ValueError("either weights or (blob, offsets) is required")
            host_blob, offsets = W.pack_blob(dims, weights, dtype)
            blob = host_blob.to(self.dev)
        elif offsets is None:
            raise ValueError("offsets are required with a pre-packed blob")
        if not blob.is_cuda or blob.dtype != torch.uint8:
            raise ValueError("blob must be a uint8 CUDA tensor")
        self.blob = blob  # keeps the HBM alive; the library only borrows it
        self.offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        self.max_batch, self.max_beam = int(max_batch), int(max_beam)
        cd = hipbind.WhisperDimsC(**dims.as_dict())
        handle = C.c_void_p()
        off = (C.c_int64 * len(self.offsets))(*self.offsets.tolist())
        _torch_sync()
        self.kv_len = dims.n_text_ctx if not kv_len else min(dims.n_text_ctx, (int(kv_len) + 7) // 8 * 8)
        self.enc_batch = self.max_batch if not enc_batch else min(self.max_batch, int(enc_batch))
        hipbind.tune("self_kv_len", 0 if self.kv_len >= dims.n_text_ctx else self.kv_len)      # read by wj_whisper_create
        hipbind.tune("enc_batch", 0 if self.enc_batch >= self.max_batch else self.enc_batch)
        check(self._lib.wj_whisper_create(self.ctx.handle, C.byref(cd), DTYPES[dtype], _ptr(blob), blob.numel(), off,
                                          len(self.offsets), self.max_batch, self.max_batch * self.max_beam,
                                          C.byref(handle)), "wj_whisper_create")
        hipbind.tune("self_kv_len", 0)
        hipbind.tune("enc_batch", 0)
        self.handle = handle
        self.decode_stream = None        # raw stream handle the decode calls run on (None = the context's stream)
        self._split = None               # (encoder stream, decode stream) over disjoint CU sets, see cu_split()
        self._suppress_mask: Optional[torch.Tensor] = None
        self._suppress_key = None

    # ---- lifetime -------------------------------------------------------------------------
    def close(self) -> None:
        if getattr(self, "handle", None):
            if getattr(self, "_split", None):
                for h in self._split:
                    self._lib.wj_stream_destroy(self.ctx.handle, h)
                self._split = None
            self._lib.wj_whisper_free(self.handle)
            self.handle = None

    def __del__(self):  # the reference may never call cleanup() (os._exit); both paths are fine
        try:
            self.close()
        except Exception:
            pass

    @property
    def workspace_bytes(self) -> int:
        return int(self._lib.wj_whisper_workspace_bytes(self.handle))

    # ---- encoder --------------------------------------------------------------------------
    def encode(self, mel: torch.Tensor, n_layers: int = -1, want_output: bool = False) -> Optional[torch.Tensor]:
        """mel: float32 CUDA ``[B, n_mels, 3000]``.  Leaves encoder output + cross K/V resident."""
        d = self.dims
        if mel.dtype != torch.float32 or not mel.is_cuda or not mel.is_contiguous():
            raise ValueError("mel must be a contiguous float32 CUDA tensor")
        if mel.dim() != 3 or mel.shape[1] != d.n_mels or mel.shape[2] != 2 * d.n_audio_ctx:
            raise ValueError(f"mel must be [B, {d.n_mels}, {2 * d.n_audio_ctx}], got {tuple(mel.shape)}")
        B = mel.shape[0]
        out = None
        if want_output:
            out = torch.empty((B, d.n_audio_ctx, d.n_audio_state), dtype=torch.float32, device=self.dev)
        _torch_sync()
        check(self._lib.wj_whisper_encode(self.handle, _ptr(mel), B, int(n_layers),
                                          _ptr(out) if out is not None else None, None), "wj_whisper_encode")
        self.ctx.sync()
        return out

    def cu_split(self, encoder_cus: int):
        """Two streams over disjoint compute units: the first ``encoder_cus`` CUs for ``encode_at``, the rest for the decode
        calls (``wj_stream_create``).  Returns ``(encoder stream, decode stream)`` as raw handles; ``encoder_cus <= 0``
        drops the split.  Plain streams only alternate on the chip: a 256-tile GEMM workgroup takes a CU's whole register
        file and LDS, so a decode kernel cannot co-reside with the encoder unless the encoder is kept off some CUs."""
        if self._split is not None:
            for h in self._split:
                self._lib.wj_stream_destroy(self.ctx.handle, h)
            self._split = None
        if encoder_cus and encoder_cus > 0:
            n = int(self.ctx.device_info()["cu_count"]) if hasattr(self.ctx, "device_info") else 256
            if not 0 < encoder_cus < n:
                raise ValueError(f"encoder_cus must be in 1..{n - 1}")
            enc, dec = C.c_void_p(), C.c_void_p()
            check(self._lib.wj_stream_create(self.ctx.handle, 0, int(encoder_cus), C.byref(enc)), "wj_stream_create")
            check(self._lib.wj_stream_create(self.ctx.handle, int(encoder_cus), n - int(encoder_cus), C.byref(dec)), "wj_stream_create")
            self._split = (enc, dec)
        return self._split

    def stream_sync(self, handle) -> None:
        check(self._lib.wj_stream_sync(self.ctx.handle, handle), "wj_stream_sync")

    def encode_at(self, mel: torch.Tensor, slot0: int, stream=None) -> None:
        """Encode ``mel`` [B, n_mels, 3000] into the resident window slots ``slot0 .. slot0 + B - 1`` WITHOUT waiting:
        the launches go to ``stream`` (a torch stream, or a raw handle from ``cu_split``) or the context's stream.  The
        caller synchronises that stream before decoding those slots and keeps ``mel`` alive until then."""
        d = self.dims
        if mel.dtype != torch.float32 or not mel.is_cuda or not mel.is_contiguous():
            raise ValueError("mel must be a contiguous float32 CUDA tensor")
        if mel.dim() != 3 or mel.shape[1] != d.n_mels or mel.shape[2] != 2 * d.n_audio_ctx:
            raise ValueError(f"mel must be [B, {d.n_mels}, {2 * d.n_audio_ctx}], got {tuple(mel.shape)}")
        _torch_sync()           # mel was produced on torch's stream
        raw = None if stream is None else (C.c_void_p(stream.cuda_stream) if hasattr(stream, "cuda_stream") else stream)
        check(self._lib.wj_whisper_encode_at(self.handle, _ptr(mel), int(mel.shape[0]), int(slot0), raw), "wj_whisper_encode_at")

    # ---- decoding -------------------------------------------------------------------------
    def _mask_for(self, suppress: Sequence[int]) -> Optional[torch.Tensor]:
        key = tuple(sorted(set(int(t) for t in suppress if t >= 0)))
        if not key:
            return None
        if self._suppress_key != key:
            m = torch.zeros(self.dims.n_vocab, dtype=torch.uint8)
            m[list(key)] = 1
            self._suppress_mask = m.to(self.dev)
            self._suppress_key = key
            _torch_sync()
        return self._suppress_mask

    def _opts(self, o: DecodeOptions) -> hipbind.DecodeOptsC:
        t = self.tokens
        mask = self._mask_for(o.suppress_tokens)
        idx = -1
        if o.max_initial_timestamp is not None:
            idx = int(round(float(o.max_initial_timestamp) / 0.02))
        return hipbind.DecodeOptsC(
            max_new_tokens=int(o.max_new_tokens), suppress_blank=int(bool(o.suppress_blank)),
            without_timestamps=int(bool(o.without_timestamps)), max_initial_timestamp_index=idx,
            eot=t.eot, no_timestamps=t.no_timestamps, timestamp_begin=t.timestamp_begin, blank=t.blank,
            no_speech=t.no_speech, suppress_mask_dev=mask.data_ptr() if mask is not None else None,
            repetition_penalty=float(o.repetition_penalty), no_repeat_ngram_size=int(o.no_repeat_ngram_size))

    def decode_greedy(self, prompts: np.ndarray, options: Optional[DecodeOptions] = None) -> GreedyResult:
        """Greedy decode of the windows currently resident (rows of ``prompts`` = windows)."""
        o = options or DecodeOptions()
        prompts = np.ascontiguousarray(prompts, dtype=np.int32)
        if prompts.ndim != 2:
            raise ValueError("prompts must be [batch, prompt_len]")
        B, P = prompts.shape
        oc = self._opts(o)
        n = o.max_new_tokens
        toks = np.empty((B, n), dtype=np.int32)
        ntok = np.empty(B, dtype=np.int32)
        slp = np.empty(B, dtype=np.float32)
        nsp = np.empty(B, dtype=np.float32)
        tlp = np.empty((B, n), dtype=np.float32)
        as_i = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))  # noqa: E731
        as_f = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))  # noqa: E731
        check(self._lib.wj_whisper_decode_greedy(self.handle, B, as_i(prompts), P, C.byref(oc), as_i(toks), as_i(ntok),
                                                 as_f(slp), as_f(nsp), as_f(tlp), self.decode_stream), "wj_whisper_decode_greedy")
        return GreedyResult(toks, ntok, slp, nsp, tlp)

    def decode_sample(self, prompts: np.ndarray, options: Optional[DecodeOptions] = None, *, temperature: float = 0.0,
                      best_of: int = 1, slots: Optional[Sequence[int]] = None, seed: int = 0) -> GreedyResult:
        """``best_of`` rows per window (sharing its cross K/V) decoded at ``temperature``; ``slots`` selects which
        resident windows are decoded (default: 0..len(prompts)-1).  Results are per row, window-major."""
        o = options or DecodeOptions()
        prompts = np.ascontiguousarray(prompts, dtype=np.int32)
        B, P = prompts.shape
        R, n = B * int(best_of), o.max_new_tokens
        oc = self._opts(o)
        toks = np.empty((R, n), dtype=np.int32)
        ntok = np.empty(R, dtype=np.int32)
        slp = np.empty(R, dtype=np.float32)
        nsp = np.empty(R, dtype=np.float32)
        tlp = np.empty((R, n), dtype=np.float32)
        as_i = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))  # noqa: E731
        as_f = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))  # noqa: E731
        sl = None
        if slots is not None:
            sl_arr = np.ascontiguousarray(slots, dtype=np.int32)
            if sl_arr.shape != (B,):
                raise ValueError("slots must name one resident window per prompt row")
            sl = as_i(sl_arr)
        check(self._lib.wj_whisper_decode_sample(self.handle, B, int(best_of), sl, as_i(prompts), P, C.byref(oc),
                                                 float(temperature), int(seed) & 0xFFFFFFFF, as_i(toks), as_i(ntok), as_f(slp),
                                                 as_f(nsp), as_f(tlp), self.decode_stream), "wj_whisper_decode_sample")
        return GreedyResult(toks, ntok, slp, nsp, tlp)

    def decode_beam(self, prompts: np.ndarray, options: Optional[DecodeOptions] = None, *, beam_size: int = 5,
                    patience: float = 1.0, length_penalty: Optional[float] = 1.0, slots: Optional[Sequence[int]] = None,
                    flavor: str = "ct2", token_logprobs: bool = False) -> GreedyResult:
        """Beam search of the resident windows, entirely on the device: ``flavor="ct2"`` CTranslate2's rules
        (faster-whisper), ``"openai"`` openai-whisper's ``BeamSearchDecoder`` + ``MaximumLikelihoodRanker`` (fidelity
        mode; ``length_penalty=None`` = rank by ``sum_logprob / length``).  Per window: best hypothesis tokens, count,
        cumulative log-prob (``sum_logprob``), no-speech probability, ``beam_score`` = the normalised score;
        ``token_logprob`` carries that score in column 0 unless ``token_logprobs=True``: the search then also carries the
        cumulative log-prob of every hypothesis (wj_tune ``beam_token_logprobs``; no batch compaction while it is on) and
        ``token_logprob`` is [B, max_new + 1]: log p of each token of the winner, then what ending the sequence added
        (log p(EOT), 0 at the length limit), NaN beyond -- they sum to ``sum_logprob``."""
        if flavor not in ("ct2", "openai"):
            raise ValueError("flavor must be 'ct2' or 'openai'")
        if length_penalty is None:
            if flavor == "ct2":
                raise ValueError("length_penalty=None is openai-whisper's default; CTranslate2 takes a number")
            length_penalty = -1.0
        o = options or DecodeOptions()
        prompts = np.ascontiguousarray(prompts, dtype=np.int32)
        B, P = prompts.shape
        n = o.max_new_tokens
        oc = self._opts(o)
        toks = np.empty((B, n), dtype=np.int32)
        ntok = np.empty(B, dtype=np.int32)
        score = np.empty(B, dtype=np.float32)
        slp = np.empty(B, dtype=np.float32)
        nsp = np.empty(B, dtype=np.float32)
        as_i = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))  # noqa: E731
        as_f = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))  # noqa: E731
        sl = None
        if slots is not None:
            sl_arr = np.ascontiguousarray(slots, dtype=np.int32)
            if sl_arr.shape != (B,):
                raise ValueError("slots must name one resident window per prompt row")
            sl = as_i(sl_arr)
        fn = self._lib.wj_whisper_decode_beam if flavor == "ct2" else self._lib.wj_whisper_decode_beam_openai
        if token_logprobs:
            hipbind.tune("beam_token_logprobs", 1)
        try:
            check(fn(self.handle, B, int(beam_size), sl, as_i(prompts), P, C.byref(oc), float(patience), float(length_penalty),
                     as_i(toks), as_i(ntok), as_f(score), as_f(slp), as_f(nsp), self.decode_stream), "wj_whisper_decode_beam")
        finally:
            if token_logprobs:
                hipbind.tune("beam_token_logprobs", 0)
        if token_logprobs:
            tlp = np.empty((B, n + 1), dtype=np.float32)
            check(self._lib.wj_whisper_last_beam_token_logprobs(self.handle, B, n + 1, as_f(tlp)), "wj_whisper_last_beam_token_logprobs")
            return GreedyResult(toks, ntok, slp, nsp, tlp, score.copy())
        return GreedyResult(toks, ntok, slp, nsp, score.reshape(B, 1), score.copy())

    def align(self, token_rows: Sequence[Sequence[int]], n_prefix: int, heads: Sequence[Tuple[int, int]],
              num_frames: Sequence[int], *, slots: Optional[Sequence[int]] = None, medfilt_width: int = 7):
        """Word-timestamp alignment of the resident windows.  ``token_rows[b]`` = sot sequence + <|notimestamps|> +
        text tokens + eot (``n_prefix`` = len(sot sequence) + 1).  Returns per window
        ``(text_indices, time_indices, text_token_probs)`` as numpy arrays (see ``wj_whisper_align``)."""
        B = len(token_rows)
        n_tok = np.array([len(r) for r in token_rows], dtype=np.int32)
        hd = np.ascontiguousarray(heads, dtype=np.int32).reshape(-1, 2)
        nf = np.ascontiguousarray(num_frames, dtype=np.int32)
        if nf.shape != (B,):
            raise ValueError("num_frames must hold one entry per window")
        sl_arr = None
        if slots is not None:
            sl_arr = np.ascontiguousarray(slots, dtype=np.int32)
            if sl_arr.shape != (B,):
                raise ValueError("slots must name one resident window per token row")
        # The pass runs windows x T rows through the decoder with T = the longest row of the call: rows of similar length go
        # together (round 6).  A pooled call holds windows of 4 to 200+ tokens -- one call for all of them pads every window to
        # the longest (the reference-preset bench: 106 k rows for 58 k tokens); sub-calls of at least `align_min_rows` windows
        # whose shortest row is within `align_waste` of their longest bound the padding.  A window's result does not depend
        # on its neighbours beyond the GEMM kernel the row count selects (float16: identical frames, tests/test_gpu_pipeline.py).
        order = np.argsort(-n_tok, kind="stable")
        groups: List[np.ndarray] = []
        if self.align_waste is None or B <= self.align_min_rows:
            groups = [np.arange(B)]
        else:
            lo = 0
            while lo < B:
                hi = min(B, lo + self.align_min_rows)
                while hi < B and n_tok[order[hi]] >= (1.0 - self.align_waste) * n_tok[order[lo]]:
                    hi += 1
                if B - hi < self.align_min_rows // 2:          # no crumbs at the end
                    hi = B
                groups.append(order[lo:hi])
                lo = hi
        out: List[Optional[tuple]] = [None] * B
        for idx in groups:
            for b, res in zip(idx, self._align_call([token_rows[i] for i in idx], n_tok[idx], n_prefix, hd, nf[idx],
                                                    sl_arr[idx] if sl_arr is not None else (idx.astype(np.int32) if len(groups) > 1 else None), medfilt_width)):
                out[int(b)] = res
        return out

    align_waste: Optional[float] = 0.25      # None = one call for all rows (rounds 3-5)
    align_min_rows = 64

    def _align_call(self, token_rows, n_tok, n_prefix, hd, nf, sl_arr, medfilt_width):
        B = len(token_rows)
        eot = self.tokens.eot
        n_tok = np.ascontiguousarray(n_tok, dtype=np.int32)
        nf = np.ascontiguousarray(nf, dtype=np.int32)
        T = int(n_tok.max())
        toks = np.full((B, T), eot, dtype=np.int32)
        for b, r in enumerate(token_rows):
            toks[b, : len(r)] = r
        plen = T + self.dims.n_audio_ctx
        p_text = np.empty((B, plen), dtype=np.int32)
        p_time = np.empty((B, plen), dtype=np.int32)
        p_len = np.empty(B, dtype=np.int32)
        probs = np.empty((B, T), dtype=np.float32)
        as_i = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))  # noqa: E731
        sl = None
        if sl_arr is not None:
            sl_arr = np.ascontiguousarray(sl_arr, dtype=np.int32)
            sl = as_i(sl_arr)
        check(self._lib.wj_whisper_align(self.handle, B, sl, as_i(toks), T, as_i(n_tok), int(n_prefix), as_i(hd), hd.shape[0],
                                         as_i(nf), int(medfilt_width), eot, as_i(p_text), as_i(p_time), as_i(p_len),
                                         probs.ctypes.data_as(C.POINTER(C.c_float)), None), "wj_whisper_align")
        out = []
        for b in range(B):
            n = int(p_len[b])
            out.append((p_text[b, :n].copy(), p_time[b, :n].copy(), probs[b, : int(n_tok[b]) - n_prefix - 1].copy()))
        return out

    def last_align_matrix(self, batch: int, n_rows: int, n_cols: int) -> np.ndarray:
        """Diagnostic (``wj_whisper_last_align_matrix``): the head-averaged, median-filtered matrix the last ``align`` call ran
        its DTW on, ``[batch][n_rows][n_cols]`` (text position x encoder frame; the DTW minimises the negated sum)."""
        out = np.empty((int(batch), int(n_rows), int(n_cols)), dtype=np.float32)
        check(self._lib.wj_whisper_last_align_matrix(self.handle, int(batch), int(n_rows), int(n_cols),
                                                     out.ctypes.data_as(C.POINTER(C.c_float))), "wj_whisper_last_align_matrix")
        return out

    def last_decode_info(self) -> dict:
        out = (C.c_int32 * 6)()
        check(self._lib.wj_whisper_last_decode_info(self.handle, out), "wj_whisper_last_decode_info")
        return {"hip_graph": bool(out[0]), "chains": int(out[1]), "steps": int(out[2]), "max_new_tokens": int(out[3]),
                "compactions": int(out[4]), "window_steps": int(out[5])}

    def sot_prompt(self, language: str = "ja", task: str = "transcribe", without_timestamps: bool = False) -> List[int]:
        from .dims import language_index
        t = self.tokens
        seq = [t.sot, t.language_token(language_index(language)), t.transcribe if task == "transcribe" else t.translate]
        if without_timestamps:
            seq.append(t.no_timestamps)
        return seq

    # ---- step-wise API for host-driven search (beam search lives in search.py) --------------
    def open(self, batch: int, beam: int) -> None:
        check(self._lib.wj_decode_open(self.handle, int(batch), int(beam), None), "wj_decode_open")
        self._rows = batch * beam

    def step(self, tokens: np.ndarray, parents: Optional[np.ndarray] = None, want_logits: bool = True) -> None:
        tokens = np.ascontiguousarray(tokens, dtype=np.int32)
        if tokens.shape != (self._rows,):
            raise ValueError(f"tokens must have shape ({self._rows},)")
        pp = None
        if parents is not None:
            parents = np.ascontiguousarray(parents, dtype=np.int32)
            pp = parents.ctypes.data_as(C.POINTER(C.c_int32))
        check(self._lib.wj_decode_step(self.handle, tokens.ctypes.data_as(C.POINTER(C.c_int32)), pp,
                                       int(want_logits), None), "wj_decode_step")

    def logits(self) -> torch.Tensor:
        """Copy of the device logits of the last step: float32 CUDA ``[rows, n_vocab]``."""
        out = torch.empty((self._rows, self.dims.n_vocab), dtype=torch.float32, device=self.dev)
        _torch_sync()
        check(self._lib.wj_decode_logits_copy(self.handle, self._rows, _ptr(out), None), "wj_decode_logits_copy")
        self.ctx.sync()
        return out

    def language_probs(self, n_windows: int) -> np.ndarray:
        """ctranslate2 ``Whisper.detect_language`` / ``whisper.decoding.detect_language`` for the resident windows:
        one decoder step on ``<|startoftranscript|>``, softmax over the language tokens only -> float32
        ``[n_windows, num_languages]`` (order of ``dims.LANGUAGE_CODES``)."""
        t = self.tokens
        self.open(int(n_windows), 1)
        self.step(np.full(int(n_windows), t.sot, dtype=np.int32), want_logits=True)
        lg = self.logits()[:, t.sot + 1: t.sot + 1 + t.num_languages].cpu().numpy().astype(np.float64)
        lg -= lg.max(axis=1, keepdims=True)
        p = np.exp(lg)
        return (p / p.sum(axis=1, keepdims=True)).astype(np.float32)

    def topk(self, k: int, ban: Optional[torch.Tensor] = None):
        rows = self._rows
        ids = np.empty((rows, k), dtype=np.int32)
        lps = np.empty((rows, k), dtype=np.float32)
        lse = np.empty(rows, dtype=np.float32)
        if ban is not None:
            if ban.dtype != torch.uint8 or not ban.is_cuda or ban.shape != (rows, self.dims.n_vocab):
                raise ValueError("ban must be a uint8 CUDA tensor [rows, n_vocab]")
            _torch_sync()
        check(self._lib.wj_decode_topk(self.handle, rows, int(k), _ptr(ban) if ban is not None else None,
                                       ids.ctypes.data_as(C.POINTER(C.c_int32)),
                                       lps.ctypes.data_as(C.POINTER(C.c_float)),
                                       lse.ctypes.data_as(C.POINTER(C.c_float)), None), "wj_decode_topk")
        return ids, lps, lse


# ---- kernel-level entry points used by the parity tests ----------------------------------------
def k_gemm(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], dtype: str, gelu=False, out_f32=False,
           variant=0, device: int = 0) -> torch.Tensor:
    """C = A @ W^T (+bias); a [M,K], w [N,K] float32 CUDA tensors, converted to ``dtype`` first."""
    ctx = hipbind.context(device)
    lib = hipbind.lib()
    td = TORCH_DTYPES[dtype]
    A, Wt = a.to(td).contiguous(), w.to(td).contiguous()
    M, K = A.shape
    N = Wt.shape[0]
    out = torch.empty((M, N), dtype=torch.float32 if out_f32 else td, device=a.device)
    _torch_sync()
    check(lib.wj_k_gemm(ctx.handle, DTYPES[dtype], _ptr(A), _ptr(Wt), _ptr(bias) if bias is not None else None,
                        _ptr(out), M, N, K, int(gelu), int(out_f32), int(variant), None), "wj_k_gemm")
    ctx.sync()
    return out.float()


def k_gemm_split(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], dtype: str, variant=0,
                 device: int = 0) -> torch.Tensor:
    """C = A @ W^T (+bias) with A (float32 CUDA [M,K]) entering the matrix cores as hi + lo 16-bit pairs and W in
    ``dtype`` -- the decode-step GEMM of the fp16 compute type (``wj_k_gemm_split``).  float32 [M,N]."""
    ctx = hipbind.context(device)
    lib = hipbind.lib()
    A, Wt = a.float().contiguous(), w.to(TORCH_DTYPES[dtype]).contiguous()
    M, K = A.shape
    N = Wt.shape[0]
    out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    _torch_sync()
    check(lib.wj_k_gemm_split(ctx.handle, DTYPES[dtype], _ptr(A), _ptr(Wt), _ptr(bias) if bias is not None else None,
                              _ptr(out), M, N, K, int(variant), None), "wj_k_gemm_split")
    ctx.sync()
    return out


def k_gemm_mx8(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, dtype: str = "float16", out_f32: bool = True,
               reps: int = 0, device: int = 0):
    """C = A @ W^T (+bias) with BOTH operands quantised on the device to MX-fp8 (OCP e4m3, one E8M0 scale per 32 elements) and
    multiplied on the block-scaled matrix-core instruction (``wj_k_gemm_mx8``: the arithmetic of the Qwen decoder's "float8w"
    type).  Returns (C, a8, a_scale, w8, w_scale, ms_per_launch): the quantised bytes come back so that a test can rebuild the
    exact product on the host."""
    import ctypes as C
    ctx = hipbind.context(device)
    lib = hipbind.lib()
    A, Wt = a.float().contiguous(), w.float().contiguous()
    M, K = A.shape
    N = Wt.shape[0]
    out = torch.empty((M, N), dtype=torch.float32 if out_f32 else TORCH_DTYPES[dtype], device=a.device)
    a8 = torch.empty((M, K), dtype=torch.uint8, device=a.device)
    sa = torch.empty((M, K // 32), dtype=torch.uint8, device=a.device)
    w8 = torch.empty((N, K), dtype=torch.uint8, device=a.device)
    sw = torch.empty((N, K // 32), dtype=torch.uint8, device=a.device)
    ms = C.c_float(0.0)
    _torch_sync()
    check(lib.wj_k_gemm_mx8(ctx.handle, DTYPES[dtype], _ptr(A), _ptr(Wt), _ptr(bias) if bias is not None else None, _ptr(out), M, N, K,
                            int(out_f32), _ptr(a8), _ptr(sa), _ptr(w8), _ptr(sw), int(reps), C.byref(ms), None), "wj_k_gemm_mx8")
    ctx.sync()
    return out, a8, sa, w8, sw, float(ms.value)


def k_gemm_timed(M: int, N: int, K: int, dtype: str = "bfloat16", variant: int = 0, reps: int = 20,
                 gelu: bool = False, device: int = 0) -> float:
    """Average milliseconds per launch of an [M,K] x [N,K]^T GEMM on uniform random [-1, 1) operands."""
    ctx = hipbind.context(device)
    lib = hipbind.lib()
    td = TORCH_DTYPES[dtype]
    dev = torch.device("cuda", device)
    g = torch.Generator(device=dev).manual_seed(M + N + K)
    A = (torch.rand((M, K), device=dev, generator=g) * 2 - 1).to(td)
    Wt = (torch.rand((N, K), device=dev, generator=g) * 2 - 1).to(td)
    bias = torch.rand(N, device=dev, generator=g)
    out = torch.empty((M, N), dtype=td, device=dev)
    ms = C.c_float()
    _torch_sync()
    check(lib.wj_k_gemm_timed(ctx.handle, DTYPES[dtype], _ptr(A), _ptr(Wt), _ptr(bias), _ptr(out), M, N, K, int(gelu), 0,
                              int(variant), int(reps), C.byref(ms)), "wj_k_gemm_timed")
    return float(ms.value)


def k_layernorm(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, dtype: str, device: int = 0) -> torch.Tensor:
    ctx = hipbind.context(device)
    lib = hipbind.lib()
    td = TORCH_DTYPES[dtype]
    M, D = x.shape
    out = torch.empty((M, D), dtype=td, device=x.device)
    _torch_sync()
    check(lib.wj_k_layernorm(ctx.handle, DTYPES[dtype], _ptr(x.contiguous()), _ptr(w), _ptr(b), _ptr(out), M, D, None),
          "wj_k_layernorm")
    ctx.sync()
    return out.float()


def k_attention_enc(qkv: torch.Tensor, heads: int, dtype: str, device: int = 0) -> torch.Tensor:
    """qkv float32 CUDA [B, T, 3*D] -> attention output float32 [B, T, D]."""
    ctx = hipbind.context(device)
    lib = hipbind.lib()
    td = TORCH_DTYPES[dtype]
    B, T, D3 = qkv.shape
    out = torch.empty((B, T, D3 // 3), dtype=td, device=qkv.device)
    _torch_sync()
    check(lib.wj_k_attention_enc(ctx.handle, DTYPES[dtype], _ptr(qkv.contiguous()), _ptr(out), B, T, heads, None),
          "wj_k_attention_enc")
    ctx.sync()
    return out.float()


def k_attention_dec(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, dtype: str, device: int = 0) -> torch.Tensor:
    """q [G, nb, H*64], k/v [G, H, n_keys, 64] float32 CUDA -> out float32 [G, nb, H*64]."""
    ctx = hipbind.context(device)
    lib = hipbind.lib()
    G, nb, D = q.shape
    H, n_keys = k.shape[1], k.shape[2]
    out = torch.empty((G, nb, D), dtype=torch.float32, device=q.device)
    _torch_sync()
    check(lib.wj_k_attention_dec(ctx.handle, DTYPES[dtype], _ptr(q.contiguous()), _ptr(k.contiguous()),
                                 _ptr(v.contiguous()), _ptr(out), G, nb, H, n_keys, None), "wj_k_attention_dec")
    ctx.sync()
    return out


def k_attention_dec_timed(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, dtype: str, layout: int = 0, reps: int = 20,
                          device: int = 0) -> Tuple[torch.Tensor, float]:
    """As ``k_attention_dec`` plus the mean milliseconds per launch over ``reps`` back-to-back launches.
    ``layout`` 0 = the engine's layout for the dtype, 1 = row-major V with the vector kernel."""
    ctx = hipbind.context(device)
    lib = hipbind.lib()
    G, nb, D = q.shape
    H, n_keys = k.shape[1], k.shape[2]
    out = torch.empty((G, nb, D), dtype=torch.float32, device=q.device)
    ms = C.c_float(0.0)
    _torch_sync()
    check(lib.wj_k_attention_dec_timed(ctx.handle, DTYPES[dtype], _ptr(q.contiguous()), _ptr(k.contiguous()),
                                       _ptr(v.contiguous()), _ptr(out), G, nb, H, n_keys, int(layout), int(reps),
                                       C.byref(ms)), "wj_k_attention_dec_timed")
    ctx.sync()
    return out, float(ms.value)
