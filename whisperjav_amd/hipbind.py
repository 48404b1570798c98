"""ctypes binding of libwjhip.so (the C ABI in include/wjhip.h).

There is deliberately NO fallback: if the library is missing, was built for another ABI version,
or no gfx950 device is visible, the calls raise.  ``lib()`` only needs the shared object (it can be
loaded without a GPU, e.g. to check the exported symbols); anything that touches a device goes
through ``Context``.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path
from typing import Dict, Optional

ABI_VERSION = 6
_LIB_PATH = Path(__file__).resolve().parent / "csrc" / "libwjhip.so"
_lib: Optional[C.CDLL] = None

WJ_F32, WJ_BF16, WJ_F16, WJ_F8W = 0, 1, 2, 3      # WJ_F8W: wj_qwen_create only (MX-fp8 decoder projections)
WJ_MEL_FW, WJ_MEL_OW, WJ_MEL_RAW = 0, 1, 2
WJ_E_UNSUPPORTED = -5       # include/wjhip.h: the status of an entry point whose run-time dependency (RCCL symbol) is absent
DTYPES = {"float32": WJ_F32, "bfloat16": WJ_BF16, "float16": WJ_F16}


class WjError(RuntimeError):
    """A libwjhip call failed (message from wj_last_error())."""


class WhisperDimsC(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "n_mels", "n_audio_ctx", "n_audio_state", "n_audio_head", "n_audio_layer",
        "n_vocab", "n_text_ctx", "n_text_state", "n_text_head", "n_text_layer")]


class DecodeOptsC(C.Structure):
    _fields_ = [
        ("max_new_tokens", C.c_int32), ("suppress_blank", C.c_int32), ("without_timestamps", C.c_int32),
        ("max_initial_timestamp_index", C.c_int32), ("eot", C.c_int32), ("no_timestamps", C.c_int32),
        ("timestamp_begin", C.c_int32), ("blank", C.c_int32), ("no_speech", C.c_int32),
        ("suppress_mask_dev", C.c_void_p), ("repetition_penalty", C.c_float), ("no_repeat_ngram_size", C.c_int32),
    ]


# every exported symbol of include/wjhip.h with its ctypes signature (restype, argtypes)
_P, _I, _I64, _F = C.c_void_p, C.c_int, C.c_int64, C.c_float
_SIGNATURES = {
    "wj_abi_version": (_I, []),
    "wj_last_error": (C.c_char_p, []),
    "wj_init": (_I, [_I, C.POINTER(_P)]),
    "wj_shutdown": (_I, [_P]),
    "wj_sync": (_I, [_P]),
    "wj_stream_create": (_I, [_P, _I, _I, C.POINTER(_P)]),
    "wj_stream_sync": (_I, [_P, _P]),
    "wj_stream_destroy": (_I, [_P, _P]),
    "wj_device_info": (_I, [_P, C.POINTER(_I64)]),
    "wj_tune": (_I, [C.c_char_p, _I]),
    "wj_profile_start": (_I, [_P]),
    "wj_profile_tags": (_I, []),
    "wj_profile_tag_name": (C.c_char_p, [_I]),
    "wj_profile_stop": (_I, [_P, C.POINTER(C.c_double), C.POINTER(_I64), _I]),
    "wj_profile_stop_ex": (_I, [_P, C.POINTER(C.c_double), C.POINTER(_I64), C.POINTER(_I64), _I]),
    "wj_logmel_frames": (_I64, [_I64, _I]),
    "wj_logmel_f32": (_I, [_P, _P, C.POINTER(_I64), _I, _I, _I, _I, _P, _P]),
    "wj_frame_sumsq": (_I, [_P, _P, _I64, C.POINTER(C.c_int64), C.POINTER(C.c_int32), _I64, C.POINTER(C.c_int64), _P]),
    "wj_whisper_create": (_I, [_P, C.POINTER(WhisperDimsC), _I, _P, _I64, C.POINTER(_I64), _I, _I, _I, C.POINTER(_P)]),
    "wj_whisper_free": (_I, [_P]),
    "wj_whisper_workspace_bytes": (_I64, [_P]),
    "wj_whisper_encode": (_I, [_P, _P, _I, _I, _P, _P]),
    "wj_whisper_encode_at": (_I, [_P, _P, _I, _I, _P]),
    "wj_whisper_decode_greedy": (_I, [_P, _I, C.POINTER(C.c_int32), _I, C.POINTER(DecodeOptsC), C.POINTER(C.c_int32),
                                      C.POINTER(C.c_int32), C.POINTER(_F), C.POINTER(_F), C.POINTER(_F), _P]),
    "wj_whisper_decode_sample": (_I, [_P, _I, _I, C.POINTER(C.c_int32), C.POINTER(C.c_int32), _I, C.POINTER(DecodeOptsC), _F,
                                      C.c_uint32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(_F), C.POINTER(_F),
                                      C.POINTER(_F), _P]),
    "wj_whisper_decode_beam": (_I, [_P, _I, _I, C.POINTER(C.c_int32), C.POINTER(C.c_int32), _I, C.POINTER(DecodeOptsC), _F, _F, C.POINTER(C.c_int32),
                                    C.POINTER(C.c_int32), C.POINTER(_F), C.POINTER(_F), C.POINTER(_F), _P]),
    "wj_whisper_decode_beam_openai": (_I, [_P, _I, _I, C.POINTER(C.c_int32), C.POINTER(C.c_int32), _I, C.POINTER(DecodeOptsC), _F, _F,
                                           C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(_F), C.POINTER(_F), C.POINTER(_F), _P]),
    "wj_qwen_create": (_I, [_P, _P, _I, _P, C.c_size_t, C.POINTER(C.c_int64), _I, _I, _I, _I, C.POINTER(_P)]),
    "wj_qwen_free": (_I, [_P]),
    "wj_qwen_embed": (_I, [_P, C.POINTER(C.c_int32), _I, _P, _P]),
    "wj_qwen_prefill": (_I, [_P, _P, _I, C.POINTER(C.c_int32), _P, _P]),
    "wj_qwen_generate_greedy": (_I, [_P, C.POINTER(C.c_int32), _I, _I, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(_F), _P]),
    "wj_qwen_generate_greedy_ex": (_I, [_P, C.POINTER(C.c_int32), _I, _I, C.POINTER(C.c_int32), _F, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                        C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(_F), _P]),
    "wj_k_gemm_mx8": (_I, [_P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _I, C.POINTER(C.c_float), _P]),
    "wj_qwen_last_used_graph": (_I, [_P]),
    "wj_qwen_last_steps": (_I, [_P]),
    "wj_qwen_last_truncated": (_I, [_P]),
    "wj_qwen_last_compactions": (_I, [_P]),
    "wj_qwen_last_row_steps": (C.c_int64, [_P]),
    "wj_qwen_classify": (_I, [_P, _P, _I, C.POINTER(C.c_int32), C.POINTER(C.c_int32), _I, _P, _P, _I, C.POINTER(C.c_int32), _P, _P]),
    "wj_qwen_audio_create": (_I, [_P, _P, _I, _P, C.c_size_t, C.POINTER(C.c_int64), _I, _I, C.POINTER(_P)]),
    "wj_qwen_audio_free": (_I, [_P]),
    "wj_qwen_audio_tokens": (_I, [_I]),
    "wj_qwen_audio_encode": (_I, [_P, _P, _I, _I, C.POINTER(C.c_int32), _P, C.POINTER(C.c_int32), _P]),
    "wj_whisper_align": (_I, [_P, _I, C.POINTER(C.c_int32), C.POINTER(C.c_int32), _I, C.POINTER(C.c_int32), _I,
                              C.POINTER(C.c_int32), _I, C.POINTER(C.c_int32), _I, _I, C.POINTER(C.c_int32),
                              C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(_F), _P]),
    "wj_whisper_last_align_matrix": (_I, [_P, _I, _I, _I, C.POINTER(_F)]),
    "wj_whisper_last_decode_info": (_I, [_P, C.POINTER(C.c_int32)]),
    "wj_whisper_last_beam_token_logprobs": (_I, [_P, _I, _I, C.POINTER(C.c_float)]),
    "wj_decode_open": (_I, [_P, _I, _I, _P]),
    "wj_decode_step": (_I, [_P, C.POINTER(C.c_int32), C.POINTER(C.c_int32), _I, _P]),
    "wj_decode_logits_dev": (_P, [_P]),
    "wj_decode_logits_copy": (_I, [_P, _I, _P, _P]),
    "wj_decode_topk": (_I, [_P, _I, _I, _P, C.POINTER(C.c_int32), C.POINTER(_F), C.POINTER(_F), _P]),
    "wj_decode_topk_rules": (_I, [_P, _I, _I, C.POINTER(DecodeOptsC), C.POINTER(C.c_int32), C.POINTER(C.c_int32), _I,
                                  C.POINTER(C.c_int32), _I, _F, C.POINTER(C.c_int32), C.POINTER(_F), _P]),
    "wj_decode_no_speech": (_I, [_P, _I, _I, C.POINTER(_F), _P]),
    "wj_vad_create": (_I, [_P, C.POINTER(_F), _I64, C.POINTER(_P)]),
    "wj_vad_free": (_I, [_P]),
    "wj_vad_scores": (_I, [_P, _P, C.POINTER(_I64), C.POINTER(_I64), _I, _P, _P]),
    "wj_vadg_create": (_I, [_P, C.POINTER(C.c_int32), _I, _I, C.POINTER(_F), _I64, C.POINTER(_F), _I, _I64, _I64, _I, _I, _I, _I, _I, _I, _I,
                            C.POINTER(_P)]),
    "wj_vadg_info": (_I, [_P, C.POINTER(C.c_int32)]),
    "wj_vadg_free": (_I, [_P]),
    "wj_vadg_scores": (_I, [_P, _P, C.POINTER(_I64), C.POINTER(_I64), _I, _P, _P]),
    "wj_comm_unique_id": (_I, [C.c_char_p]),
    "wj_comm_init": (_I, [_P, _I, _I, C.c_char_p, C.POINTER(_P)]),
    "wj_bcast_weights": (_I, [_P, _P, _I64, _I, _P]),
    "wj_comm_count": (_I, [_P, C.POINTER(_I)]),
    "wj_comm_destroy": (_I, [_P]),
    "wj_k_gemm": (_I, [_P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "wj_k_gemm_split": (_I, [_P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "wj_k_gemm_timed": (_I, [_P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, C.POINTER(_F)]),
    "wj_k_layernorm": (_I, [_P, _I, _P, _P, _P, _P, _I, _I, _P]),
    "wj_k_attention_enc": (_I, [_P, _I, _P, _P, _I, _I, _I, _P]),
    "wj_k_attention_enc_timed": (_I, [_P, _I, _P, _P, _I, _I, _I, _I, C.POINTER(_F)]),
    "wj_k_attention_dec": (_I, [_P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "wj_k_attention_dec_timed": (_I, [_P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, C.POINTER(_F)]),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)


def lib_path() -> Path:
    return Path(os.environ.get("WJHIP_LIB", _LIB_PATH))


def lib() -> C.CDLL:
    """Load libwjhip.so (once) and attach signatures.  Raises if it is absent or mismatched."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not path.exists():
        raise WjError(f"{path} not found: build it with `python -m whisperjav_amd.build` "
                      "(there is no CPU fallback for the HIP path)")
    handle = C.CDLL(str(path), mode=C.RTLD_GLOBAL)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(handle, name)  # AttributeError => the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    got = handle.wj_abi_version()
    if got != ABI_VERSION:
        raise WjError(f"libwjhip ABI version {got}, binding expects {ABI_VERSION}: rebuild the library")
    _lib = handle
    return handle


_TUNED: Dict[str, int] = {}       # switches set through tune() in this process (wj_tune has no getter)


def tune(key: str, value: int) -> None:
    check(lib().wj_tune(key.encode(), int(value)), "wj_tune")
    _TUNED[key] = int(value)


def tuned(key: str, default: int) -> int:
    """The value the process last set for a wj_tune switch (``default`` = the library's own default when it never did)."""
    return _TUNED.get(key, int(default))


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().wj_last_error()
        raise WjError(f"{what or 'libwjhip call'} failed (code {rc}): {msg.decode(errors='replace') if msg else ''}")


class Context:
    """One HIP context + stream on one GPU (``wj_init``)."""

    def __init__(self, device: int = 0):
        self._lib = lib()
        handle = _P()
        check(self._lib.wj_init(int(device), C.byref(handle)), "wj_init")
        self.handle = handle
        self.device = int(device)

    def sync(self) -> None:
        check(self._lib.wj_sync(self.handle), "wj_sync")

    def device_info(self) -> dict:
        out = (_I64 * 4)()
        check(self._lib.wj_device_info(self.handle, out), "wj_device_info")
        return {"cu_count": int(out[0]), "clock_khz": int(out[1]), "hbm_bytes": int(out[2]) | (int(out[3]) << 32)}

    def profile_start(self) -> None:
        check(self._lib.wj_profile_start(self.handle), "wj_profile_start")

    def profile_stop(self) -> dict:
        """{launch class: (count, total_ms)} measured with HIP events on the engine stream."""
        n = self._lib.wj_profile_tags()
        ms = (C.c_double * n)()
        cnt = (_I64 * n)()
        check(self._lib.wj_profile_stop(self.handle, ms, cnt, n), "wj_profile_stop")
        return {self._lib.wj_profile_tag_name(i).decode(): (int(cnt[i]), float(ms[i]))
                for i in range(n) if cnt[i]}

    def profile_stop_units(self) -> dict:
        """{launch class: (count, total_ms, units)}; units = windows summed over the class's launches."""
        n = self._lib.wj_profile_tags()
        ms = (C.c_double * n)()
        cnt = (_I64 * n)()
        units = (_I64 * n)()
        check(self._lib.wj_profile_stop_ex(self.handle, ms, cnt, units, n), "wj_profile_stop_ex")
        return {self._lib.wj_profile_tag_name(i).decode(): (int(cnt[i]), float(ms[i]), int(units[i]))
                for i in range(n) if cnt[i]}

    def close(self) -> None:
        if self.handle:
            self._lib.wj_shutdown(self.handle)
            self.handle = None


_contexts = {}


def context(device: int = 0) -> Context:
    """Process-wide context per device ordinal (created lazily, spawn-safe: nothing at import)."""
    ctx = _contexts.get(device)
    if ctx is None:
        ctx = _contexts[device] = Context(device)
    return ctx
