"""CPU-side checks of the C-ABI library: it builds, loads, exports every symbol that
include/wjhip.h declares, and refuses to run without a device (no silent CPU fallback)."""
import ctypes
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _declared_symbols():
    text = (ROOT / "include" / "wjhip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(wj_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(hip):
    from whisperjav_amd import hipbind
    declared = _declared_symbols()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(hip, name), f"libwjhip.so does not export {name}"
    assert set(declared) == set(hipbind.EXPORTED_SYMBOLS), set(declared) ^ set(hipbind.EXPORTED_SYMBOLS)


def test_abi_version_and_pure_helpers(hip):
    from whisperjav_amd import hipbind
    assert hip.wj_abi_version() == hipbind.ABI_VERSION == 6
    # frame arithmetic is host code: faster-whisper (N + 160) // 160, openai-whisper (N + 480000) // 160
    assert hip.wj_logmel_frames(96000, 0) == 601
    assert hip.wj_logmel_frames(480000, 0) == 3001
    assert hip.wj_logmel_frames(96000, 1) == 3600
    assert hip.wj_profile_tags() > 10 and hip.wj_profile_tag_name(0) == b"mel_to_rows"


def test_no_cpu_fallback_without_device(hip):
    import torch
    from whisperjav_amd import engine, hipbind
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible here")
    with pytest.raises(hipbind.WjError):
        engine.HipLogMel(128, "fw")
    handle = ctypes.c_void_p()
    rc = hip.wj_init(0, ctypes.byref(handle))
    assert rc != 0 and hip.wj_last_error()


def test_struct_layouts_match_header():
    from whisperjav_amd import hipbind
    assert ctypes.sizeof(hipbind.WhisperDimsC) == 40
    assert hipbind.DecodeOptsC.suppress_mask_dev.offset == 40 and ctypes.sizeof(hipbind.DecodeOptsC) == 56
    assert hipbind.DecodeOptsC.repetition_penalty.offset == 48 and hipbind.DecodeOptsC.no_repeat_ngram_size.offset == 52
