"""Whisper model dimensions and special-token layout (host side, no torch dependency).

Mirrors the fields of openai-whisper's ``ModelDimensions`` / faster-whisper's
``WhisperModel`` config that WhisperJAV's ASR wrappers rely on
(/root/reference/whisperjav/modules/faster_whisper_pro_asr.py:247-253 loads by model
*name*; sizes are upstream's published table).
"""
from __future__ import annotations

from dataclasses import dataclass, asdict
from typing import Dict


@dataclass(frozen=True)
class WhisperDims:
    n_mels: int
    n_audio_ctx: int
    n_audio_state: int
    n_audio_head: int
    n_audio_layer: int
    n_vocab: int
    n_text_ctx: int
    n_text_state: int
    n_text_head: int
    n_text_layer: int

    def as_dict(self) -> Dict[str, int]:
        return asdict(self)

    @property
    def head_dim(self) -> int:
        return self.n_audio_state // self.n_audio_head


_TABLE = {
    # name: (n_mels, d_model, heads, layers, vocab)
    "tiny": (80, 384, 6, 4, 51865),
    "base": (80, 512, 8, 6, 51865),
    "small": (80, 768, 12, 12, 51865),
    "medium": (80, 1024, 16, 24, 51865),
    "large-v1": (80, 1280, 20, 32, 51865),
    "large-v2": (80, 1280, 20, 32, 51865),
    "large-v3": (128, 1280, 20, 32, 51866),
    "large": (128, 1280, 20, 32, 51866),
}


def dims_for(name: str) -> WhisperDims:
    """Dimensions of a named multilingual Whisper checkpoint."""
    key = name.lower()
    if key not in _TABLE:
        raise KeyError(f"unknown Whisper model name {name!r}; known: {sorted(_TABLE)}")
    m, d, h, l, v = _TABLE[key]
    return WhisperDims(m, 1500, d, h, l, v, 448, d, h, l)


def custom_dims(n_mels: int, d_model: int, heads: int, layers: int, n_vocab: int = 51865,
                n_audio_ctx: int = 1500, n_text_ctx: int = 448) -> WhisperDims:
    """Arbitrary-size Whisper (used by the parity tests for second-scale oracles)."""
    if d_model % heads or d_model // heads != 64:
        raise ValueError("the HIP engine is specialised for head_dim == 64 (all Whisper sizes)")
    return WhisperDims(n_mels, n_audio_ctx, d_model, heads, layers, n_vocab, n_text_ctx,
                       d_model, heads, layers)


@dataclass(frozen=True)
class SpecialTokens:
    """Special-token ids of the multilingual Whisper vocabulary (``whisper.tokenizer``)."""
    eot: int
    sot: int
    translate: int
    transcribe: int
    sot_lm: int
    sot_prev: int
    no_speech: int
    no_timestamps: int
    timestamp_begin: int
    num_languages: int
    blank: int = 220  # the single-space token " " that SuppressBlank masks

    def language_token(self, index: int) -> int:
        if not 0 <= index < self.num_languages:
            raise ValueError("language index out of range")
        return self.sot + 1 + index


# language codes in upstream order (whisper.tokenizer.LANGUAGES keys)
LANGUAGE_CODES = (
    "en zh de es ru ko fr ja pt tr pl ca nl ar sv it id hi fi vi he uk el ms cs ro da hu ta no "
    "th ur hr bg lt la mi ml cy sk te fa lv bn sr az sl kn et mk br eu is hy ne mn bs kk sq sw "
    "gl mr pa si km sn yo so af oc ka be tg sd gu am yi lo uz fo ht ps tk nn mt sa lb my bo tl "
    "mg as tt haw ln ha ba jw su yue"
).split()


def special_tokens(n_vocab: int) -> SpecialTokens:
    """Derive the special-token ids from the vocabulary size (51865 -> 99 languages, 51866 -> 100)."""
    num_lang = n_vocab - 51765 - 1
    if num_lang < 1 or num_lang > len(LANGUAGE_CODES):
        raise ValueError(f"not a multilingual Whisper vocabulary size: {n_vocab}")
    eot = 50257
    sot = 50258
    translate = sot + 1 + num_lang
    return SpecialTokens(eot=eot, sot=sot, translate=translate, transcribe=translate + 1,
                         sot_lm=translate + 2, sot_prev=translate + 3, no_speech=translate + 4,
                         no_timestamps=translate + 5, timestamp_begin=translate + 6,
                         num_languages=num_lang)


def language_index(code: str) -> int:
    return LANGUAGE_CODES.index(code)
