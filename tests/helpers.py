"""Shared helpers for the parity tests (bridges product-side and oracle-side types)."""
from __future__ import annotations

import numpy as np
import torch

from oracle import whisper_ref
from whisperjav_amd import dims as pdims
from whisperjav_amd import weights as pweights


def oracle_dims(d: pdims.WhisperDims) -> whisper_ref.WhisperDims:
    return whisper_ref.WhisperDims(**d.as_dict())


def small_dims(n_mels=80, d_model=128, heads=2, layers=2, n_vocab=51865) -> pdims.WhisperDims:
    return pdims.custom_dims(n_mels, d_model, heads, layers, n_vocab)


def make_oracle(d: pdims.WhisperDims, seed=1234, emulate_bf16=False, emulate: str = "", eot=None, cross_gain: float = 1.0,
                exact: str = "", logit_std: float = 0.0):
    """``emulate`` = "bfloat16" / "float16": the oracle rounds every GEMM / attention operand to that type at the
    engine's rounding points (fp32 accumulation), so a 16-bit engine can be checked against something tighter than
    "fp32 +- rounding noise"; "" / "float32" = the plain fp32 oracle.  ``eot`` / ``cross_gain``: the end-of-text ramp and
    the peaked cross-attention of ``weights.synth_weights`` (hypotheses then END, at window-dependent lengths)."""
    w = pweights.synth_weights(d, seed=seed, eot=eot, cross_gain=cross_gain, exact=exact, logit_std=logit_std)
    if emulate_bf16:
        emulate = "bfloat16"
    rnd = {"bfloat16": whisper_ref.bf16_round, "float16": whisper_ref.f16_round}.get(emulate)
    return whisper_ref.WhisperOracle(oracle_dims(d), w, act_round=rnd), w


import functools


@functools.lru_cache(maxsize=4)
def cached_weights(dims: pdims.WhisperDims, seed: int, exact: str = "", eot=None, cross_gain: float = 1.0, logit_std: float = 0.0):
    """Seeded synthetic weights, generated once per (geometry, seed, rounding mode) and test session: a large-v3 set
    is 1.5 G normal draws (~30 s of single-threaded NumPy) and nine GPU tests want one of three of them.  Callers
    must not modify the arrays."""
    return pweights.synth_weights(dims, seed=seed, exact=exact, eot=eot, cross_gain=cross_gain, logit_std=logit_std)


def hf_state_dict(d: pdims.WhisperDims, w):
    """Map openai-style names onto transformers' WhisperForConditionalGeneration names."""
    sd = {}
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))  # noqa: E731
    sd["model.encoder.conv1.weight"] = t(w["encoder.conv1.weight"])
    sd["model.encoder.conv1.bias"] = t(w["encoder.conv1.bias"])
    sd["model.encoder.conv2.weight"] = t(w["encoder.conv2.weight"])
    sd["model.encoder.conv2.bias"] = t(w["encoder.conv2.bias"])
    sd["model.encoder.embed_positions.weight"] = t(w["encoder.positional_embedding"])
    sd["model.encoder.layer_norm.weight"] = t(w["encoder.ln_post.weight"])
    sd["model.encoder.layer_norm.bias"] = t(w["encoder.ln_post.bias"])
    sd["model.decoder.embed_tokens.weight"] = t(w["decoder.token_embedding.weight"])
    sd["model.decoder.embed_positions.weight"] = t(w["decoder.positional_embedding"])
    sd["model.decoder.layer_norm.weight"] = t(w["decoder.ln.weight"])
    sd["model.decoder.layer_norm.bias"] = t(w["decoder.ln.bias"])
    sd["proj_out.weight"] = sd["model.decoder.embed_tokens.weight"]

    def attn(src, dst):
        for a, b in (("query", "q_proj"), ("key", "k_proj"), ("value", "v_proj"), ("out", "out_proj")):
            sd[f"{dst}.{b}.weight"] = t(w[f"{src}.{a}.weight"])
            if a != "key":
                sd[f"{dst}.{b}.bias"] = t(w[f"{src}.{a}.bias"])

    def ln(src, dst):
        sd[dst + ".weight"] = t(w[src + ".weight"])
        sd[dst + ".bias"] = t(w[src + ".bias"])

    for i in range(d.n_audio_layer):
        s, h = f"encoder.blocks.{i}", f"model.encoder.layers.{i}"
        attn(s + ".attn", h + ".self_attn")
        ln(s + ".attn_ln", h + ".self_attn_layer_norm")
        ln(s + ".mlp_ln", h + ".final_layer_norm")
        ln(s + ".mlp.0", h + ".fc1")
        ln(s + ".mlp.2", h + ".fc2")
    for i in range(d.n_text_layer):
        s, h = f"decoder.blocks.{i}", f"model.decoder.layers.{i}"
        attn(s + ".attn", h + ".self_attn")
        attn(s + ".cross_attn", h + ".encoder_attn")
        ln(s + ".attn_ln", h + ".self_attn_layer_norm")
        ln(s + ".cross_attn_ln", h + ".encoder_attn_layer_norm")
        ln(s + ".mlp_ln", h + ".final_layer_norm")
        ln(s + ".mlp.0", h + ".fc1")
        ln(s + ".mlp.2", h + ".fc2")
    return sd


def synth_mel(batch, n_mels, seed=7, frames=3000):
    """Log-mel-shaped random input in the value range Whisper sees ([-1, 1.5])."""
    rng = np.random.default_rng(seed)
    return (rng.standard_normal((batch, n_mels, frames), dtype=np.float32) * 0.4).clip(-1.0, 1.5)


def scripted_segments(n_samples: int):
    """The "model" of the ASR-adapter fixtures: segments that depend only on the clip length (the reference side
    in tests/golden/make_asr_adapter_fixtures.py and the product side in tests/test_asr_adapter.py call the same
    function, so any difference in the results comes from the adapter logic around the model call)."""
    dur = n_samples / 16000.0
    k = n_samples % 7
    segs = [dict(id=1, seek=0, start=round(0.05 * dur, 3), end=round(0.45 * dur, 3), text=f" 台詞{n_samples % 1000} ", tokens=[1, 2],
                 avg_logprob=-0.2 - 0.1 * k, compression_ratio=1.0, no_speech_prob=0.01, temperature=0.0),
            dict(id=2, seek=0, start=round(0.5 * dur, 3), end=round(0.6 * dur, 3), text="ご視聴ありがとうございました", tokens=[3],
                 avg_logprob=-0.1, compression_ratio=1.0, no_speech_prob=0.01, temperature=0.0),
            dict(id=3, seek=0, start=round(0.62 * dur, 3), end=round(0.7 * dur, 3), text="Thank you", tokens=[4],
                 avg_logprob=-0.9, compression_ratio=1.0, no_speech_prob=0.01, temperature=0.0),
            dict(id=4, seek=0, start=round(0.72 * dur, 3), end=round(0.8 * dur, 3), text="あっ…", tokens=[5],
                 avg_logprob=-0.45, compression_ratio=1.0, no_speech_prob=0.01, temperature=0.0),
            dict(id=5, seek=0, start=round(0.82 * dur, 3), end=round(0.82 * dur, 3) + 0.0, text="   ", tokens=[6],
                 avg_logprob=-0.3, compression_ratio=1.0, no_speech_prob=0.01, temperature=0.0),
            dict(id=6, seek=0, start=round(0.85 * dur, 3), end=round(0.99 * dur, 3), text="低い確率の行", tokens=[7],
                 avg_logprob=-1.4 + 0.15 * k, compression_ratio=1.0, no_speech_prob=0.01, temperature=0.0)]
    return segs
