"""Whisper weights: synthetic generation, checkpoint import and the HBM blob layout.

The HIP engine consumes ONE contiguous device blob plus a table of byte offsets whose order
is fixed by ``include/wjhip.h`` (``WJ_T_*`` enumerators): matrices in the engine's compute
dtype (bf16 or fp32, row-major ``[out, in]`` = K-contiguous for both MFMA operands), vectors
(biases, LayerNorm, positional tables) always fp32, every tensor 256-byte aligned.  One blob
means one RCCL broadcast over xGMI at start-up (``whisperjav_amd.sharding``) and no per-tensor
allocation.

Weight names follow openai-whisper's state dict (``encoder.blocks.0.attn.query.weight`` ...),
which is what ``whisper.load_model`` (reference: whisperjav/modules/whisper_pro_asr.py:182)
yields; the fused tensors the engine wants (QKV, cross K/V, im2col-ordered conv kernels) are
assembled here.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np
import torch

from .dims import WhisperDims

ALIGN = 256

# ---- canonical tensor order (must match include/wjhip.h) ------------------------------------
GLOBAL_TENSORS = (
    "ENC_CONV1_W", "ENC_CONV1_B", "ENC_CONV2_W", "ENC_CONV2_B", "ENC_POS",
    "ENC_LNPOST_W", "ENC_LNPOST_B", "DEC_TOK_EMB", "DEC_POS", "DEC_LN_W", "DEC_LN_B",
)
ENC_LAYER_TENSORS = (
    "LN1_W", "LN1_B", "QKV_W", "QKV_B", "OUT_W", "OUT_B",
    "LN2_W", "LN2_B", "FC1_W", "FC1_B", "FC2_W", "FC2_B",
)
DEC_LAYER_TENSORS = (
    "LN1_W", "LN1_B", "QKV_W", "QKV_B", "OUT_W", "OUT_B",
    "LNX_W", "LNX_B", "CQ_W", "CQ_B", "CKV_W", "CKV_B", "COUT_W", "COUT_B",
    "LN2_W", "LN2_B", "FC1_W", "FC1_B", "FC2_W", "FC2_B",
)


def sinusoids(length: int, channels: int, max_timescale: float = 10000.0) -> np.ndarray:
    half = channels // 2
    inc = np.log(max_timescale) / (half - 1)
    inv = np.exp(-inc * np.arange(half, dtype=np.float32)).astype(np.float32)
    t = np.arange(length, dtype=np.float32)[:, None] * inv[None, :]
    return np.concatenate([np.sin(t), np.cos(t)], axis=1).astype(np.float32)


def round_to_f16(a: np.ndarray) -> np.ndarray:
    """fp32 array rounded (RNE) to the nearest fp16-representable fp32 value (published Whisper checkpoints are
    stored in fp16, so this is what real weights look like)."""
    t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    return t.to(torch.float16).to(torch.float32).numpy()


def round_to_bf16(a: np.ndarray) -> np.ndarray:
    """fp32 array rounded (RNE) to the nearest bf16-representable fp32 value."""
    t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    return t.to(torch.bfloat16).to(torch.float32).numpy()


def synth_weights(dims: WhisperDims, seed: int = 1234, bf16_exact: bool = True,
                  exact: str = "") -> Dict[str, np.ndarray]:
    """Seeded random weights with trained-like statistics (there are no checkpoints offline).

    Variances are chosen so activations stay O(1) through the stack, attention scores have
    unit-ish spread and logits have a std of ~1.8 over the vocabulary (top-2 gap ~0.4), so
    greedy/beam decisions are numerically well separated -- a uniform-logit random model would
    make token parity meaningless.  With ``bf16_exact`` every matrix is pre-rounded to bf16 so
    the fp32 oracle and the bf16 engine see bit-identical parameters.  ``exact`` overrides it:
    ``"none"`` keeps the raw fp32 draws (a 16-bit engine then also carries weight-rounding error),
    ``"float16"`` rounds to fp16 (the storage type of the published checkpoints), ``"bfloat16"`` to bf16.
    The random draws are the same in every mode.
    """
    rng = np.random.default_rng(seed)
    w: Dict[str, np.ndarray] = {}
    mode = exact or ("bfloat16" if bf16_exact else "none")
    if mode not in ("none", "float16", "bfloat16"):
        raise ValueError("exact must be '', 'none', 'float16' or 'bfloat16'")
    rnd = {"none": (lambda a: a), "float16": round_to_f16, "bfloat16": round_to_bf16}[mode]

    def mat(name, shape, fan_in, gain=1.0):
        a = rng.standard_normal(shape, dtype=np.float32) * np.float32(gain / np.sqrt(fan_in))
        w[name] = rnd(a)

    def vec(name, n, scale=0.1, base=0.0):
        w[name] = (base + scale * rng.standard_normal(n, dtype=np.float32)).astype(np.float32)

    d = dims.n_audio_state
    mat("encoder.conv1.weight", (d, dims.n_mels, 3), 3 * dims.n_mels)
    vec("encoder.conv1.bias", d)
    mat("encoder.conv2.weight", (d, d, 3), 3 * d)
    vec("encoder.conv2.bias", d)
    w["encoder.positional_embedding"] = sinusoids(dims.n_audio_ctx, d)

    def block(prefix, dm, cross):
        vec(prefix + "attn_ln.weight", dm, 0.1, 1.0)
        vec(prefix + "attn_ln.bias", dm)
        names = ["attn"] + (["cross_attn"] if cross else [])
        for a in names:
            mat(f"{prefix}{a}.query.weight", (dm, dm), dm)
            vec(f"{prefix}{a}.query.bias", dm)
            mat(f"{prefix}{a}.key.weight", (dm, dm), dm)
            mat(f"{prefix}{a}.value.weight", (dm, dm), dm)
            vec(f"{prefix}{a}.value.bias", dm)
            mat(f"{prefix}{a}.out.weight", (dm, dm), dm, 0.5)
            vec(f"{prefix}{a}.out.bias", dm)
        if cross:
            vec(prefix + "cross_attn_ln.weight", dm, 0.1, 1.0)
            vec(prefix + "cross_attn_ln.bias", dm)
        vec(prefix + "mlp_ln.weight", dm, 0.1, 1.0)
        vec(prefix + "mlp_ln.bias", dm)
        mat(prefix + "mlp.0.weight", (4 * dm, dm), dm)
        vec(prefix + "mlp.0.bias", 4 * dm)
        mat(prefix + "mlp.2.weight", (dm, 4 * dm), 4 * dm, 0.5)
        vec(prefix + "mlp.2.bias", dm)

    for i in range(dims.n_audio_layer):
        block(f"encoder.blocks.{i}.", d, False)
    vec("encoder.ln_post.weight", d, 0.1, 1.0)
    vec("encoder.ln_post.bias", d)

    dt = dims.n_text_state
    a = rng.standard_normal((dims.n_vocab, dt), dtype=np.float32) * np.float32(0.05)
    w["decoder.token_embedding.weight"] = rnd(a)
    w["decoder.positional_embedding"] = (0.05 * rng.standard_normal(
        (dims.n_text_ctx, dt), dtype=np.float32)).astype(np.float32)
    for i in range(dims.n_text_layer):
        block(f"decoder.blocks.{i}.", dt, True)
    vec("decoder.ln.weight", dt, 0.1, 1.0)
    vec("decoder.ln.bias", dt)
    return w


# ---- blob packing ---------------------------------------------------------------------------
def _im2col_conv(wt: np.ndarray) -> np.ndarray:
    """Conv1d kernel ``[out, in, 3]`` -> GEMM operand ``[out, 3*in]`` with K index = tap*in + c."""
    return np.ascontiguousarray(wt.transpose(0, 2, 1).reshape(wt.shape[0], -1))


def engine_tensors(dims: WhisperDims, w: Dict[str, np.ndarray]) -> List[Tuple[str, np.ndarray, bool]]:
    """Flatten a state dict into the canonical ``(label, array, is_matrix)`` list."""
    out: List[Tuple[str, np.ndarray, bool]] = []
    d = dims.n_audio_state
    zeros = lambda n: np.zeros(n, dtype=np.float32)  # noqa: E731

    g = {
        "ENC_CONV1_W": (_im2col_conv(w["encoder.conv1.weight"]), True),
        "ENC_CONV1_B": (w["encoder.conv1.bias"], False),
        "ENC_CONV2_W": (_im2col_conv(w["encoder.conv2.weight"]), True),
        "ENC_CONV2_B": (w["encoder.conv2.bias"], False),
        "ENC_POS": (w["encoder.positional_embedding"], False),
        "ENC_LNPOST_W": (w["encoder.ln_post.weight"], False),
        "ENC_LNPOST_B": (w["encoder.ln_post.bias"], False),
        "DEC_TOK_EMB": (w["decoder.token_embedding.weight"], True),
        "DEC_POS": (w["decoder.positional_embedding"], False),
        "DEC_LN_W": (w["decoder.ln.weight"], False),
        "DEC_LN_B": (w["decoder.ln.bias"], False),
    }
    for name in GLOBAL_TENSORS:
        arr, is_mat = g[name]
        out.append((name, arr, is_mat))

    def attn_fused(p):
        qkv_w = np.concatenate([w[p + "query.weight"], w[p + "key.weight"], w[p + "value.weight"]], 0)
        qkv_b = np.concatenate([w[p + "query.bias"], zeros(w[p + "key.weight"].shape[0]),
                                w[p + "value.bias"]], 0)
        return qkv_w, qkv_b

    for i in range(dims.n_audio_layer):
        p = f"encoder.blocks.{i}."
        qkv_w, qkv_b = attn_fused(p + "attn.")
        t = {
            "LN1_W": (w[p + "attn_ln.weight"], False), "LN1_B": (w[p + "attn_ln.bias"], False),
            "QKV_W": (qkv_w, True), "QKV_B": (qkv_b, False),
            "OUT_W": (w[p + "attn.out.weight"], True), "OUT_B": (w[p + "attn.out.bias"], False),
            "LN2_W": (w[p + "mlp_ln.weight"], False), "LN2_B": (w[p + "mlp_ln.bias"], False),
            "FC1_W": (w[p + "mlp.0.weight"], True), "FC1_B": (w[p + "mlp.0.bias"], False),
            "FC2_W": (w[p + "mlp.2.weight"], True), "FC2_B": (w[p + "mlp.2.bias"], False),
        }
        for name in ENC_LAYER_TENSORS:
            out.append((f"enc{i}.{name}", t[name][0], t[name][1]))

    for i in range(dims.n_text_layer):
        p = f"decoder.blocks.{i}."
        qkv_w, qkv_b = attn_fused(p + "attn.")
        ckv_w = np.concatenate([w[p + "cross_attn.key.weight"], w[p + "cross_attn.value.weight"]], 0)
        ckv_b = np.concatenate([zeros(d), w[p + "cross_attn.value.bias"]], 0)
        t = {
            "LN1_W": (w[p + "attn_ln.weight"], False), "LN1_B": (w[p + "attn_ln.bias"], False),
            "QKV_W": (qkv_w, True), "QKV_B": (qkv_b, False),
            "OUT_W": (w[p + "attn.out.weight"], True), "OUT_B": (w[p + "attn.out.bias"], False),
            "LNX_W": (w[p + "cross_attn_ln.weight"], False), "LNX_B": (w[p + "cross_attn_ln.bias"], False),
            "CQ_W": (w[p + "cross_attn.query.weight"], True), "CQ_B": (w[p + "cross_attn.query.bias"], False),
            "CKV_W": (ckv_w, True), "CKV_B": (ckv_b, False),
            "COUT_W": (w[p + "cross_attn.out.weight"], True), "COUT_B": (w[p + "cross_attn.out.bias"], False),
            "LN2_W": (w[p + "mlp_ln.weight"], False), "LN2_B": (w[p + "mlp_ln.bias"], False),
            "FC1_W": (w[p + "mlp.0.weight"], True), "FC1_B": (w[p + "mlp.0.bias"], False),
            "FC2_W": (w[p + "mlp.2.weight"], True), "FC2_B": (w[p + "mlp.2.bias"], False),
        }
        for name in DEC_LAYER_TENSORS:
            out.append((f"dec{i}.{name}", t[name][0], t[name][1]))
    return out


def expected_tensor_count(dims: WhisperDims) -> int:
    return (len(GLOBAL_TENSORS) + len(ENC_LAYER_TENSORS) * dims.n_audio_layer
            + len(DEC_LAYER_TENSORS) * dims.n_text_layer)


def pack_blob(dims: WhisperDims, w: Dict[str, np.ndarray], dtype: str = "bfloat16"
              ) -> Tuple[torch.Tensor, np.ndarray]:
    """Pack a state dict into (host uint8 blob, int64 byte offsets) for ``wj_whisper_create``.

    ``dtype``: ``"bfloat16"`` / ``"float16"`` (matrices stored in that 16-bit type) or ``"float32"``.
    """
    if dtype not in ("bfloat16", "float16", "float32"):
        raise ValueError("dtype must be 'bfloat16', 'float16' or 'float32'")
    half = {"bfloat16": torch.bfloat16, "float16": torch.float16}.get(dtype)
    tensors = engine_tensors(dims, w)
    assert len(tensors) == expected_tensor_count(dims)
    offsets = np.zeros(len(tensors), dtype=np.int64)
    cursor = 0
    sizes = []
    for idx, (_, arr, is_mat) in enumerate(tensors):
        item = 2 if (is_mat and half is not None) else 4
        nbytes = int(arr.size) * item
        offsets[idx] = cursor
        sizes.append(nbytes)
        cursor += (nbytes + ALIGN - 1) // ALIGN * ALIGN
    blob = torch.zeros(cursor, dtype=torch.uint8)
    for (label, arr, is_mat), off, nbytes in zip(tensors, offsets, sizes):
        t = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float32))
        if is_mat and half is not None:
            t = t.to(half)
        blob[off:off + nbytes] = t.reshape(-1).view(torch.uint8)
    return blob, offsets


def pack_blob_device(dims: WhisperDims, w: Dict[str, np.ndarray], dtype: str, device) -> Tuple[torch.Tensor, np.ndarray]:
    """``pack_blob`` with the type conversion done on the GPU: every tensor is uploaded as it is (fp32 over PCIe) and
    rounded / laid out into the device blob there -- seconds of single-threaded CPU conversion become milliseconds
    (bench start-up).  Same bytes as ``pack_blob(...)[0].to(device)``."""
    if dtype not in ("bfloat16", "float16", "float32"):
        raise ValueError("dtype must be 'bfloat16', 'float16' or 'float32'")
    half = {"bfloat16": torch.bfloat16, "float16": torch.float16}.get(dtype)
    tensors = engine_tensors(dims, w)
    offsets = np.zeros(len(tensors), dtype=np.int64)
    cursor, sizes = 0, []
    for idx, (_, arr, is_mat) in enumerate(tensors):
        nbytes = int(arr.size) * (2 if (is_mat and half is not None) else 4)
        offsets[idx] = cursor
        sizes.append(nbytes)
        cursor += (nbytes + ALIGN - 1) // ALIGN * ALIGN
    blob = torch.zeros(cursor, dtype=torch.uint8, device=device)
    for (_, arr, is_mat), off, nbytes in zip(tensors, offsets, sizes):
        t = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float32)).to(device, non_blocking=True)
        if is_mat and half is not None:
            t = t.to(half)
        blob[off:off + nbytes] = t.reshape(-1).view(torch.uint8)
    return blob, offsets


def load_openai_checkpoint(path: str) -> Tuple[WhisperDims, Dict[str, np.ndarray]]:
    """Import an openai-whisper ``.pt`` checkpoint (``{"dims": ..., "model_state_dict": ...}``)."""
    ckpt = torch.load(path, map_location="cpu", weights_only=True)
    dims = WhisperDims(**{k: int(v) for k, v in ckpt["dims"].items()})
    sd = {k: v.float().numpy() for k, v in ckpt["model_state_dict"].items()}
    return dims, sd


def load_hf_checkpoint(path: str) -> Tuple[WhisperDims, Dict[str, np.ndarray], Dict[str, object]]:
    """Import a Hugging Face ``WhisperForConditionalGeneration`` directory (``config.json`` + ``model.safetensors`` or
    its sharded form, e.g. a local copy of openai/whisper-large-v3).  Returns ``(dims, openai-named state dict,
    extras)`` where extras may hold ``alignment_heads`` from ``generation_config.json``."""
    import json
    import os
    from safetensors.torch import load_file
    with open(os.path.join(path, "config.json")) as f:
        cfg = json.load(f)
    dims = WhisperDims(n_mels=int(cfg["num_mel_bins"]), n_audio_ctx=int(cfg["max_source_positions"]),
                       n_audio_state=int(cfg["d_model"]), n_audio_head=int(cfg["encoder_attention_heads"]),
                       n_audio_layer=int(cfg["encoder_layers"]), n_vocab=int(cfg["vocab_size"]),
                       n_text_ctx=int(cfg["max_target_positions"]), n_text_state=int(cfg["d_model"]),
                       n_text_head=int(cfg["decoder_attention_heads"]), n_text_layer=int(cfg["decoder_layers"]))
    index = os.path.join(path, "model.safetensors.index.json")
    if os.path.exists(index):
        with open(index) as f:
            files = sorted(set(json.load(f)["weight_map"].values()))
    else:
        files = ["model.safetensors"]
    hf: Dict[str, np.ndarray] = {}
    for name in files:
        for k, v in load_file(os.path.join(path, name)).items():      # any stored dtype (fp16 / bf16 / fp32)
            hf[k[6:] if k.startswith("model.") else k] = v.float().numpy()
    sd: Dict[str, np.ndarray] = {}

    def take(dst: str, src: str, required: bool = True) -> None:
        if src in hf:
            sd[dst] = np.ascontiguousarray(hf[src], dtype=np.float32)
        elif required:
            raise KeyError(f"{path}: tensor {src!r} missing (not a Whisper checkpoint?)")

    for side in ("encoder", "decoder"):
        take(f"{side}.positional_embedding", f"{side}.embed_positions.weight")
    for n in ("conv1", "conv2"):
        take(f"encoder.{n}.weight", f"encoder.{n}.weight")
        take(f"encoder.{n}.bias", f"encoder.{n}.bias")
    take("encoder.ln_post.weight", "encoder.layer_norm.weight")
    take("encoder.ln_post.bias", "encoder.layer_norm.bias")
    take("decoder.token_embedding.weight", "decoder.embed_tokens.weight")
    take("decoder.ln.weight", "decoder.layer_norm.weight")
    take("decoder.ln.bias", "decoder.layer_norm.bias")

    def attn(dst: str, src: str) -> None:
        for a, b in (("query", "q_proj"), ("key", "k_proj"), ("value", "v_proj"), ("out", "out_proj")):
            take(f"{dst}.{a}.weight", f"{src}.{b}.weight")
            take(f"{dst}.{a}.bias", f"{src}.{b}.bias", required=(a != "key"))

    def pair(dst: str, src: str) -> None:
        take(dst + ".weight", src + ".weight")
        take(dst + ".bias", src + ".bias")

    for i in range(dims.n_audio_layer):
        d, h = f"encoder.blocks.{i}", f"encoder.layers.{i}"
        attn(d + ".attn", h + ".self_attn")
        pair(d + ".attn_ln", h + ".self_attn_layer_norm")
        pair(d + ".mlp_ln", h + ".final_layer_norm")
        pair(d + ".mlp.0", h + ".fc1")
        pair(d + ".mlp.2", h + ".fc2")
    for i in range(dims.n_text_layer):
        d, h = f"decoder.blocks.{i}", f"decoder.layers.{i}"
        attn(d + ".attn", h + ".self_attn")
        attn(d + ".cross_attn", h + ".encoder_attn")
        pair(d + ".attn_ln", h + ".self_attn_layer_norm")
        pair(d + ".cross_attn_ln", h + ".encoder_attn_layer_norm")
        pair(d + ".mlp_ln", h + ".final_layer_norm")
        pair(d + ".mlp.0", h + ".fc1")
        pair(d + ".mlp.2", h + ".fc2")
    extras: Dict[str, object] = {}
    gen = os.path.join(path, "generation_config.json")
    if os.path.exists(gen):
        with open(gen) as f:
            heads = json.load(f).get("alignment_heads")
        if heads:
            extras["alignment_heads"] = [(int(a), int(b)) for a, b in heads]
    return dims, sd, extras
