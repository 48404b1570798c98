"""CTranslate2 model directories (``model.bin`` + ``config.json`` + ``vocabulary.json`` / ``tokenizer.json``).

The reference's balanced mode opens its model with ``faster_whisper.WhisperModel(model_size_or_path=...)``
(/root/reference/whisperjav/modules/faster_whisper_pro_asr.py:246-253), i.e. a CTranslate2 conversion of Whisper
(``Systran/faster-whisper-large-v3`` and friends in the Hugging Face cache, or a local directory).  This module reads
that format and hands the tensors to the same packer every other loader feeds (``weights.engine_tensors`` /
``pack_blob``), so ``HipWhisperModel("/path/to/faster-whisper-large-v3")`` opens what the reference's users have.

Format (ctranslate2 4.x, ``python/ctranslate2/specs/model_spec.py`` ``ModelSpec._serialize`` and ``src/models/model.cc``
``Model::load`` -- restated from the published sources; the wheel is not installable here, the round trip is pinned by this
module's own writer in tests/test_ct2_format.py and by a wheel-gated test against ``ctranslate2``'s converter):

    u32  binary version (6)                      | strings: u16 length INCLUDING the trailing NUL, bytes, NUL
    str  spec name ("WhisperSpec")               | type ids: 0 float32, 1 int8, 2 int16, 3 int32, 4 float16, 5 bfloat16
    u32  spec revision (3 for Whisper)
    u32  number of variables, then per variable:  str name, u8 rank, u32 dims[rank], u8 type id, u32 byte count, raw data
    u32  number of aliases, then per alias:       str alias, str variable name      (binary version >= 3)

Variables are named by their path in the spec tree (``encoder/layer_3/self_attention/linear_0/weight``); quantised matrices
carry a ``<name>_scale`` companion (int8: one float per output row, ``q = round(w * scale)``; int16: one scalar).  Whisper's
tree: ``WhisperSpec`` -> ``encoder`` (``conv1``, ``conv2``, ``position_encodings/encodings``, ``layer_norm``, ``layer_i`` with
``self_attention`` {``layer_norm``, ``linear_0`` = fused q;k;v, ``linear_1`` = out} and ``ffn`` {``layer_norm``, ``linear_0``,
``linear_1``}) and ``decoder`` (``embeddings/weight``, ``position_encodings/encodings``, ``layer_norm``, ``projection/weight``
(an alias of the embedding), ``layer_i`` with ``self_attention`` as above, ``attention`` {``layer_norm``, ``linear_0`` = q,
``linear_1`` = fused k;v, ``linear_2`` = out} and ``ffn``), plus scalar bookkeeping (``num_heads`` ...).  The k projections
have no bias in Whisper; the converter's ``fuse_linear`` stores zeros for them.
"""
from __future__ import annotations

import json
import os
import struct
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np

from .dims import WhisperDims

BINARY_VERSION = 6
_TYPE_IDS = ("float32", "int8", "int16", "int32", "float16", "bfloat16")
_ITEM = {"float32": 4, "int8": 1, "int16": 2, "int32": 4, "float16": 2, "bfloat16": 2}


class Ct2FormatError(ValueError):
    """The file is not a CTranslate2 model this reader understands (message says which field)."""


def _bf16_to_f32(raw: np.ndarray) -> np.ndarray:
    return (raw.astype(np.uint32) << 16).view(np.float32)


def _f32_to_bf16(a: np.ndarray) -> np.ndarray:
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
    return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)        # round to nearest even


class _Reader:
    def __init__(self, path: str):
        self.path = path
        self.buf = np.memmap(path, dtype=np.uint8, mode="r")
        self.pos = 0

    def take(self, n: int) -> np.ndarray:
        if self.pos + n > self.buf.shape[0]:
            raise Ct2FormatError(f"{self.path}: truncated at byte {self.pos} (wanted {n} more of {self.buf.shape[0]})")
        out = self.buf[self.pos: self.pos + n]
        self.pos += n
        return out

    def u8(self) -> int:
        return int(self.take(1)[0])

    def u16(self) -> int:
        return struct.unpack("<H", self.take(2).tobytes())[0]

    def u32(self) -> int:
        return struct.unpack("<I", self.take(4).tobytes())[0]

    def string(self) -> str:
        n = self.u16()
        if n < 1:
            raise Ct2FormatError(f"{self.path}: empty string field at byte {self.pos}")
        raw = self.take(n).tobytes()
        if raw[-1] != 0:
            raise Ct2FormatError(f"{self.path}: string at byte {self.pos - n} is not NUL terminated")
        return raw[:-1].decode("utf-8")


def read_model_bin(path: str) -> Tuple[Dict[str, np.ndarray], Dict[str, str], Dict[str, object]]:
    """``(variables, aliases, header)``: every variable as a NumPy array in its STORED type (bfloat16 as uint16 bit patterns,
    flagged in ``header["bfloat16"]``), alias name -> variable name, header = binary version / spec name / revision."""
    r = _Reader(path)
    version = r.u32()
    if not 4 <= version <= BINARY_VERSION:
        raise Ct2FormatError(f"{path}: binary version {version}; this reader covers 4..{BINARY_VERSION} (typed tensors)")
    spec = r.string()
    revision = r.u32()
    n_vars = r.u32()
    variables: Dict[str, np.ndarray] = {}
    bf16: List[str] = []
    for _ in range(n_vars):
        name = r.string()
        rank = r.u8()
        shape = tuple(r.u32() for _ in range(rank))
        type_id = r.u8()
        nbytes = r.u32()
        if type_id >= len(_TYPE_IDS):
            raise Ct2FormatError(f"{path}: variable {name!r} has unknown type id {type_id}")
        tname = _TYPE_IDS[type_id]
        count = int(np.prod(shape, dtype=np.int64)) if rank else 1
        if count * _ITEM[tname] != nbytes:
            raise Ct2FormatError(f"{path}: variable {name!r} shape {shape} x {tname} != {nbytes} bytes")
        raw = r.take(nbytes)
        np_t = np.uint16 if tname == "bfloat16" else np.dtype(tname)
        variables[name] = np.frombuffer(raw, dtype=np_t).reshape(shape)
        if tname == "bfloat16":
            bf16.append(name)
    aliases: Dict[str, str] = {}
    if r.pos < r.buf.shape[0]:
        for _ in range(r.u32()):
            alias = r.string()
            aliases[alias] = r.string()
    for alias, target in aliases.items():
        if target not in variables:
            raise Ct2FormatError(f"{path}: alias {alias!r} names the missing variable {target!r}")
    return variables, aliases, {"binary_version": version, "spec": spec, "revision": revision, "bfloat16": bf16}


def write_model_bin(path: str, variables: Dict[str, np.ndarray], aliases: Optional[Dict[str, str]] = None, spec: str = "WhisperSpec",
                    revision: int = 3, bfloat16: Iterable[str] = ()) -> None:
    """The converter's serialisation (sorted names, as ``ModelSpec.variables(ordered=True)``).  ``bfloat16``: names whose uint16
    arrays are bfloat16 bit patterns."""
    bf = set(bfloat16)

    def wstr(f, s: str) -> None:
        raw = s.encode("utf-8")
        f.write(struct.pack("<H", len(raw) + 1))
        f.write(raw)
        f.write(b"\0")

    with open(path, "wb") as f:
        f.write(struct.pack("<I", BINARY_VERSION))
        wstr(f, spec)
        f.write(struct.pack("<I", int(revision)))
        f.write(struct.pack("<I", len(variables)))
        for name in sorted(variables):
            a = np.asarray(variables[name])
            tname = "bfloat16" if name in bf else str(a.dtype)
            if tname not in _TYPE_IDS:
                raise Ct2FormatError(f"variable {name!r}: dtype {a.dtype} has no CTranslate2 type id")
            wstr(f, name)
            f.write(struct.pack("<B", a.ndim))
            for dim in a.shape:
                f.write(struct.pack("<I", int(dim)))
            f.write(struct.pack("<B", _TYPE_IDS.index(tname)))
            f.write(struct.pack("<I", int(a.size) * _ITEM[tname]))
            f.write(np.ascontiguousarray(a).tobytes())
        aliases = aliases or {}
        f.write(struct.pack("<I", len(aliases)))
        for alias in sorted(aliases):
            wstr(f, alias)
            wstr(f, aliases[alias])


def _as_f32(variables: Dict[str, np.ndarray], aliases: Dict[str, str], bf16: Sequence[str], name: str) -> np.ndarray:
    """Variable ``name`` (or its alias target) as float32, dequantised when a ``_scale`` companion exists."""
    key = name if name in variables else aliases.get(name)
    if key is None or key not in variables:
        raise KeyError(name)
    a = variables[key]
    if key in bf16:
        return _bf16_to_f32(a)
    if a.dtype in (np.int8, np.int16):
        skey = key + "_scale"
        skey = skey if skey in variables else aliases.get(skey, skey)
        if skey not in variables:
            raise Ct2FormatError(f"quantised variable {key!r} ({a.dtype}) has no {key + '_scale'!r} companion")
        scale = variables[skey]
        scale = (_bf16_to_f32(scale) if skey in bf16 else scale.astype(np.float32))
        if scale.ndim == 0 or scale.size == 1:
            return a.astype(np.float32) / np.float32(scale.reshape(-1)[0])
        if scale.shape[0] != a.shape[0]:
            raise Ct2FormatError(f"{skey!r}: {scale.shape[0]} scales for {a.shape[0]} rows of {key!r}")
        return a.astype(np.float32) / scale.reshape((-1,) + (1,) * (a.ndim - 1))
    return a.astype(np.float32)


def _count_layers(names: Iterable[str], side: str) -> int:
    idx = set()
    prefix = side + "/layer_"
    for n in names:
        if n.startswith(prefix) and n[len(prefix):].split("/", 1)[0].isdigit():      # not encoder/layer_norm/...
            idx.add(int(n[len(prefix):].split("/", 1)[0]))
    if idx and idx != set(range(len(idx))):
        raise Ct2FormatError(f"{side}: layer indices {sorted(idx)} are not 0..{len(idx) - 1}")
    return len(idx)


def load_ct2_whisper(path: str) -> Tuple[WhisperDims, Dict[str, np.ndarray], Dict[str, object]]:
    """A CTranslate2 Whisper directory -> ``(dims, openai-named float32 state dict, extras)``.  extras: ``alignment_heads``,
    ``suppress_ids``, ``suppress_ids_begin``, ``lang_ids`` (config.json, as faster-whisper reads them), ``stored_types``,
    ``vocabulary_size`` when ``vocabulary.json`` / ``vocabulary.txt`` is present (checked against the embedding)."""
    model_bin = os.path.join(path, "model.bin")
    variables, aliases, header = read_model_bin(model_bin)
    if header["spec"] != "WhisperSpec":
        raise Ct2FormatError(f"{model_bin}: spec {header['spec']!r}; a WhisperSpec conversion is needed (ct2-transformers-converter "
                             "of a WhisperForConditionalGeneration checkpoint)")
    bf16 = header["bfloat16"]
    names = set(variables) | set(aliases)

    def get(name: str) -> np.ndarray:
        try:
            return _as_f32(variables, aliases, bf16, name)
        except KeyError:
            raise Ct2FormatError(f"{model_bin}: variable {name!r} missing (have e.g. {sorted(names)[:4]} ...)") from None

    def scalar(name: str, default: Optional[int] = None) -> int:
        key = name if name in variables else aliases.get(name)
        if key is None:
            if default is None:
                raise Ct2FormatError(f"{model_bin}: scalar {name!r} missing")
            return default
        return int(np.asarray(variables[key]).reshape(-1)[0])

    conv1 = get("encoder/conv1/weight")
    n_audio_layer, n_text_layer = _count_layers(names, "encoder"), _count_layers(names, "decoder")
    emb = get("decoder/embeddings/weight")
    d_model = int(conv1.shape[0])
    dims = WhisperDims(n_mels=int(conv1.shape[1]), n_audio_ctx=int(get("encoder/position_encodings/encodings").shape[0]),
                       n_audio_state=d_model, n_audio_head=scalar("encoder/num_heads", max(1, d_model // 64)),
                       n_audio_layer=n_audio_layer, n_vocab=int(emb.shape[0]),
                       n_text_ctx=int(get("decoder/position_encodings/encodings").shape[0]), n_text_state=int(emb.shape[1]),
                       n_text_head=scalar("decoder/num_heads", max(1, int(emb.shape[1]) // 64)), n_text_layer=n_text_layer)
    if n_audio_layer == 0 or n_text_layer == 0:
        raise Ct2FormatError(f"{model_bin}: no encoder/layer_i or decoder/layer_i variables")
    sd: Dict[str, np.ndarray] = {}
    sd["encoder.conv1.weight"], sd["encoder.conv1.bias"] = conv1, get("encoder/conv1/bias")
    sd["encoder.conv2.weight"], sd["encoder.conv2.bias"] = get("encoder/conv2/weight"), get("encoder/conv2/bias")
    sd["encoder.positional_embedding"] = get("encoder/position_encodings/encodings")
    sd["encoder.ln_post.weight"], sd["encoder.ln_post.bias"] = get("encoder/layer_norm/gamma"), get("encoder/layer_norm/beta")
    sd["decoder.token_embedding.weight"] = emb
    sd["decoder.positional_embedding"] = get("decoder/position_encodings/encodings")
    sd["decoder.ln.weight"], sd["decoder.ln.bias"] = get("decoder/layer_norm/gamma"), get("decoder/layer_norm/beta")
    if "decoder/projection/weight" in names:       # Whisper ties the output projection to the embedding; a model that does not is not one
        proj = get("decoder/projection/weight")
        if proj.shape != emb.shape or not np.array_equal(proj, emb):
            raise Ct2FormatError(f"{model_bin}: decoder/projection/weight differs from the token embedding (untied output layer)")

    def ln(dst: str, src: str) -> None:
        sd[dst + ".weight"], sd[dst + ".bias"] = get(src + "/gamma"), get(src + "/beta")

    def linear(dst: str, src: str) -> None:
        sd[dst + ".weight"], sd[dst + ".bias"] = get(src + "/weight"), get(src + "/bias")

    def fused(dst: str, src: str, parts: Sequence[str], d: int) -> None:
        w, b = get(src + "/weight"), get(src + "/bias")
        if w.shape[0] != d * len(parts):
            raise Ct2FormatError(f"{model_bin}: {src}/weight has {w.shape[0]} rows, {len(parts)} x {d} expected")
        for i, p in enumerate(parts):
            sd[f"{dst}.{p}.weight"] = np.ascontiguousarray(w[i * d: (i + 1) * d])
            if p == "key":          # no bias in Whisper: the converter wrote zeros
                if np.any(b[i * d: (i + 1) * d] != 0):
                    raise Ct2FormatError(f"{model_bin}: {src}/bias carries a non-zero key bias; Whisper has none")
            else:
                sd[f"{dst}.{p}.bias"] = np.ascontiguousarray(b[i * d: (i + 1) * d])

    for i in range(n_audio_layer):
        s, p = f"encoder/layer_{i}", f"encoder.blocks.{i}"
        ln(p + ".attn_ln", s + "/self_attention/layer_norm")
        fused(p + ".attn", s + "/self_attention/linear_0", ("query", "key", "value"), dims.n_audio_state)
        linear(p + ".attn.out", s + "/self_attention/linear_1")
        ln(p + ".mlp_ln", s + "/ffn/layer_norm")
        linear(p + ".mlp.0", s + "/ffn/linear_0")
        linear(p + ".mlp.2", s + "/ffn/linear_1")
    for i in range(n_text_layer):
        s, p = f"decoder/layer_{i}", f"decoder.blocks.{i}"
        ln(p + ".attn_ln", s + "/self_attention/layer_norm")
        fused(p + ".attn", s + "/self_attention/linear_0", ("query", "key", "value"), dims.n_text_state)
        linear(p + ".attn.out", s + "/self_attention/linear_1")
        ln(p + ".cross_attn_ln", s + "/attention/layer_norm")
        linear(p + ".cross_attn.query", s + "/attention/linear_0")
        fused(p + ".cross_attn", s + "/attention/linear_1", ("key", "value"), dims.n_text_state)
        linear(p + ".cross_attn.out", s + "/attention/linear_2")
        ln(p + ".mlp_ln", s + "/ffn/layer_norm")
        linear(p + ".mlp.0", s + "/ffn/linear_0")
        linear(p + ".mlp.2", s + "/ffn/linear_1")
    extras: Dict[str, object] = {"stored_types": sorted({("bfloat16" if k in bf16 else str(v.dtype)) for k, v in variables.items() if v.ndim >= 2}),
                                 "spec_revision": header["revision"], "binary_version": header["binary_version"]}
    cfg_path = os.path.join(path, "config.json")
    if os.path.exists(cfg_path):
        with open(cfg_path) as f:
            cfg = json.load(f)
        if cfg.get("alignment_heads"):
            extras["alignment_heads"] = [(int(a), int(b)) for a, b in cfg["alignment_heads"]]
        for k in ("suppress_ids", "suppress_ids_begin", "lang_ids"):
            if k in cfg:
                extras[k] = [int(x) for x in cfg[k]]
    for vocab_name in ("vocabulary.json", "vocabulary.txt"):
        vp = os.path.join(path, vocab_name)
        if os.path.exists(vp):
            with open(vp, encoding="utf-8") as f:
                n = len(json.load(f)) if vocab_name.endswith(".json") else sum(1 for _ in f)
            extras["vocabulary_size"] = n
            if n != dims.n_vocab:
                raise Ct2FormatError(f"{vp}: {n} entries, the embedding has {dims.n_vocab} rows")
            break
    return dims, sd, extras


def whisper_to_ct2_variables(dims: WhisperDims, sd: Dict[str, np.ndarray], dtype: str = "float16", quantization: Optional[str] = None
                             ) -> Tuple[Dict[str, np.ndarray], Dict[str, str], List[str]]:
    """The inverse mapping (openai-named state dict -> the converter's variable tree): what ``ct2-transformers-converter
    --quantization float16|bfloat16|float32|int8|int8_float16`` writes.  Returns ``(variables, aliases, bfloat16 names)`` for
    ``write_model_bin``.  Used to export a model for the reference's own stack and by the round-trip tests."""
    if dtype not in ("float32", "float16", "bfloat16"):
        raise ValueError("dtype must be float32, float16 or bfloat16")
    if quantization not in (None, "int8"):
        raise ValueError("quantization must be None or 'int8'")
    out: Dict[str, np.ndarray] = {}
    bf: List[str] = []

    def put(name: str, a: np.ndarray, quantise: bool = False) -> None:
        a = np.ascontiguousarray(a, dtype=np.float32)
        if quantise and quantization == "int8" and a.ndim == 2:
            amax = np.abs(a).max(axis=1)
            amax[amax == 0] = 127.0
            scale = (127.0 / amax).astype(np.float32)
            out[name] = np.rint(a * scale[:, None]).astype(np.int8)
            put(name + "_scale", scale)
            return
        if dtype == "bfloat16":
            out[name] = _f32_to_bf16(a)
            bf.append(name)
        else:
            out[name] = a.astype(dtype)

    def ln(dst: str, src: str) -> None:
        put(dst + "/gamma", sd[src + ".weight"])
        put(dst + "/beta", sd[src + ".bias"])

    def linear(dst: str, w: np.ndarray, b: np.ndarray) -> None:
        put(dst + "/weight", w, quantise=True)
        put(dst + "/bias", b)

    def fuse(dst: str, src: str, parts: Sequence[str]) -> None:
        ws = [sd[f"{src}.{p}.weight"] for p in parts]
        bs = [sd.get(f"{src}.{p}.bias", np.zeros(ws[i].shape[0], np.float32)) for i, p in enumerate(parts)]
        linear(dst, np.concatenate(ws, axis=0), np.concatenate(bs, axis=0))

    put("encoder/conv1/weight", sd["encoder.conv1.weight"])
    put("encoder/conv1/bias", sd["encoder.conv1.bias"])
    put("encoder/conv2/weight", sd["encoder.conv2.weight"])
    put("encoder/conv2/bias", sd["encoder.conv2.bias"])
    put("encoder/position_encodings/encodings", sd["encoder.positional_embedding"])
    ln("encoder/layer_norm", "encoder.ln_post")
    out["encoder/num_heads"] = np.int16(dims.n_audio_head)
    for i in range(dims.n_audio_layer):
        s, p = f"encoder.blocks.{i}", f"encoder/layer_{i}"
        ln(p + "/self_attention/layer_norm", s + ".attn_ln")
        fuse(p + "/self_attention/linear_0", s + ".attn", ("query", "key", "value"))
        linear(p + "/self_attention/linear_1", sd[s + ".attn.out.weight"], sd[s + ".attn.out.bias"])
        ln(p + "/ffn/layer_norm", s + ".mlp_ln")
        linear(p + "/ffn/linear_0", sd[s + ".mlp.0.weight"], sd[s + ".mlp.0.bias"])
        linear(p + "/ffn/linear_1", sd[s + ".mlp.2.weight"], sd[s + ".mlp.2.bias"])
    put("decoder/embeddings/weight", sd["decoder.token_embedding.weight"], quantise=True)
    put("decoder/position_encodings/encodings", sd["decoder.positional_embedding"])
    ln("decoder/layer_norm", "decoder.ln")
    out["decoder/num_heads"] = np.int16(dims.n_text_head)
    out["decoder/scale_embeddings"] = np.int8(0)
    out["decoder/start_from_zero_embedding"] = np.int8(0)
    for i in range(dims.n_text_layer):
        s, p = f"decoder.blocks.{i}", f"decoder/layer_{i}"
        ln(p + "/self_attention/layer_norm", s + ".attn_ln")
        fuse(p + "/self_attention/linear_0", s + ".attn", ("query", "key", "value"))
        linear(p + "/self_attention/linear_1", sd[s + ".attn.out.weight"], sd[s + ".attn.out.bias"])
        ln(p + "/attention/layer_norm", s + ".cross_attn_ln")
        linear(p + "/attention/linear_0", sd[s + ".cross_attn.query.weight"], sd[s + ".cross_attn.query.bias"])
        fuse(p + "/attention/linear_1", s + ".cross_attn", ("key", "value"))
        linear(p + "/attention/linear_2", sd[s + ".cross_attn.out.weight"], sd[s + ".cross_attn.out.bias"])
        ln(p + "/ffn/layer_norm", s + ".mlp_ln")
        linear(p + "/ffn/linear_0", sd[s + ".mlp.0.weight"], sd[s + ".mlp.0.bias"])
        linear(p + "/ffn/linear_1", sd[s + ".mlp.2.weight"], sd[s + ".mlp.2.bias"])
    # the tied output layer: the converter stores one array and an alias (ModelSpec._alias_variables deduplicates equal values)
    aliases = {"decoder/projection/weight": "decoder/embeddings/weight"}
    if "decoder/embeddings/weight_scale" in out:
        aliases["decoder/projection/weight_scale"] = "decoder/embeddings/weight_scale"
    return out, aliases, bf


def write_ct2_whisper(path: str, dims: WhisperDims, sd: Dict[str, np.ndarray], dtype: str = "float16", quantization: Optional[str] = None,
                      alignment_heads: Optional[Sequence[Tuple[int, int]]] = None, vocabulary: Optional[Sequence[str]] = None,
                      suppress_ids: Sequence[int] = (), suppress_ids_begin: Sequence[int] = ()) -> None:
    """Write a CTranslate2-format Whisper directory (``model.bin``, ``config.json``, ``vocabulary.json``)."""
    os.makedirs(path, exist_ok=True)
    variables, aliases, bf = whisper_to_ct2_variables(dims, sd, dtype, quantization)
    write_model_bin(os.path.join(path, "model.bin"), variables, aliases, bfloat16=bf)
    cfg = {"alignment_heads": [list(map(int, h)) for h in (alignment_heads or [])], "lang_ids": [], "suppress_ids": list(map(int, suppress_ids)),
           "suppress_ids_begin": list(map(int, suppress_ids_begin))}
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump(cfg, f)
    vocab = list(vocabulary) if vocabulary is not None else [f"<{i}>" for i in range(dims.n_vocab)]
    with open(os.path.join(path, "vocabulary.json"), "w", encoding="utf-8") as f:
        json.dump(vocab, f, ensure_ascii=False)


# faster_whisper/utils.py _MODELS: the repositories `WhisperModel("large-v3")` resolves to
FASTER_WHISPER_REPOS = {
    "tiny.en": "Systran/faster-whisper-tiny.en", "tiny": "Systran/faster-whisper-tiny", "base.en": "Systran/faster-whisper-base.en",
    "base": "Systran/faster-whisper-base", "small.en": "Systran/faster-whisper-small.en", "small": "Systran/faster-whisper-small",
    "medium.en": "Systran/faster-whisper-medium.en", "medium": "Systran/faster-whisper-medium", "large-v1": "Systran/faster-whisper-large-v1",
    "large-v2": "Systran/faster-whisper-large-v2", "large-v3": "Systran/faster-whisper-large-v3", "large": "Systran/faster-whisper-large-v3",
    "distil-large-v2": "Systran/faster-distil-whisper-large-v2", "distil-medium.en": "Systran/faster-distil-whisper-medium.en",
    "distil-small.en": "Systran/faster-distil-whisper-small.en", "distil-large-v3": "Systran/faster-distil-whisper-large-v3",
    "large-v3-turbo": "mobiuslabsgmbh/faster-whisper-large-v3-turbo", "turbo": "mobiuslabsgmbh/faster-whisper-large-v3-turbo",
}


def resolve_cached_model(size_or_repo: str) -> Optional[str]:
    """What ``faster_whisper.utils.download_model(size, local_files_only=True)`` returns: the snapshot directory of the model's
    repository in the local Hugging Face cache, or None when it is not cached (nothing is ever downloaded here)."""
    repo = FASTER_WHISPER_REPOS.get(size_or_repo, size_or_repo if "/" in size_or_repo else None)
    if repo is None:
        return None
    try:
        import huggingface_hub
        return huggingface_hub.snapshot_download(repo, local_files_only=True,
                                                 allow_patterns=["config.json", "preprocessor_config.json", "model.bin", "tokenizer.json",
                                                                 "vocabulary.*"])
    except Exception:
        return None
