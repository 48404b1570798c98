"""CTranslate2 model directories (whisperjav_amd/ct2_format.py): what ``faster_whisper.WhisperModel(model_size_or_path=...)``
opens at /root/reference/whisperjav/modules/faster_whisper_pro_asr.py:246-253.  CPU only: the byte format (every field of the
published serialisation), the variable tree <-> openai names mapping in every storage type, the packer's bytes, and the error
paths.  The wheel-gated comparison with ctranslate2's own converter lives in tests/test_upstream_wheels.py."""
import json
import os
import struct

import numpy as np
import pytest

from tests import helpers
from whisperjav_amd import ct2_format as ct2, weights as pweights


def _model(seed=5):
    d = helpers.small_dims(n_mels=80, d_model=128, heads=2, layers=2, n_vocab=51865)
    return d, pweights.synth_weights(d, seed=seed, exact="float16")


def test_byte_layout_of_the_serialisation(tmp_path):
    """Hand-decoded header and first variable: u32 version 6, u16-prefixed NUL-terminated strings, u32 revision / count, then
    name, u8 rank, u32 dims, u8 type id, u32 byte count, raw little-endian data; aliases after the variables."""
    path = str(tmp_path / "model.bin")
    a = np.arange(6, dtype=np.float16).reshape(2, 3)
    ct2.write_model_bin(path, {"b/weight": a, "a/num_heads": np.int16(6)}, {"c/weight": "b/weight"}, spec="WhisperSpec", revision=3)
    raw = open(path, "rb").read()
    assert struct.unpack_from("<I", raw, 0)[0] == 6
    assert struct.unpack_from("<H", raw, 4)[0] == len("WhisperSpec") + 1 and raw[6:17] == b"WhisperSpec" and raw[17] == 0
    assert struct.unpack_from("<II", raw, 18) == (3, 2)
    pos = 26          # variables sorted by name: "a/num_heads" first -- a rank-0 int16
    assert struct.unpack_from("<H", raw, pos)[0] == 12 and raw[pos + 2: pos + 13] == b"a/num_heads"
    pos += 2 + 12
    assert raw[pos] == 0 and raw[pos + 1] == 2 and struct.unpack_from("<I", raw, pos + 2)[0] == 2       # rank 0, type id 2 = int16, 2 bytes
    assert struct.unpack_from("<h", raw, pos + 6)[0] == 6
    pos += 8
    assert raw[pos + 2: pos + 10] == b"b/weight"
    pos += 2 + 9
    assert raw[pos] == 2 and struct.unpack_from("<II", raw, pos + 1) == (2, 3) and raw[pos + 9] == 4    # rank 2, dims, type id 4 = float16
    assert struct.unpack_from("<I", raw, pos + 10)[0] == 12
    assert np.array_equal(np.frombuffer(raw, np.float16, 6, pos + 14).reshape(2, 3), a)
    v, al, hdr = ct2.read_model_bin(path)
    assert hdr == {"binary_version": 6, "spec": "WhisperSpec", "revision": 3, "bfloat16": []}
    assert al == {"c/weight": "b/weight"} and np.array_equal(v["b/weight"], a) and int(v["a/num_heads"]) == 6 and v["a/num_heads"].shape == ()


@pytest.mark.parametrize("dtype,quant,tol", [("float16", None, 0.0), ("float32", None, 0.0), ("bfloat16", None, 2.0 ** -8),
                                             ("float16", "int8", 1.0 / 127), ("float32", "int8", 1.0 / 127)])
def test_round_trip_through_a_model_directory(tmp_path, dtype, quant, tol):
    """write_ct2_whisper -> load_ct2_whisper gives the dims and (fp16-representable) weights back: exactly for the float
    storage types, to half a quantisation step per row for int8; the fused q;k;v / k;v matrices are split, the zero key bias the
    converter writes is dropped, the tied projection alias is honoured, config.json's alignment heads arrive."""
    d, w = _model()
    heads = [(1, 0), (1, 1)]
    ct2.write_ct2_whisper(str(tmp_path), d, w, dtype=dtype, quantization=quant, alignment_heads=heads, suppress_ids=[1, 2], suppress_ids_begin=[220])
    dims, sd, extras = ct2.load_ct2_whisper(str(tmp_path))
    assert dims == d
    assert extras["alignment_heads"] == heads and extras["suppress_ids"] == [1, 2] and extras["vocabulary_size"] == d.n_vocab
    assert set(sd) == set(w), sorted(set(sd) ^ set(w))[:6]
    for k in w:
        assert sd[k].shape == w[k].shape and sd[k].dtype == np.float32, k
        quantised = bool(quant) and w[k].ndim == 2          # vectors, conv kernels: never quantised
        if dtype == "bfloat16":
            stored = w[k]
        else:                                               # what the storage type itself keeps (a float16 model stores fp16 biases too)
            stored = w[k].astype(dtype).astype(np.float32)
        if not quantised and dtype != "bfloat16":
            assert np.array_equal(sd[k], stored), k
            continue
        scale = np.abs(w[k]).max(axis=-1, keepdims=True) * 0.5 if quantised else np.abs(w[k])
        slack = np.abs(w[k]) * 2.0 ** -10 if (quantised and dtype == "float16") else 0.0
        assert np.all(np.abs(sd[k] - stored) <= tol * scale * (1 + 1e-5) + slack + 1e-12), k      # an exact tie (x.5) sits ON the bound
    assert ("int8" in extras["stored_types"]) == bool(quant)


def test_packed_blob_is_the_one_the_openai_names_give(tmp_path):
    """The reader feeds the SAME packer as every other loader: the engine blob of a float16 CTranslate2 directory is byte for
    byte the blob of the weights it was written from."""
    d, w = _model(seed=9)
    w = {k: a.astype(np.float16).astype(np.float32) for k, a in w.items()}     # a float16 model stores its vectors in fp16 as well
    ct2.write_ct2_whisper(str(tmp_path), d, w, dtype="float16")
    dims, sd, _ = ct2.load_ct2_whisper(str(tmp_path))
    blob_a, off_a = pweights.pack_blob(d, w, "float16")
    blob_b, off_b = pweights.pack_blob(dims, sd, "float16")
    assert np.array_equal(off_a, off_b) and bool((blob_a == blob_b).all())


def test_errors_name_the_field(tmp_path):
    d, w = _model()
    ct2.write_ct2_whisper(str(tmp_path), d, w)
    raw = open(tmp_path / "model.bin", "rb").read()
    bad = tmp_path / "bad"
    bad.mkdir()
    (bad / "model.bin").write_bytes(struct.pack("<I", 3) + raw[4:])
    with pytest.raises(ct2.Ct2FormatError, match="binary version 3"):
        ct2.load_ct2_whisper(str(bad))
    (bad / "model.bin").write_bytes(raw[: len(raw) // 2])
    with pytest.raises(ct2.Ct2FormatError, match="truncated"):
        ct2.load_ct2_whisper(str(bad))
    v, al, _ = ct2.read_model_bin(str(tmp_path / "model.bin"))
    ct2.write_model_bin(str(bad / "model.bin"), v, al, spec="TransformerSpec")
    with pytest.raises(ct2.Ct2FormatError, match="TransformerSpec"):
        ct2.load_ct2_whisper(str(bad))
    v2 = {k: a for k, a in v.items() if k != "decoder/layer_1/attention/linear_1/weight"}
    ct2.write_model_bin(str(bad / "model.bin"), v2, al)
    with pytest.raises(ct2.Ct2FormatError, match="decoder/layer_1/attention/linear_1/weight"):
        ct2.load_ct2_whisper(str(bad))
    ct2.write_model_bin(str(bad / "model.bin"), v, al)
    (bad / "vocabulary.json").write_text(json.dumps(["a", "b"]))
    with pytest.raises(ct2.Ct2FormatError, match="2 entries"):
        ct2.load_ct2_whisper(str(bad))


def test_model_names_resolve_only_through_the_local_cache(monkeypatch, tmp_path):
    """``WhisperModel("large-v3")`` = Systran/faster-whisper-large-v3 in the Hugging Face cache; nothing is downloaded."""
    monkeypatch.setenv("HF_HOME", str(tmp_path))
    monkeypatch.setenv("HF_HUB_OFFLINE", "1")
    assert ct2.FASTER_WHISPER_REPOS["large-v3"] == "Systran/faster-whisper-large-v3"
    assert ct2.resolve_cached_model("large-v3") is None and ct2.resolve_cached_model("not-a-model") is None


def test_vocabulary_file_decodes_text_when_the_directory_has_no_tokenizer_json(tmp_path):
    """A CTranslate2 directory without ``tokenizer.json`` (faster-whisper would fetch one from the hub): ``VocabTokenizer`` decodes
    the byte-level BPE token strings of ``vocabulary.json`` -- pinned against the ``tokenizers`` library's own ByteLevel decoder
    and alphabet; multi-byte characters split across tokens come back whole; encoding text is refused with the reason."""
    tokenizers = pytest.importorskip("tokenizers")
    from whisperjav_amd import whisper_model as wm
    table = wm._bytes_to_unicode()
    assert sorted(table.values()) == sorted(tokenizers.pre_tokenizers.ByteLevel.alphabet()) and len(set(table.values())) == 256
    rng = np.random.default_rng(1)
    text = " こんにちは、世界。 Thank you — naïve café ✓ あっ…"
    raw = text.encode("utf-8")
    cuts = sorted(set(rng.integers(1, len(raw), 14).tolist()))
    pieces = [raw[a:b] for a, b in zip([0] + cuts, cuts + [len(raw)])]
    vocab = ["<|endoftext|>"] + ["".join(table[b] for b in p) for p in pieces]
    (tmp_path / "vocabulary.json").write_text(json.dumps(vocab, ensure_ascii=False), encoding="utf-8")
    tok = wm.VocabTokenizer.from_directory(str(tmp_path))
    ids = list(range(1, len(vocab)))
    assert tok.decode(ids) == text == tokenizers.decoders.ByteLevel().decode(vocab[1:])
    assert tok.decode(ids + [0]).endswith("<|endoftext|>") and tok.token_to_id("<|endoftext|>") == 0
    words, word_tokens = tok.split_tokens_on_unicode(ids)
    assert "".join(words) == text and sum(len(t) for t in word_tokens) == len(ids) and all("�" not in w_ for w_ in words)
    with pytest.raises(ValueError, match="merge table"):
        tok.encode("こんにちは")


def test_vocabulary_tokenizer_non_speech_list(tmp_path):
    """``suppress_tokens=[-1]`` without a tokenizer.json: config.json's ``suppress_ids`` when present, else the single-entry symbols."""
    from whisperjav_amd import whisper_model as wm
    table = wm._bytes_to_unicode()
    enc = lambda t: "".join(table[b] for b in t.encode("utf-8"))      # noqa: E731
    vocab = ["<|endoftext|>", enc(" -"), enc("("), enc(" ("), enc("♪♪"), enc(" hello"), enc("「")]
    assert wm.VocabTokenizer(vocab).non_speech_tokens() == [1, 2, 3, 4, 6]
    assert wm.VocabTokenizer(vocab, non_speech=[9, 3, 3]).non_speech_tokens() == [3, 9]
