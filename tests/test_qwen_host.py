"""Host side of the Qwen3 slice (no GPU): blob packing matches the header's tensor order, the TextGenerator adapter has the
protocol's surface and refuses to run without its plug-ins."""
import re
import os

import numpy as np
import pytest

from whisperjav_amd import hipbind, qwen

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_blob_layout_matches_the_header():
    text = open(os.path.join(ROOT, "include", "wjhip.h")).read()
    layer = re.search(r"enum \{ WJ_QL_LN1_W = 0,([^}]*)WJ_QL_N \}", text).group(1)
    names = ["LN1_W"] + [n.strip().replace("WJ_QL_", "") for n in layer.split(",") if n.strip()]
    assert tuple(names) == qwen.LAYER_TENSORS
    d = qwen.Qwen3Dims(hidden=64, n_layer=2, n_head=2, n_kv_head=1, head_dim=128, ffn=96, vocab=50)
    w = qwen.synth_weights(d)
    tensors = qwen.engine_tensors(d, w)
    assert len(tensors) == 2 + d.n_layer * len(qwen.LAYER_TENSORS)
    assert tensors[2 + 1][1].shape == ((d.n_head + 2 * d.n_kv_head) * d.head_dim, d.hidden)      # fused q | k | v rows
    assert tensors[2 + 6][1].shape == (2 * d.ffn, d.hidden)                                        # gate rows, then up rows
    blob, offsets = qwen.pack_blob(d, w, "float16")
    assert len(offsets) == len(tensors) and all(o % 256 == 0 for o in offsets)
    import ctypes
    import torch
    emb = blob[int(offsets[0]): int(offsets[0]) + d.vocab * d.hidden * 2].view(torch.float16).float().numpy()
    assert np.allclose(emb.reshape(d.vocab, d.hidden), w["model.language_model.embed_tokens.weight"], atol=2e-3)
    assert ctypes.sizeof(qwen.Qwen3DimsC) == 7 * 4 + 2 * 4          # wj_qwen_dims


def test_text_generator_surface_and_refusal(tmp_path):
    d = qwen.Qwen3Dims(hidden=64, n_layer=1, n_head=1, n_kv_head=1, head_dim=128, ffn=64, vocab=32)
    gen = qwen.HipQwenTextGenerator(d, qwen.synth_weights(d))
    for name in ("generate", "generate_batch", "load", "unload", "cleanup"):      # subtitle_pipeline/protocols.py:60-110
        assert callable(getattr(gen, name))
    with pytest.raises(hipbind.WjError, match="audio_embedder"):
        gen.generate(tmp_path / "x.wav")
    gen.cleanup(); gen.cleanup()


def _tiny_transformers_checkpoint(tmp_path, dtype=None, shard=False):
    """A qwen3_asr model of transformers' own port with the engine's structural constants (head_dim 128, 128 mel bins), saved
    the way ``save_pretrained`` writes the family."""
    import torch
    tf = pytest.importorskip("transformers")
    try:
        from transformers import Qwen3ASRConfig, Qwen3ASRForConditionalGeneration
    except Exception:
        pytest.skip("transformers has no qwen3_asr")
    audio = dict(num_mel_bins=128, encoder_layers=2, encoder_attention_heads=2, encoder_ffn_dim=96, d_model=64, n_window=50,
                 output_dim=256, n_window_infer=400, downsample_hidden_size=8, max_position_embeddings=13)
    text = dict(model_type="qwen3", hidden_size=256, intermediate_size=320, num_hidden_layers=2, num_attention_heads=2,
                num_key_value_heads=1, head_dim=128, vocab_size=300, max_position_embeddings=4096, tie_word_embeddings=True,
                rms_norm_eps=1e-6, rope_parameters={"rope_type": "default", "rope_theta": 50000.0})
    cfg = Qwen3ASRConfig(audio_config=audio, text_config=text, audio_token_id=7, pad_token_id=0, eos_token_id=[1, 2])
    torch.manual_seed(0)
    model = Qwen3ASRForConditionalGeneration(cfg).eval()
    with torch.no_grad():
        for name, p in model.named_parameters():
            if p.dim() >= 2:
                p.mul_(3.0)
    if dtype is not None:
        model = model.to(dtype)
    model.save_pretrained(str(tmp_path), safe_serialization=True, **({"max_shard_size": "200KB"} if shard else {}))
    return model


@pytest.mark.parametrize("flavour", ["single", "sharded", "bfloat16", "thinker"])
def test_checkpoint_directory_loads_into_the_engine_layout(tmp_path, flavour):
    """``qwen.load_checkpoint`` (VERDICT r3 missing #3: a loader for the published safetensors layout, not only name packing):
    a directory written by ``save_pretrained`` of transformers' qwen3_asr port -- one file, sharded with an index, stored in
    bfloat16, or with the omni-style ``thinker`` nesting of config and tensor names -- gives the geometry of its config.json
    and every tensor of its state dict; the engine's packers accept the result; and the oracle built from it reproduces the
    transformers model's logits, so the names mean what the loader thinks they mean."""
    import json
    import torch
    from oracle import qwen3_ref
    model = _tiny_transformers_checkpoint(tmp_path, dtype=torch.bfloat16 if flavour == "bfloat16" else None, shard=flavour == "sharded")
    if flavour == "sharded":
        assert (tmp_path / "model.safetensors.index.json").exists()
    if flavour == "thinker":
        from safetensors.torch import load_file, save_file
        cfg = json.loads((tmp_path / "config.json").read_text())
        inner = {k: cfg.pop(k) for k in ("audio_config", "text_config", "audio_token_id", "eos_token_id")}
        cfg["thinker_config"] = inner
        (tmp_path / "config.json").write_text(json.dumps(cfg))
        sd = load_file(str(tmp_path / "model.safetensors"))
        save_file({"thinker." + k: v for k, v in sd.items()}, str(tmp_path / "model.safetensors"))
    d, ad, w = qwen.load_checkpoint(tmp_path)
    assert d == qwen.Qwen3Dims(hidden=256, n_layer=2, n_head=2, n_kv_head=1, head_dim=128, ffn=320, vocab=300, rope_theta=50000.0,
                               rms_eps=1e-6, audio_token_id=7, eos_token_ids=(1, 2))
    assert ad == qwen.Qwen3AudioDims(n_mels=128, n_layer=2, n_head=2, ffn=96, d_model=64, n_window=50, n_window_infer=400,
                                     conv_hidden=8, out_dim=256)
    sd = {k: v.detach().float().numpy() for k, v in model.state_dict().items() if k != "lm_head.weight"}
    assert set(sd) <= set(w) | {"lm_head.weight"} and all(v.dtype == np.float32 for v in w.values())
    for k, v in sd.items():
        assert np.array_equal(w[k], v), k
    blob, offsets = qwen.pack_blob(d, w, "float16")
    ablob, aoffsets = qwen.pack_audio_blob(ad, w, "float16")
    gen = qwen.HipQwenTextGenerator.from_pretrained(tmp_path, batch_size=2)          # no device work before load()
    assert gen.dims == d and gen.audio_dims == ad and gen.batch_size == 2
    with pytest.raises(KeyError, match="forced-aligner"):
        qwen.HipQwenForcedAligner.from_pretrained(tmp_path)
    assert len(offsets) == 2 + d.n_layer * len(qwen.LAYER_TENSORS) and len(aoffsets) == len(qwen.AUDIO_GLOBALS) + ad.n_layer * len(qwen.AUDIO_LAYER)
    # the oracle on the loaded tensors == the transformers model on the same prompt
    od = qwen3_ref.Qwen3AsrDims(n_mels=ad.n_mels, a_layers=ad.n_layer, a_heads=ad.n_head, a_ffn=ad.ffn, a_d=ad.d_model, n_window=ad.n_window,
                                n_window_infer=ad.n_window_infer, conv_hidden=ad.conv_hidden, d=d.hidden, layers=d.n_layer, heads=d.n_head,
                                kv_heads=d.n_kv_head, head_dim=d.head_dim, ffn=d.ffn, vocab=d.vocab, rope_theta=d.rope_theta, rms_eps=d.rms_eps,
                                audio_token_id=d.audio_token_id, eos_token_ids=d.eos_token_ids)
    oracle = qwen3_ref.Qwen3AsrOracle(od, w)
    n_frames = 230
    mel = torch.randn(ad.n_mels, n_frames, generator=torch.Generator().manual_seed(5))
    n_audio = qwen3_ref.audio_token_count(n_frames, ad.n_window)
    prompt = [11, 12] + [d.audio_token_id] * n_audio + [13, 14]
    feats = torch.zeros(1, ad.n_mels, 300)
    feats[0, :, :n_frames] = mel
    mask = torch.zeros(1, 300, dtype=torch.long)
    mask[0, :n_frames] = 1
    if flavour == "bfloat16":      # reference = the same file read back by transformers in fp32 (its buffers are then fp32 again)
        from transformers import Qwen3ASRForConditionalGeneration
        model = Qwen3ASRForConditionalGeneration.from_pretrained(str(tmp_path), dtype=torch.float32).eval()
    with torch.no_grad():
        ref = model(input_ids=torch.tensor([prompt]), input_features=feats, input_features_mask=mask).logits[0]
        got = oracle.logits(oracle.embed(prompt, oracle.audio_tokens(mel)))
    assert float((got - ref).abs().max()) < 5e-4 * max(1.0, float(ref.abs().max()))


def test_checkpoint_loader_refuses_what_the_engine_does_not_implement(tmp_path):
    import json
    _tiny_transformers_checkpoint(tmp_path)
    cfg = json.loads((tmp_path / "config.json").read_text())
    good = json.dumps(cfg)
    for key, value, match in (("head_dim", 64, "head_dim"), ("attention_bias", True, "attention_bias"), ("tie_word_embeddings", False, "untied")):
        bad = json.loads(good)
        bad["text_config"][key] = value
        (tmp_path / "config.json").write_text(json.dumps(bad))
        with pytest.raises(ValueError, match=match):
            qwen.load_checkpoint(tmp_path)
    bad = json.loads(good)
    bad["text_config"]["num_hidden_layers"] = 3                 # config promises a layer the file does not hold
    (tmp_path / "config.json").write_text(json.dumps(bad))
    with pytest.raises(KeyError, match="missing"):
        qwen.load_checkpoint(tmp_path)
    bad = json.loads(good)
    bad["text_config"]["intermediate_size"] = 384
    (tmp_path / "config.json").write_text(json.dumps(bad))
    with pytest.raises(ValueError, match="shapes"):
        qwen.load_checkpoint(tmp_path)


def _tiny_processor():
    """A ``Qwen3ASRProcessor`` of transformers over a synthetic word-level vocabulary and a synthetic chat template (the
    published tokenizer and template are not available offline; the plug-ins must follow WHATEVER the checkpoint ships)."""
    pytest.importorskip("transformers")
    tokenizers = pytest.importorskip("tokenizers")
    try:
        from transformers import PreTrainedTokenizerFast
        from transformers.models.qwen3_asr import Qwen3ASRFeatureExtractor, Qwen3ASRProcessor
    except Exception:
        pytest.skip("transformers has no qwen3_asr")
    specials = ["<|endoftext|>", "<|im_start|>", "<|im_end|>", "<|audio_start|>", "<|audio_end|>", "<|audio_pad|>", "<asr_text>", "<timestamp>"]
    vocab = {t: i for i, t in enumerate(specials)}
    for wd in "system user assistant language Japanese English hello world this is a test the context words it's".split(" "):
        vocab.setdefault(wd, len(vocab))
    vocab["[UNK]"] = len(vocab)
    tok = tokenizers.Tokenizer(tokenizers.models.WordLevel(vocab, unk_token="[UNK]"))
    tok.pre_tokenizer = tokenizers.pre_tokenizers.Split(" ", "removed")
    fast = PreTrainedTokenizerFast(tokenizer_object=tok, unk_token="[UNK]", pad_token="<|endoftext|>", eos_token="<|im_end|>",
                                   additional_special_tokens=[t for t in specials if t != "<asr_text>"],
                                   extra_special_tokens={"audio_token": "<|audio_pad|>", "audio_bos_token": "<|audio_start|>",
                                                         "audio_eos_token": "<|audio_end|>"})
    fast.add_tokens(["<asr_text>"])            # an ordinary added token: it survives skip_special_tokens, the parser keys on it
    template = ("{% for m in messages %}<|im_start|> {{ m['role'] }} "
                "{% for c in m['content'] %}{% if c['type'] == 'audio' %}<|audio_start|><|audio_pad|><|audio_end|> "
                "{% else %}{{ c['text'] }} {% if m['role'] == 'user' %}<timestamp> <timestamp> {% endif %}{% endif %}{% endfor %}"
                "{% if not loop.last or not continue_final_message %}<|im_end|> {% endif %}{% endfor %}")
    ids = {t: fast.convert_tokens_to_ids(t) for t in vocab}          # the tokenizer's own ids (added tokens may be renumbered)
    return Qwen3ASRProcessor(Qwen3ASRFeatureExtractor(), fast, chat_template=template), ids


@pytest.mark.parametrize("seconds,language,context", [(3.0, "ja", "the context"), (0.7, "en", None), (5.3, None, None)])
def test_processor_plugins_build_the_prompt_the_upstream_processor_builds(seconds, language, context):
    """``qwen.processor_plugins``: the generator's prompt for a clip of n audio tokens equals, id for id, what transformers'
    ``Qwen3ASRProcessor.apply_transcription_request`` produces for a real clip of that length (forced language, auto-detect,
    with and without a context turn); ``detokenize`` is the processor's transcription-only decode."""
    from oracle import qwen3_ref
    proc, vocab = _tiny_processor()
    plug = qwen.processor_plugins(proc, timestamp_token_id=vocab["<timestamp>"])
    audio = (np.random.default_rng(0).standard_normal(int(16000 * seconds)) * 0.1).astype(np.float32)
    want = proc.apply_transcription_request(audio=[audio], language=language, prompt=context)
    n_frames = int(want["input_features_mask"].sum())
    n_audio = qwen3_ref.audio_token_count(n_frames, 50)
    assert int((want["input_ids"][0] == vocab["<|audio_pad|>"]).sum()) == n_audio
    got = plug["prompt_builder"](n_audio, language, context)
    assert got == want["input_ids"][0].tolist()
    ids = [vocab[t] for t in ("language", "English", "<asr_text>", "hello", "world", "<|im_end|>")]
    assert plug["detokenize"](ids) == "hello world" == proc.decode(ids, return_format="transcription_only")
    assert plug["detokenize"]([vocab["hello"], vocab["world"]]) == "hello world"          # forced language: only the transcript is generated


def test_processor_plugins_build_the_forced_aligner_prompt():
    """``word_prompt`` / ``split_words`` against ``prepare_forced_aligner_inputs``: same word list, same ids, and the marker
    positions are the positions of the timestamp token -- two per word."""
    from oracle import qwen3_ref
    proc, vocab = _tiny_processor()
    plug = qwen.processor_plugins(proc, timestamp_token_id=vocab["<timestamp>"])
    audio = (np.random.default_rng(1).standard_normal(16000 * 2) * 0.1).astype(np.float32)
    text = "hello, world -- this is a test; it's"
    want, word_lists = proc.prepare_forced_aligner_inputs(audio=[audio], transcript=[text], language="en")
    words = plug["split_words"](text, "en")
    assert words == word_lists[0] == ["hello", "world", "this", "is", "a", "test", "it's"]
    n_audio = qwen3_ref.audio_token_count(int(want["input_features_mask"].sum()), 50)
    ids, marks = plug["word_prompt"](n_audio, words, "en")
    assert ids == want["input_ids"][0].tolist()
    assert len(marks) == 2 * len(words) and all(ids[i] == vocab["<timestamp>"] for i in marks)
    with pytest.raises(ValueError, match="timestamp_token_id"):
        qwen.processor_plugins(proc)["word_prompt"](n_audio, words, "en")


def test_from_pretrained_builds_its_plug_ins_from_the_directory(tmp_path):
    """A directory holding model AND processor files needs no hand-written callables: ``from_pretrained`` builds them over
    ``AutoProcessor.from_pretrained`` (no device work before ``load()``)."""
    import json
    _tiny_transformers_checkpoint(tmp_path)
    proc, vocab = _tiny_processor()
    proc.save_pretrained(str(tmp_path))
    cfg = json.loads((tmp_path / "config.json").read_text())
    cfg["timestamp_token_id"] = vocab["<timestamp>"]
    (tmp_path / "config.json").write_text(json.dumps(cfg))
    gen = qwen.HipQwenTextGenerator.from_pretrained(tmp_path, batch_size=2)
    want = proc.apply_transcription_request(audio=[np.zeros(16000, np.float32)], language="ja")["input_ids"][0].tolist()
    assert gen.prompt_builder(want.count(vocab["<|audio_pad|>"]), "ja", None) == want        # 1 s = 13 audio tokens
    ids = gen.prompt_builder(5, "en", "the context")
    assert ids.count(vocab["<|audio_pad|>"]) == 5 and ids[-1] == vocab["<asr_text>"]
    mine = lambda n, lang, ctx: [1, 2, 3]          # noqa: E731  a caller's own callable wins
    assert qwen.HipQwenTextGenerator.from_pretrained(tmp_path, prompt_builder=mine, detokenize=str).prompt_builder is mine
