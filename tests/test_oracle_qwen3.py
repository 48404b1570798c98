"""The Qwen3-ASR oracle (oracle/qwen3_ref.py) pinned against transformers' independent implementation
(``transformers.models.qwen3_asr``) on seeded random weights: mel -> audio tower -> projector -> Qwen3 decoder -> logits,
incremental (KV-cached) decoding and greedy generation."""
import numpy as np
import pytest
import torch

from oracle import qwen3_ref

tf = pytest.importorskip("transformers")
try:
    from transformers import Qwen3ASRConfig, Qwen3ASRForConditionalGeneration
except Exception:       # older transformers: the model family is absent
    pytest.skip("transformers has no qwen3_asr", allow_module_level=True)


def tiny(dims: qwen3_ref.Qwen3AsrDims, seed=0):
    audio = dict(num_mel_bins=dims.n_mels, encoder_layers=dims.a_layers, encoder_attention_heads=dims.a_heads,
                 encoder_ffn_dim=dims.a_ffn, d_model=dims.a_d, n_window=dims.n_window, output_dim=dims.d,
                 n_window_infer=dims.n_window_infer, downsample_hidden_size=dims.conv_hidden, max_position_embeddings=dims.a_max_pos)
    text = dict(model_type="qwen3", hidden_size=dims.d, intermediate_size=dims.ffn, num_hidden_layers=dims.layers,
                num_attention_heads=dims.heads, num_key_value_heads=dims.kv_heads, head_dim=dims.head_dim, vocab_size=dims.vocab,
                max_position_embeddings=4096, tie_word_embeddings=True, rms_norm_eps=dims.rms_eps,
                rope_parameters={"rope_type": "default", "rope_theta": dims.rope_theta})
    cfg = Qwen3ASRConfig(audio_config=audio, text_config=text, audio_token_id=dims.audio_token_id, pad_token_id=0,
                         eos_token_id=list(dims.eos_token_ids))
    torch.manual_seed(seed)
    model = Qwen3ASRForConditionalGeneration(cfg).eval()
    with torch.no_grad():       # the default init is tiny (std 0.02): spread the weights so the logits discriminate
        for name, p in model.named_parameters():
            if p.dim() >= 2:
                p.mul_(3.0)
            elif "norm" in name and name.endswith("weight"):
                p.add_(0.1 * torch.randn_like(p))
    sd = {k: v.detach().float().numpy() for k, v in model.state_dict().items()}
    return model, sd


DIMS = qwen3_ref.Qwen3AsrDims(n_mels=32, a_layers=2, a_heads=2, a_ffn=96, a_d=48, n_window=50, n_window_infer=400, conv_hidden=8,
                              d=64, layers=3, heads=4, kv_heads=2, head_dim=32, ffn=160, vocab=300, rope_theta=10000.0,
                              audio_token_id=7, eos_token_ids=(1, 2))


@pytest.mark.parametrize("n_frames", [30, 60, 100, 250, 730])      # 30 / 60: a sub-second clip alone (its tail chunk is 100 frames wide here AND in transformers)
def test_audio_tower_and_projector_match_transformers(n_frames):
    model, sd = tiny(DIMS)
    oracle = qwen3_ref.Qwen3AsrOracle(DIMS, sd)
    g = torch.Generator().manual_seed(n_frames)
    mel = torch.randn(DIMS.n_mels, n_frames, generator=g)
    padded = (n_frames + 99) // 100 * 100
    feats = torch.zeros(1, DIMS.n_mels, padded)
    feats[0, :, :n_frames] = mel
    mask = torch.zeros(1, padded, dtype=torch.long)
    mask[0, :n_frames] = 1
    with torch.no_grad():
        ref = model.model.get_audio_features(feats, mask, return_dict=True).pooler_output
        got = oracle.audio_tokens(mel)
    assert got.shape == ref.shape and got.shape[0] == qwen3_ref.audio_token_count(n_frames, DIMS.n_window)
    assert float((got - ref).abs().max()) < 2e-4 * max(1.0, float(ref.abs().max()))


def test_decoder_logits_cache_and_greedy_match_transformers():
    model, sd = tiny(DIMS, seed=3)
    oracle = qwen3_ref.Qwen3AsrOracle(DIMS, sd)
    n_frames = 430
    mel = torch.randn(DIMS.n_mels, n_frames, generator=torch.Generator().manual_seed(5))
    n_audio = qwen3_ref.audio_token_count(n_frames, DIMS.n_window)
    prompt = [11, 12] + [DIMS.audio_token_id] * n_audio + [13, 14, 15]
    padded = (n_frames + 99) // 100 * 100
    feats = torch.zeros(1, DIMS.n_mels, padded)
    feats[0, :, :n_frames] = mel
    mask = torch.zeros(1, padded, dtype=torch.long)
    mask[0, :n_frames] = 1
    ids = torch.tensor([prompt])
    with torch.no_grad():
        ref = model(input_ids=ids, input_features=feats, input_features_mask=mask).logits[0]
        audio = oracle.audio_tokens(mel)
        x = oracle.embed(prompt, audio)
        full = oracle.logits(x)
        cache = [None] * DIMS.layers
        first = oracle.logits(x[:-2], 0, cache)
        rest = oracle.logits(x[-2:], len(prompt) - 2, cache)          # incremental == full
    assert float((full - ref).abs().max()) < 5e-4
    assert float((torch.cat([first, rest]) - full).abs().max()) < 1e-4
    with torch.no_grad():
        gen = model.generate(input_ids=ids, input_features=feats, input_features_mask=mask, max_new_tokens=12, do_sample=False)
        toks, lps = oracle.greedy(prompt, audio, 12)
    ref_new = gen[0, len(prompt):].tolist()
    ref_new = ref_new[: next((i for i, t in enumerate(ref_new) if t in DIMS.eos_token_ids), len(ref_new))]
    assert toks == ref_new and len(lps) >= len(toks)
    # the reference's pipeline default: repetition_penalty 1.1 (here stronger, so that it changes the tokens), prompt ids included
    with torch.no_grad():
        gen_p = model.generate(input_ids=ids, input_features=feats, input_features_mask=mask, max_new_tokens=12, do_sample=False,
                               repetition_penalty=1.6)
        toks_p, _ = oracle.greedy(prompt, audio, 12, repetition_penalty=1.6)
    ref_p = gen_p[0, len(prompt):].tolist()
    ref_p = ref_p[: next((i for i, t in enumerate(ref_p) if t in DIMS.eos_token_ids), len(ref_p))]
    assert toks_p == ref_p
    assert toks_p != toks            # the penalty took effect on this seed


def test_repetition_penalty_matches_transformers_processor():
    from transformers import RepetitionPenaltyLogitsProcessor
    g = torch.Generator().manual_seed(9)
    for penalty in (1.1, 1.5, 0.8):
        logits = torch.randn(50, generator=g) * 3
        ids = torch.randint(0, 50, (17,), generator=g)          # with repeats
        ref = RepetitionPenaltyLogitsProcessor(penalty)(ids[None], logits[None].clone())[0]
        got = qwen3_ref.repetition_penalised(logits, ids.tolist(), penalty)
        assert torch.equal(got, ref)
    assert qwen3_ref.repetition_penalised(logits, [], 1.3) is logits


def test_dynamic_token_limit_follows_the_reference_formula():
    f = qwen3_ref.dynamic_token_limit
    assert f(10.0, 4096, 20.0) == 256            # floor
    assert f(30.0, 4096, 20.0) == 600
    assert f(1000.0, 4096, 20.0) == 4096         # static ceiling
    assert f(30.0, 4096, 0.0) == 4096 and f(0.0, 4096, 20.0) == 4096     # disabled
    assert f(30.0, 100, 20.0) == 100             # the floor never lifts the budget over the static limit


def test_feature_extractor_is_whispers_formula_without_padding():
    """``Qwen3ASRFeatureExtractor`` == ``oracle.logmel.logmel_ow(audio, 128, padding=0)`` on the clip zero-padded to its
    8000-sample minimum (what the HIP extractor's RAW mode computes; csrc/logmel.hip)."""
    from transformers.models.qwen3_asr.feature_extraction_qwen3_asr import Qwen3ASRFeatureExtractor
    from oracle import logmel
    from whisperjav_amd import synth
    fe = Qwen3ASRFeatureExtractor()
    for seconds, seed in ((0.3, 1), (2.37, 2), (7.0, 3)):
        audio = synth.speech_like(seconds, seed=seed)
        out = fe(audio, sampling_rate=16000, padding=True, return_attention_mask=True)
        n = int(out["attention_mask"][0].sum()) if "attention_mask" in out else None
        feats = out["input_features"][0].numpy()
        padded = np.pad(audio, (0, max(0, 8000 - len(audio))))
        ref = logmel.logmel_ow(padded, 128, padding=0)
        if n is None:
            n = ref.shape[1]
        assert n == ref.shape[1] == len(padded) // 160, (n, ref.shape)
        assert np.abs(feats[:, :n] - ref).max() < 5e-5
        assert feats.shape[1] % 100 == 0 and np.abs(feats[:, n:]).max() == 0.0 if feats.shape[1] > n else True


def test_fix_timestamps_matches_transformers():
    """The forced aligner's monotonic repair: oracle and product restatements == transformers' ``_fix_timestamps`` on random
    sequences with planted outliers (short blocks, long blocks, outliers at both ends)."""
    from transformers.models.qwen3_asr.processing_qwen3_asr import _fix_timestamps
    from whisperjav_amd import qwen
    rng = np.random.default_rng(0)
    for trial in range(60):
        n = int(rng.integers(1, 40))
        base = np.sort(rng.integers(0, 6000, n)).astype(np.float64) * 80.0
        k = int(rng.integers(0, max(1, n // 2) + 1))
        idx = rng.choice(n, size=min(k, n), replace=False)
        base[idx] = rng.integers(0, 6000, len(idx)) * 80.0
        ref = _fix_timestamps(base.copy())
        assert qwen3_ref.fix_timestamps(base) == ref, (trial, base.tolist())
        assert qwen.fix_timestamps(base) == ref


def test_token_classification_head_matches_transformers():
    """The forced aligner's forward (``Qwen3ASRForTokenClassification``: the model + a linear ``score`` head) == the
    oracle's ``classify`` on the same weights."""
    from transformers import Qwen3ASRForTokenClassification
    model, sd = tiny(DIMS, seed=4)
    cfg = model.config
    cfg.num_labels = 50
    torch.manual_seed(1)
    clf = Qwen3ASRForTokenClassification(cfg).eval()
    clf.model.load_state_dict(model.model.state_dict())
    sd2 = {k: v.detach().float().numpy() for k, v in clf.state_dict().items()}
    oracle = qwen3_ref.Qwen3AsrOracle(DIMS, sd2)
    n_frames = 260
    mel = torch.randn(DIMS.n_mels, n_frames, generator=torch.Generator().manual_seed(9))
    n_audio = qwen3_ref.audio_token_count(n_frames, DIMS.n_window)
    prompt = [11] + [DIMS.audio_token_id] * n_audio + [20, 5, 21, 22, 5, 5, 23, 5]
    feats = torch.zeros(1, DIMS.n_mels, 300)
    feats[0, :, :n_frames] = mel
    mask = torch.zeros(1, 300, dtype=torch.long)
    mask[0, :n_frames] = 1
    with torch.no_grad():
        ref = clf(input_ids=torch.tensor([prompt]), input_features=feats, input_features_mask=mask).logits[0]
        x = oracle.embed(prompt, oracle.audio_tokens(mel))
        head_b = torch.from_numpy(sd2["score.bias"]) if "score.bias" in sd2 else None
        got = oracle.classify(x, torch.from_numpy(sd2["score.weight"]), head_b)
    assert got.shape == ref.shape and float((got - ref).abs().max()) < 5e-4
