"""Pin the transformer oracle against transformers' independent Whisper implementation."""
import numpy as np
import pytest
import torch

from tests import helpers
from oracle import whisper_ref


@pytest.fixture(scope="module")
def pair():
    from transformers import WhisperConfig, WhisperForConditionalGeneration
    d = helpers.small_dims(n_mels=80, d_model=128, heads=2, layers=2, n_vocab=51865)
    oracle, w = helpers.make_oracle(d, seed=11)
    cfg = WhisperConfig(vocab_size=d.n_vocab, num_mel_bins=d.n_mels, d_model=d.n_audio_state,
                        encoder_layers=d.n_audio_layer, decoder_layers=d.n_text_layer,
                        encoder_attention_heads=d.n_audio_head, decoder_attention_heads=d.n_text_head,
                        encoder_ffn_dim=4 * d.n_audio_state, decoder_ffn_dim=4 * d.n_text_state,
                        max_source_positions=d.n_audio_ctx, max_target_positions=d.n_text_ctx,
                        activation_function="gelu", dropout=0.0, attention_dropout=0.0,
                        activation_dropout=0.0, attn_implementation="eager")
    hf = WhisperForConditionalGeneration(cfg).eval()
    missing, unexpected = hf.load_state_dict(helpers.hf_state_dict(d, w), strict=False)
    assert not unexpected, unexpected
    assert all("k_proj.bias" in m or m == "proj_out.weight" for m in missing), missing
    return d, oracle, hf


def test_encoder_matches_hf(pair):
    d, oracle, hf = pair
    mel = torch.from_numpy(helpers.synth_mel(2, d.n_mels))
    with torch.no_grad():
        ref = hf.model.encoder(mel).last_hidden_state
        got = oracle.encode(mel)
    assert got.shape == (2, 1500, d.n_audio_state)
    assert torch.allclose(got, ref, atol=2e-4, rtol=1e-4), (got - ref).abs().max()


def test_decoder_logits_match_hf(pair):
    d, oracle, hf = pair
    mel = torch.from_numpy(helpers.synth_mel(2, d.n_mels, seed=3))
    toks = torch.tensor([[50258, 50266, 50359, 50363, 1000, 2000, 345],
                         [50258, 50266, 50359, 50363, 17, 50000, 9]])
    with torch.no_grad():
        xa = oracle.encode(mel)
        ref = hf(input_features=mel, decoder_input_ids=toks).logits
        got = oracle.decoder_logits(toks, xa)
    assert torch.allclose(got, ref, atol=5e-4, rtol=1e-4), (got - ref).abs().max()


def test_cached_decoder_equals_full_pass(pair):
    d, oracle, _ = pair
    mel = torch.from_numpy(helpers.synth_mel(1, d.n_mels, seed=5))
    toks = torch.tensor([[50258, 50266, 50359, 7, 8, 9]])
    with torch.no_grad():
        xa = oracle.encode(mel)
        full = oracle.decoder_logits(toks, xa)
        dec = whisper_ref.CachedDecoder(oracle, xa)
        a = dec.step(toks[:, :3])
        b = dec.step(toks[:, 3:4])
        c = dec.step(toks[:, 4:5])
    assert torch.allclose(a, full[:, 2], atol=1e-4)
    assert torch.allclose(b, full[:, 3], atol=1e-4)
    assert torch.allclose(c, full[:, 4], atol=1e-4)
    # logits spread is wide enough for stable argmax (design goal of synth_weights)
    top2 = full[0, -1].topk(2).values
    assert full[0, -1].std() > 0.5 and (top2[0] - top2[1]) > 1e-3


def test_hf_checkpoint_directory_loads_with_the_right_names(tmp_path):
    """weights.load_hf_checkpoint on a directory written by transformers' own save_pretrained: the oracle run on
    the imported tensors reproduces the HF model's logits (pins the name mapping and the dims parsing), and the
    engine's packer accepts the result."""
    import json
    from transformers import WhisperConfig, WhisperForConditionalGeneration
    from whisperjav_amd import weights as pweights
    torch.manual_seed(3)
    cfg = WhisperConfig(vocab_size=51865, num_mel_bins=80, d_model=128, encoder_layers=2, decoder_layers=2,
                        encoder_attention_heads=2, decoder_attention_heads=2, encoder_ffn_dim=512, decoder_ffn_dim=512,
                        max_source_positions=1500, max_target_positions=448)
    hf = WhisperForConditionalGeneration(cfg).eval()
    hf.save_pretrained(tmp_path, safe_serialization=True)
    with open(tmp_path / "generation_config.json") as f:
        gen = json.load(f)
    gen["alignment_heads"] = [[1, 0], [1, 1]]
    with open(tmp_path / "generation_config.json", "w") as f:
        json.dump(gen, f)
    dims, sd, extras = pweights.load_hf_checkpoint(str(tmp_path))
    assert (dims.n_mels, dims.n_audio_state, dims.n_text_layer, dims.n_vocab) == (80, 128, 2, 51865)
    assert extras["alignment_heads"] == [(1, 0), (1, 1)]
    oracle = whisper_ref.WhisperOracle(helpers.oracle_dims(dims), sd)
    mel = torch.from_numpy(helpers.synth_mel(1, 80, seed=2))
    toks = torch.tensor([[50258, 50266, 50359, 50363, 11, 12]])
    with torch.no_grad():
        ref = hf(input_features=mel, decoder_input_ids=toks).logits
        got = oracle.decoder_logits(toks, oracle.encode(mel))
    assert float((got - ref).abs().max()) < 5e-4
    blob, offsets = pweights.pack_blob(dims, sd, "bfloat16")
    assert len(offsets) == pweights.expected_tensor_count(dims)


def test_timestamp_rules_match_transformers_logits_processor():
    """oracle/decoding.filter_logits (ApplyTimestampRules + SuppressTokens + SuppressBlank) against transformers'
    independent WhisperTimeStampLogitsProcessor / SuppressTokens(AtBegin)LogitsProcessor on random logits and
    histories covering every rule branch (first step, open pair, closed pair, monotonic floor, max initial
    timestamp, timestamp-mass rule)."""
    from types import SimpleNamespace
    from transformers.generation.logits_process import (SuppressTokensAtBeginLogitsProcessor, SuppressTokensLogitsProcessor,
                                                        WhisperTimeStampLogitsProcessor)
    from oracle import decoding
    V = 51865
    lay = decoding.TokenLayout.for_vocab(V)
    tb, P = lay.timestamp_begin, 3
    gen_cfg = SimpleNamespace(no_timestamps_token_id=lay.no_timestamps, eos_token_id=lay.eot, bos_token_id=lay.eot,
                              max_initial_timestamp_index=50, _detect_timestamp_from_logprob=True)
    ts_proc = WhisperTimeStampLogitsProcessor(gen_cfg, begin_index=P)
    suppress = [1, 2, 7, 8, 50256, lay.sot, lay.no_speech]
    sup_proc = SuppressTokensLogitsProcessor(suppress)
    blank_proc = SuppressTokensAtBeginLogitsProcessor([lay.blank, lay.eot], begin_index=P)
    prompt = [lay.sot, lay.sot + 8, lay.sot + 101]
    histories = [
        [],                                   # first step: only initial timestamps <= 1.0 s
        [tb],                                 # single opening timestamp
        [tb, 500, 600],                       # inside a segment
        [tb, 500, tb + 120],                  # pair open: text forbidden
        [tb, 500, tb + 120, tb + 120],        # pair closed: no third timestamp, floor at tb + 121
        [tb, 500, tb + 120, tb + 120, 41],    # text after a closed pair
        [tb + 3, tb + 3, tb + 700],           # odd but reachable
        [tb + 1499],                          # last timestamp value
    ]
    rng = np.random.default_rng(0)
    cfg = decoding.FilterConfig(suppress_tokens=tuple(suppress), max_initial_timestamp_index=50, suppress_blank=True)
    checked = 0
    for trial in range(6):
        for gen in histories:
            logits = torch.from_numpy(rng.standard_normal((1, V)).astype(np.float32) * 3.0)
            if trial % 2:        # push probability mass onto the timestamps: exercises the timestamp-mass rule both ways
                logits[0, tb:] += 4.0
            ids = torch.tensor([prompt + gen])
            ref = blank_proc(ids, sup_proc(ids, logits))
            ref = ts_proc(ids, ref)
            got = decoding.filter_logits(logits, [prompt + gen], P, lay, cfg)
            assert torch.equal(torch.isinf(got), torch.isinf(ref)), (trial, gen)
            keep = ~torch.isinf(ref)
            assert torch.equal(got[keep], ref[keep])
            checked += 1
    assert checked == 48


def test_greedy_loop_to_eot_matches_a_transformers_driven_loop():
    """Loop-level pin of ``oracle.decoding.greedy_decode`` on EOT-bearing weights (``weights.SPEECHLIKE``: hypotheses END, at
    lengths that grow with the audio in the window): transformers' model computes the logits of the whole history at every
    step (no cache of ours), transformers' own processors (SuppressTokens, SuppressTokensAtBegin, WhisperTimeStamp) filter
    them, arg-max until ``eos``.  Tokens, stop position and the summed log-prob must equal the oracle's cached loop."""
    from types import SimpleNamespace
    from transformers import WhisperConfig, WhisperForConditionalGeneration
    from transformers.generation.logits_process import (SuppressTokensAtBeginLogitsProcessor, SuppressTokensLogitsProcessor,
                                                        WhisperTimeStampLogitsProcessor)
    from oracle import decoding, logmel
    from whisperjav_amd import dims as pdims, synth, weights as pweights
    d = helpers.small_dims(n_mels=80, d_model=64, heads=1, layers=1, n_vocab=51865)
    oracle, w = helpers.make_oracle(d, seed=5, **pweights.SPEECHLIKE)
    cfg = WhisperConfig(vocab_size=d.n_vocab, num_mel_bins=d.n_mels, d_model=d.n_audio_state, encoder_layers=d.n_audio_layer,
                        decoder_layers=d.n_text_layer, encoder_attention_heads=d.n_audio_head, decoder_attention_heads=d.n_text_head,
                        encoder_ffn_dim=4 * d.n_audio_state, decoder_ffn_dim=4 * d.n_text_state, max_source_positions=d.n_audio_ctx,
                        max_target_positions=d.n_text_ctx, activation_function="gelu", dropout=0.0, attention_dropout=0.0,
                        activation_dropout=0.0, attn_implementation="eager")
    hf = WhisperForConditionalGeneration(cfg).eval()
    _, unexpected = hf.load_state_dict(helpers.hf_state_dict(d, w), strict=False)
    assert not unexpected
    lay = decoding.TokenLayout.for_vocab(d.n_vocab)
    toks = pdims.special_tokens(d.n_vocab)
    prompt = [toks.sot, toks.language_token(pdims.language_index("ja")), toks.transcribe]
    suppress = (toks.sot, toks.translate, toks.transcribe, toks.sot_lm, toks.sot_prev, toks.no_speech)
    fcfg = decoding.FilterConfig(suppress_tokens=suppress, max_initial_timestamp_index=50, suppress_blank=True)
    gen_cfg = SimpleNamespace(no_timestamps_token_id=lay.no_timestamps, eos_token_id=lay.eot, bos_token_id=lay.eot,
                              max_initial_timestamp_index=50, _detect_timestamp_from_logprob=True)
    procs = [SuppressTokensLogitsProcessor(list(suppress)), SuppressTokensAtBeginLogitsProcessor([lay.blank, lay.eot], begin_index=len(prompt)),
             WhisperTimeStampLogitsProcessor(gen_cfg, begin_index=len(prompt))]
    mel = torch.from_numpy(np.stack([logmel.window_features(synth.speech_like(s, seed=40 + i), 80, "fw") for i, s in enumerate((1.0, 4.0))]))
    max_new = 40
    with torch.no_grad():
        xa = oracle.encode(mel)
        out = decoding.greedy_decode(oracle, xa, prompt, max_new, fcfg)
        lengths = set()
        for b in range(mel.shape[0]):
            enc = hf.model.encoder(mel[b:b + 1])
            hist, total = list(prompt), 0.0
            for _ in range(max_new):
                ids = torch.tensor([hist])
                logits = hf(encoder_outputs=enc, decoder_input_ids=ids).logits[:, -1].float()
                for p in procs:
                    logits = p(ids, logits)
                lp = torch.log_softmax(logits, -1)
                t = int(lp.argmax())
                total += float(lp[0, t])
                hist.append(t)
                if t == lay.eot:
                    break
            ref = [t for t in hist[len(prompt):] if t != lay.eot]
            assert hist[-1] == lay.eot, "this window did not end within max_new: the fixture lost its point"
            got = [int(t) for t in out.tokens[b]]
            assert got == ref, (b, got, ref)
            assert abs(float(out.sum_logprob[b]) - total) < 2e-3, (b, float(out.sum_logprob[b]), total)
            lengths.add(len(ref))
    assert len(lengths) == 2          # the two windows ended at different lengths


def test_sharpened_logits_is_the_same_model_at_a_lower_temperature():
    """weights.sharpened_logits (the fidelity bench workload: hypotheses that pass the reference's avg_logprob > -1.0 gate): three
    tensors change, the greedy tokens up to the end-of-text decision stay the ones of the original model, the per-token
    log-probs rise from the -3 class to above -1, and patching the packed blob equals packing the updated weights."""
    import torch
    from oracle import decoding, logmel
    from whisperjav_amd import dims as pdims, synth, weights as pweights
    d = helpers.small_dims()
    w = pweights.synth_weights(d, seed=33, **pweights.SPEECHLIKE)
    up = pweights.sharpened_logits(d, w, 33, pweights.SPEECHLIKE["eot"], 1.8, 2.5)
    assert sorted(up) == ["decoder.ln.bias", "decoder.ln.weight", "decoder.positional_embedding"]
    w2 = dict(w)
    w2.update(up)
    toks = pdims.special_tokens(d.n_vocab)
    prompt = [toks.sot, toks.language_token(pdims.language_index("ja")), toks.transcribe]
    sup = (toks.sot, toks.translate, toks.transcribe, toks.sot_lm, toks.sot_prev, toks.no_speech)
    cfg = decoding.FilterConfig(suppress_tokens=sup, max_initial_timestamp_index=50)
    mel = torch.from_numpy(logmel.window_features(synth.speech_like(3.0, seed=20), 80, "fw")[None])
    out = []
    for ww in (w, w2):
        o = whisper_ref.WhisperOracle(helpers.oracle_dims(d), ww)
        with torch.no_grad():
            out.append(decoding.greedy_decode(o, o.encode(mel), prompt, 48, cfg))
    a, b = out
    n = min(len(a.tokens[0]), len(b.tokens[0])) - 1
    assert n >= 8 and a.tokens[0][:n] == b.tokens[0][:n]
    assert np.mean(a.token_logprob[0]) < -2.0 and np.mean(b.token_logprob[0]) > -1.0
    blob, offs = pweights.pack_blob(d, w, "float16")
    want, offs2 = pweights.pack_blob(d, w2, "float16")
    got = pweights.patch_blob_device(blob, offs, d, up)
    assert np.array_equal(offs, offs2) and bool((got == want).all()) and not bool((blob == want).all())
