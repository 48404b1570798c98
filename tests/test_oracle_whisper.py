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
