"""Qwen3 decoder on the MI355X (csrc/qwen.hip through the C ABI) against oracle/qwen3_ref.py (first slice):
ragged batched prefill from embeddings (with ``<audio>`` rows replaced), logits of the last prompt position, greedy
generation until EOS.  float32: logits within 2e-4 and identical tokens; float16 / bfloat16: argmax agreement and log-probs
within the type's bound."""
import numpy as np
import pytest
import torch

from oracle import qwen3_ref

pytestmark = pytest.mark.gpu


def _setup(dtype, seed=7):
    from whisperjav_amd import qwen
    d = qwen.Qwen3Dims(hidden=256, n_layer=3, n_head=4, n_kv_head=2, head_dim=128, ffn=640, vocab=4096, rope_theta=10000.0,
                       audio_token_id=9, eos_token_ids=(1, 2))
    w = qwen.synth_weights(d, seed=seed)
    od = qwen3_ref.Qwen3AsrDims(d=d.hidden, layers=d.n_layer, heads=d.n_head, kv_heads=d.n_kv_head, head_dim=d.head_dim, ffn=d.ffn,
                                vocab=d.vocab, rope_theta=d.rope_theta, rms_eps=d.rms_eps, audio_token_id=d.audio_token_id,
                                eos_token_ids=d.eos_token_ids)
    oracle = qwen3_ref.Qwen3AsrOracle(od, w)
    model = qwen.HipQwen3Decoder(d, w, dtype=dtype, max_seqs=4, max_ctx=256)
    return d, w, oracle, model


def _prompts(d, rng):
    out = []
    for n_audio, n_text in ((37, 5), (80, 3), (0, 9), (130, 4)):      # ragged; the last one spans three 64-key attention chunks
        ids = [11, 12] + [d.audio_token_id] * n_audio + rng.integers(20, d.vocab, n_text).tolist()
        audio = torch.from_numpy(rng.standard_normal((n_audio, d.hidden)).astype(np.float32)) if n_audio else None
        out.append((ids, audio))
    return out


@pytest.mark.parametrize("dtype", ["float32", "float16", "bfloat16"])
def test_prefill_logits_and_greedy_generation_match_oracle(hip, dtype):
    d, w, oracle, model = _setup(dtype)
    rng = np.random.default_rng(3)
    prompts = _prompts(d, rng)
    embeds = [model.prompt_embeddings(ids, audio) for ids, audio in prompts]
    logits = model.prefill(embeds, want_logits=True).cpu()
    res = model.generate(max_new_tokens=24)
    tol = {"float32": 2e-4, "float16": 3e-2, "bfloat16": 0.25}[dtype]
    agree = 0
    with torch.no_grad():
        for b, (ids, audio) in enumerate(prompts):
            x = oracle.embed(ids, audio)
            assert torch.allclose(embeds[b].cpu(), x, atol=1e-6 if dtype == "float32" else 2e-2)
            ref = oracle.logits(x)[-1]
            err = float((logits[b] - ref).abs().max())
            assert err < tol * max(1.0, float(ref.abs().max())), (b, err)
            toks, lps = oracle.greedy(ids, audio, 24)
            got = res.tokens[b]
            n = next((i for i, (a, c) in enumerate(zip(got, toks)) if a != c), min(len(got), len(toks)))
            agree += got == toks
            if dtype == "float32":
                assert got == toks, (b, got, toks)
                assert len(res.token_logprob[b]) == len(lps)
                assert np.abs(np.array(res.token_logprob[b]) - np.array(lps)).max() < 1e-3
            else:
                assert n >= min(4, len(toks)), (b, got, toks)
                assert np.abs(np.array(res.token_logprob[b][:n]) - np.array(lps[:n])).max() < (0.05 if dtype == "float16" else 0.4)
    if dtype == "float16":
        assert agree >= 3
    model.close()


def test_batched_equals_single_and_context_limits(hip):
    from whisperjav_amd import hipbind
    d, w, oracle, model = _setup("float32", seed=8)
    rng = np.random.default_rng(5)
    prompts = _prompts(d, rng)
    embeds = [model.prompt_embeddings(ids, audio) for ids, audio in prompts]
    model.prefill(embeds)
    batch = model.generate(max_new_tokens=16)
    for b in range(len(prompts)):
        model.prefill([embeds[b]])
        one = model.generate(max_new_tokens=16)
        assert one.tokens[0] == batch.tokens[b]
        assert np.abs(np.array(one.token_logprob[0]) - np.array(batch.token_logprob[b])).max() < 1e-4
    with pytest.raises(hipbind.WjError, match="context"):
        model.prefill([torch.zeros((300, d.hidden), device="cuda")])
    with pytest.raises(ValueError, match="placeholders"):
        model.prompt_embeddings([d.audio_token_id] * 3, torch.zeros((2, d.hidden)))
    model.close()
