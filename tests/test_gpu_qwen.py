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


def _audio_setup(dtype):
    from whisperjav_amd import qwen
    ad = qwen.Qwen3AudioDims(n_layer=2, n_head=2, ffn=256, d_model=128, conv_hidden=16, out_dim=256, n_window_infer=400)
    aw = qwen.synth_audio_weights(ad, seed=11)
    od = qwen3_ref.Qwen3AsrDims(n_mels=128, a_layers=ad.n_layer, a_heads=ad.n_head, a_ffn=ad.ffn, a_d=ad.d_model, n_window=50,
                                n_window_infer=ad.n_window_infer, conv_hidden=ad.conv_hidden, d=ad.out_dim)
    return ad, aw, qwen3_ref.Qwen3AsrOracle(od, aw), qwen.HipQwenAudioTower(ad, aw, dtype=dtype, max_seconds=40)


@pytest.mark.parametrize("dtype", ["float32", "float16"])
def test_audio_tower_matches_oracle(hip, dtype):
    """clips -> HIP log-mel (RAW mode) -> convolution stem as patch GEMMs -> windowed-attention layers -> projector, a batch
    of ragged clips (a sub-second one, one ending mid-chunk, one spanning several attention windows) against
    ``oracle.qwen3_ref.audio_tokens`` on the oracle's own log-mel."""
    from oracle import logmel
    from whisperjav_amd import synth
    ad, aw, oracle, tower = _audio_setup(dtype)
    clips = [synth.speech_like(s, seed=60 + i) for i, s in enumerate((0.3, 2.37, 11.5, 5.0))]
    got = tower.encode(clips)
    mel, frames = tower.features(clips)
    tol = 2e-3 if dtype == "float32" else 6e-2
    for i, c in enumerate(clips):
        padded = np.pad(c, (0, max(0, 8000 - len(c))))
        ref_mel = logmel.logmel_ow(padded, 128, padding=0)
        assert ref_mel.shape[1] == frames[i]
        assert np.abs(mel[i, :, : frames[i]].cpu().numpy() - ref_mel).max() < 2e-4
        with torch.no_grad():
            ref = oracle.audio_tokens(torch.from_numpy(ref_mel))
        assert got[i].shape == ref.shape == (qwen3_ref.audio_token_count(int(frames[i])), ad.out_dim)
        err = float((got[i].cpu() - ref).abs().max())
        assert err < tol * max(1.0, float(ref.abs().max())), (i, err, float(ref.abs().max()))
    tower.close()


def test_text_generator_end_to_end_on_the_device(hip, tmp_path):
    """TextGenerator surface (protocols.py:60-110): scene files -> device audio tower -> prompt with <audio> rows -> device
    decoder -> token ids, equal to the oracle's greedy generation on the oracle's audio tokens (float32)."""
    import wave
    from oracle import logmel
    from whisperjav_amd import qwen, synth
    d = qwen.Qwen3Dims(hidden=256, n_layer=2, n_head=2, n_kv_head=1, head_dim=128, ffn=512, vocab=2048, rope_theta=10000.0,
                       audio_token_id=9, eos_token_ids=(1, 2))
    ad = qwen.Qwen3AudioDims(n_layer=2, n_head=2, ffn=256, d_model=128, conv_hidden=16, out_dim=d.hidden, n_window_infer=400)
    w = {**qwen.synth_weights(d, seed=5), **qwen.synth_audio_weights(ad, seed=6)}
    od = qwen3_ref.Qwen3AsrDims(n_mels=128, a_layers=2, a_heads=2, a_ffn=256, a_d=128, n_window=50, n_window_infer=400, conv_hidden=16,
                                d=d.hidden, layers=d.n_layer, heads=d.n_head, kv_heads=d.n_kv_head, head_dim=128, ffn=d.ffn, vocab=d.vocab,
                                rope_theta=d.rope_theta, audio_token_id=9, eos_token_ids=(1, 2))
    oracle = qwen3_ref.Qwen3AsrOracle(od, w)
    build = lambda n_audio, language, context: [11, 12] + [d.audio_token_id] * n_audio + [13, 14]      # noqa: E731
    gen = qwen.HipQwenTextGenerator(d, w, audio_dims=ad, prompt_builder=build, detokenize=lambda t: " ".join(map(str, t)),
                                    dtype="float32", batch_size=2, max_ctx=256, max_new_tokens=12)
    paths = []
    for i, s in enumerate((1.7, 3.2, 0.9)):
        audio = synth.speech_like(s, seed=80 + i)
        pcm = np.clip(np.rint(audio * 32767), -32768, 32767).astype("<i2")
        path = tmp_path / f"scene_{i}.wav"
        with wave.open(str(path), "wb") as wf:
            wf.setnchannels(1); wf.setsampwidth(2); wf.setframerate(16000); wf.writeframes(pcm.tobytes())
        paths.append(path)
    results = gen.generate_batch(paths, language="ja")
    assert len(results) == 3 and all(r.language == "ja" for r in results)
    for path, res in zip(paths, results):
        with wave.open(str(path), "rb") as wf:
            audio = np.frombuffer(wf.readframes(wf.getnframes()), dtype="<i2").astype(np.float32) / 32768.0
        padded = np.pad(audio, (0, max(0, 8000 - len(audio))))
        with torch.no_grad():
            a = oracle.audio_tokens(torch.from_numpy(logmel.logmel_ow(padded, 128, padding=0)))
            toks, _ = oracle.greedy(build(a.shape[0], "ja", None), a, 12)
        assert res.text == " ".join(map(str, toks)), (path.name, res.text, toks)
    gen.cleanup()
