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
    assert model._lib.wj_qwen_last_used_graph(model.handle) == 1          # the decode iteration is replayed from a hipGraph
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
                                    dtype="float32", batch_size=2, max_ctx=256, max_new_tokens=12, max_tokens_per_audio_second=3.0,
                                    min_tokens_floor=2)
    assert gen.repetition_penalty == 1.1          # the reference generator's default (generators/qwen3.py:39)
    seconds = (1.7, 3.2, 0.9)
    budgets = [qwen.dynamic_token_limit(s, 12, 3.0, 2) for s in seconds]
    assert budgets == [5, 9, 2]
    paths = []
    for i, s in enumerate(seconds):
        audio = synth.speech_like(s, seed=80 + i)
        pcm = np.clip(np.rint(audio * 32767), -32768, 32767).astype("<i2")
        path = tmp_path / f"scene_{i}.wav"
        with wave.open(str(path), "wb") as wf:
            wf.setnchannels(1); wf.setsampwidth(2); wf.setframerate(16000); wf.writeframes(pcm.tobytes())
        paths.append(path)
    results = gen.generate_batch(paths, language="ja", audio_durations=list(seconds))     # what the orchestrator passes (step 3)
    assert len(results) == 3 and all(r.language == "ja" for r in results)
    for path, res, budget in zip(paths, results, budgets):
        with wave.open(str(path), "rb") as wf:
            audio = np.frombuffer(wf.readframes(wf.getnframes()), dtype="<i2").astype(np.float32) / 32768.0
        padded = np.pad(audio, (0, max(0, 8000 - len(audio))))
        with torch.no_grad():
            a = oracle.audio_tokens(torch.from_numpy(logmel.logmel_ow(padded, 128, padding=0)))
            toks, _ = oracle.greedy(build(a.shape[0], "ja", None), a, budget, repetition_penalty=1.1)
        assert res.text == " ".join(map(str, toks)), (path.name, res.text, toks)
        assert res.metadata["n_tokens"] <= budget
    # pooling seam (qwen_pipeline.HipDecoupledSubtitlePipeline): every clip announced once, the orchestrator's per-scene calls
    # are answered from the pooled results -- same texts as the per-scene computation above
    gen.prime(paths, language="ja", audio_durations=list(seconds))
    assert len(gen._primed) == 3
    one_by_one = [gen.generate_batch([p], language="ja", audio_durations=[s])[0] for p, s in zip(paths, seconds)]
    assert [r.text for r in one_by_one] == [r.text for r in results] and not gen._primed
    assert gen.generate_batch([paths[1]], language="ja", audio_durations=[seconds[1]])[0].text == results[1].text      # nothing primed: computed
    gen.cleanup()


def test_forced_aligner_on_the_device(hip, tmp_path):
    """TextAligner surface (protocols.py:128-179): one full pass + the linear head over time bins at the <timestamp> markers
    (``wj_qwen_classify``) == the oracle's ``classify`` (float32: identical bins, logits 2e-3), then the monotonic repair
    and the bins -> seconds conversion of the upstream processor."""
    import wave
    from oracle import logmel
    from whisperjav_amd import qwen, synth
    d = qwen.Qwen3Dims(hidden=256, n_layer=2, n_head=2, n_kv_head=1, head_dim=128, ffn=512, vocab=2048, rope_theta=10000.0,
                       audio_token_id=9, eos_token_ids=(1, 2))
    ad = qwen.Qwen3AudioDims(n_layer=2, n_head=2, ffn=256, d_model=128, conv_hidden=16, out_dim=d.hidden, n_window_infer=400)
    rng = np.random.default_rng(2)
    w = {**qwen.synth_weights(d, seed=15), **qwen.synth_audio_weights(ad, seed=16),
         "score.weight": (rng.standard_normal((96, d.hidden)) * 2.0 / np.sqrt(d.hidden)).astype(np.float32),
         "score.bias": (0.1 * rng.standard_normal(96)).astype(np.float32)}
    od = qwen3_ref.Qwen3AsrDims(n_mels=128, a_layers=2, a_heads=2, a_ffn=256, a_d=128, n_window=50, n_window_infer=400, conv_hidden=16,
                                d=d.hidden, layers=d.n_layer, heads=d.n_head, kv_heads=d.n_kv_head, head_dim=128, ffn=d.ffn, vocab=d.vocab,
                                rope_theta=d.rope_theta, audio_token_id=9, eos_token_ids=(1, 2))
    oracle = qwen3_ref.Qwen3AsrOracle(od, w)
    TS = 5                                                               # the <timestamp> marker's token id in this toy vocabulary

    def word_prompt(n_audio, words, language):
        ids = [11] + [d.audio_token_id] * n_audio
        marks = []
        for wd in words:
            ids += [100 + (hash(wd) % 900)]
            marks += [len(ids), len(ids) + 1]
            ids += [TS, TS]
        return ids, marks
    strip = str.maketrans("", "", "「」、。")
    split = lambda text, language: text.translate(strip).split()      # noqa: E731  the upstream splitter drops punctuation
    al = qwen.HipQwenForcedAligner(d, ad, w, segment_ms=80.0, word_prompt=word_prompt, split_words=split, dtype="float32", batch_size=2,
                                   max_ctx=256)
    paths, texts = [], ["「ka、 ki ku。 ke」", "sa shi su", "ta", "  "]          # the last scene has no text: not aligned
    for i, s in enumerate((2.1, 1.3, 0.8, 0.6)):
        audio = synth.speech_like(s, seed=90 + i)
        path = tmp_path / f"a{i}.wav"
        with wave.open(str(path), "wb") as wf:
            wf.setnchannels(1); wf.setsampwidth(2); wf.setframerate(16000)
            wf.writeframes(np.clip(np.rint(audio * 32767), -32768, 32767).astype("<i2").tobytes())
        paths.append(path)
    results = al.align_batch(paths, texts, language="ja")
    assert len(results) == 4 and results[3].words == [] and results[3].metadata == {"scene_index": 3, "skipped": True}
    assert [w_.word for w_ in results[0].words] == ["「ka、 ", "ki ", "ku。 ", "ke」"]      # punctuation back from the transcript
    for path, text, res in list(zip(paths, texts, results))[:3]:
        with wave.open(str(path), "rb") as wf:
            audio = np.frombuffer(wf.readframes(wf.getnframes()), dtype="<i2").astype(np.float32) / 32768.0
        padded = np.pad(audio, (0, max(0, 8000 - len(audio))))
        words = split(text, "ja")
        with torch.no_grad():
            a = oracle.audio_tokens(torch.from_numpy(logmel.logmel_ow(padded, 128, padding=0)))
            ids, marks = word_prompt(a.shape[0], words, "ja")
            lg = oracle.classify(oracle.embed(ids, a), torch.from_numpy(w["score.weight"]), torch.from_numpy(w["score.bias"]))
        bins = lg[marks].argmax(-1).numpy()
        assert res.metadata["raw_bins"] == bins.tolist(), (path.name, res.metadata["raw_bins"], bins.tolist())
        ms = qwen3_ref.fix_timestamps(bins.astype(np.float64) * 80.0)
        assert [w_.word.translate(strip).strip() for w_ in res.words] == words
        assert res.metadata["raw_word_count"] == res.metadata["merged_word_count"] == len(words)
        assert [(w_.start, w_.end) for w_ in res.words] == [(round(ms[2 * i] / 1000.0, 3), round(ms[2 * i + 1] / 1000.0, 3)) for i in range(len(words))]
    al.prime(paths[:3], texts[:3], language="ja")                       # pooling seam: per-scene calls answered from one pooled pass
    for i in range(3):
        again = al.align_batch([paths[i]], [texts[i]], language="ja")[0]
        assert [(w_.word, w_.start, w_.end) for w_ in again.words] == [(w_.word, w_.start, w_.end) for w_ in results[i].words]
        assert again.metadata["scene_index"] == 0
    assert not al._primed
    al.cleanup()


def _bf16_representable(w):
    """Matrices as a published bfloat16 checkpoint holds them (Qwen3-ASR ships bf16): exactly representable in float16 too."""
    out = {}
    for k, v in w.items():
        out[k] = torch.from_numpy(v).to(torch.bfloat16).to(torch.float32).numpy() if v.ndim >= 2 else v
    return out


@pytest.mark.parametrize("split", [5])      # 2, 3 and 4 are valid values too (pass `-k` nothing: edit the list); their measurements are kept in
                                              # profiles/r05_parity_diag_qwen_split_ablation.jsonl -- each costs ~60 s of GPU-box time in a suite the driver caps
def test_published_geometry_one_clip_within_the_north_star_bar(hip, split):
    """Qwen3-ASR-1.7B's own dimensions (audio tower 24 x 1024 / conv 480, decoder 28 x 2048, 16 / 8 heads of 128, vocabulary
    151 936) on seeded bf16-representable weights with the EOS ramp, float16 on the device against the fp32 oracle, a 4 s clip
    decoded TO EOS.  Round 4 bar: tokens identical and every per-token log-prob within the north-star's
    1e-3 for the decoder fed the oracle's audio embeddings (decoder parity), and end to end through the device tower."""
    import psutil
    if psutil.virtual_memory().available < 40 * 2 ** 30:
        pytest.skip("needs ~30 GB of host memory for the 1.7 B-parameter fp32 weights and their fp16 blob")
    from oracle import logmel
    from whisperjav_amd import qwen, synth
    d, ad = qwen.Qwen3Dims(), qwen.Qwen3AudioDims()
    ramp = qwen.QwenEosRamp.for_dims(d)
    w = _bf16_representable({**qwen.synth_weights(d, seed=1, eos=ramp), **qwen.synth_audio_weights(ad, seed=2, ramp=ramp)})
    od = qwen3_ref.Qwen3AsrDims()
    oracle = qwen3_ref.Qwen3AsrOracle(od, w)
    from whisperjav_amd import hipbind
    tower = qwen.HipQwenAudioTower(ad, w, dtype="float16", max_seconds=8)
    hipbind.tune("qwen_split_act", split)
    try:
        model = qwen.HipQwen3Decoder(d, w, dtype="float16", max_seqs=1, max_ctx=256)
    finally:
        hipbind.tune("qwen_split_act", 5)
    clip = synth.speech_like(4.0, seed=7)
    a = tower.encode([clip])[0]
    with torch.no_grad():
        ref_a = oracle.audio_tokens(torch.from_numpy(logmel.logmel_ow(clip, 128, padding=0)))
    err_a = float((a.cpu() - ref_a).abs().max()) / max(1.0, float(ref_a.abs().max()))
    ids = [151644, 872] + [d.audio_token_id] * int(a.shape[0]) + [151645, 198, 151644, 77091]
    budget = 120
    with torch.no_grad():
        ref_l = oracle.logits(oracle.embed(ids, ref_a))[-1]
        toks, lps = oracle.greedy(ids, ref_a, budget)
    assert 4 <= len(toks) < budget, len(toks)                 # the oracle's sequence ends on EOS
    rows = {}
    for name, audio in (("decoder", ref_a), ("end_to_end", a)):
        logits = model.prefill([model.prompt_embeddings(ids, audio)], want_logits=True).cpu()[0]
        res = model.generate(max_new_tokens=budget)
        n = next((i for i, (g, r) in enumerate(zip(res.tokens[0], toks)) if g != r), min(len(res.tokens[0]), len(toks)))
        k = min(n + 1, len(lps), len(res.token_logprob[0]))
        errs = np.abs(np.array(res.token_logprob[0][:k]) - np.array(lps[:k]))
        rows[name] = dict(identical=res.tokens[0] == toks, n_tokens=len(res.tokens[0]), first_diff=n,
                          max_logprob_err=float(errs.max()), argmax_err=int(errs.argmax()),
                          max_logprob_err_without_eos_step=float(errs[: len(toks)].max()) if len(toks) else 0.0,
                          worst5=[(int(i), round(float(errs[i]), 5)) for i in np.argsort(-errs)[:5]],
                          prompt_logit_err=float((logits - ref_l).abs().max()), steps=res.steps)
    import json, os
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/diag_qwen.jsonl", "a") as f:
        f.write(json.dumps({"test": "qwen_published_geometry_f16", "split_mode": split, "audio_rel_err": err_a, "oracle_tokens": len(toks),
                            "logit_spread": float(ref_l.std()), **rows}) + "\n")
    print("published geometry:", err_a, rows)
    # The north-star's bar, END TO END (device log-mel -> device tower -> device decoder against the fp32 oracle of all three), in the
    # mode that ships and is benchmarked: split mode 5 (the default since round 5: o_proj / down_proj / LM-head / gate-up inputs as
    # [hi | lo] pairs) with the tower's GEMM inputs split as well (qwen_tower_split, default on: embedding error 4.6e-4 -> 6e-5 of
    # their range, scripts/precision_qwen_tower.py).  Mode 3 (every projection input) meets it too; modes 2 (round 4's default) and
    # 4 (2 + the q/k/v input only) are the ablations that show WHICH input matters -- the gate/up one -- and are recorded with the
    # bound they were measured under (profiles/r05_parity_diag_qwen_split_ablation.jsonl).
    assert err_a < 2.5e-4, err_a
    bar = 1e-3 if split in (5, 3) else 2e-3
    assert rows["decoder"]["identical"] and rows["decoder"]["max_logprob_err"] < bar, rows
    assert rows["decoder"]["steps"] < budget
    assert rows["end_to_end"]["identical"] and rows["end_to_end"]["max_logprob_err"] < bar, rows
    if split == 5:      # the default mode on a SECOND clip (another length, another realisation of the rounding errors): same bar
        clip2 = synth.speech_like(3.0, seed=8)
        a2 = tower.encode([clip2])[0]
        with torch.no_grad():
            ref_a2 = oracle.audio_tokens(torch.from_numpy(logmel.logmel_ow(clip2, 128, padding=0)))
            ids2 = [151644, 872] + [d.audio_token_id] * int(a2.shape[0]) + [151645, 198, 151644, 77091]
            toks2, lps2 = oracle.greedy(ids2, ref_a2, budget)
        model.prefill([model.prompt_embeddings(ids2, a2)])
        res2 = model.generate(max_new_tokens=budget)
        k2 = min(len(lps2), len(res2.token_logprob[0]))
        err2 = float(np.abs(np.array(res2.token_logprob[0][:k2]) - np.array(lps2[:k2])).max())
        with open("gpurun_out/diag_qwen.jsonl", "a") as f:
            f.write(json.dumps({"test": "qwen_published_geometry_f16_second_clip", "split_mode": split, "n_tokens": len(toks2),
                                "identical": res2.tokens[0] == toks2, "end_to_end_max_logprob_err": err2}) + "\n")
        assert res2.tokens[0] == toks2 and err2 < 1e-3, (len(toks2), err2)
    tower.close(); model.close()


def test_packed_prompt_assembly_equals_the_per_prompt_path(hip):
    """``prompt_embeddings_many`` (one embedding launch + one scatter for the batch) and ``prefill_packed`` give what the
    per-prompt calls give, bit for bit, and refuse a placeholder / audio-row mismatch per prompt."""
    d, w, oracle, model = _setup("float32", seed=9)
    rng = np.random.default_rng(6)
    prompts = _prompts(d, rng)
    one_by_one = [model.prompt_embeddings(ids, audio) for ids, audio in prompts]
    packed, n = model.prompt_embeddings_many([p[0] for p in prompts], [p[1] for p in prompts])
    assert n.tolist() == [len(p[0]) for p in prompts]
    assert torch.equal(packed, torch.cat(one_by_one, 0))
    a = model.prefill(one_by_one, want_logits=True).clone()
    b = model.prefill_packed(packed, n, want_logits=True)
    assert torch.equal(a, b)
    with pytest.raises(ValueError, match="prompt 1"):
        model.prompt_embeddings_many([[11, 12], [d.audio_token_id] * 3], [None, torch.zeros((2, d.hidden))])
    model.close()


def test_lm_head_with_1024_rows_runs_on_the_padded_vocabulary(hip):
    """From 1024 sequences on, the tied LM head goes through the 256-wide MFMA tile kernel over the vocabulary padded to a
    multiple of 256 with zero rows (``qwen.engine_tensors``); the padding columns must never be chosen or reach the caller."""
    from whisperjav_amd import qwen
    d = qwen.Qwen3Dims(hidden=256, n_layer=1, n_head=2, n_kv_head=1, head_dim=128, ffn=256, vocab=1000, rope_theta=10000.0,
                       audio_token_id=9, eos_token_ids=(1, 2))
    w = qwen.synth_weights(d, seed=4)
    od = qwen3_ref.Qwen3AsrDims(d=d.hidden, layers=d.n_layer, heads=d.n_head, kv_heads=d.n_kv_head, head_dim=d.head_dim, ffn=d.ffn,
                                vocab=d.vocab, rope_theta=d.rope_theta, rms_eps=d.rms_eps, audio_token_id=d.audio_token_id,
                                eos_token_ids=d.eos_token_ids)
    oracle = qwen3_ref.Qwen3AsrOracle(od, w)
    rng = np.random.default_rng(12)
    S = 1030
    prompts = [rng.integers(20, d.vocab, int(rng.integers(1, 4))).tolist() for _ in range(S)]
    big = qwen.HipQwen3Decoder(d, w, dtype="float16", max_seqs=S, max_ctx=16)
    packed, n = big.prompt_embeddings_many(prompts, [None] * S)
    logits = big.prefill_packed(packed, n, want_logits=True)
    assert logits.shape == (S, d.vocab)
    res = big.generate(max_new_tokens=4)
    assert all(0 <= t < d.vocab for toks in res.tokens for t in toks)
    small = qwen.HipQwen3Decoder(d, w, dtype="float16", max_seqs=8, max_ctx=16)
    for b in (0, 517, S - 1):
        one = small.prefill([small.embed(prompts[b])], want_logits=True)[0]
        assert float((one - logits[b]).abs().max()) < 2e-2          # the tile kernel and the skinny kernel sum in different orders
        with torch.no_grad():
            ref = oracle.logits(oracle.embed(prompts[b], None))[-1]
        assert float((logits[b].cpu() - ref).abs().max()) < 3e-2 * max(1.0, float(ref.abs().max()))
    big.close(); small.close()


def test_audio_tower_slices_a_batch_that_exceeds_its_workspace(hip):
    """``encode`` cuts the batch at ``max_seconds`` one-second chunks; clips are independent, so the slices give what one pass gives."""
    from whisperjav_amd import qwen, synth
    ad = qwen.Qwen3AudioDims(n_layer=2, n_head=2, ffn=256, d_model=128, conv_hidden=16, out_dim=256, n_window_infer=400)
    aw = qwen.synth_audio_weights(ad, seed=11)
    clips = [synth.speech_like(s, seed=70 + i) for i, s in enumerate((2.37, 5.0, 0.3, 3.1, 4.9))]
    whole = qwen.HipQwenAudioTower(ad, aw, dtype="float32", max_seconds=40)
    sliced = qwen.HipQwenAudioTower(ad, aw, dtype="float32", max_seconds=6)
    a, b = whole.encode(clips), sliced.encode(clips)
    assert [tuple(x.shape) for x in a] == [tuple(x.shape) for x in b]
    for x, y in zip(a, b):
        assert float((x - y).abs().max()) < 1e-5 * max(1.0, float(x.abs().max()))
    whole.close(); sliced.close()


def test_generation_controls_match_oracle(hip):
    """The two generation controls the reference's pipeline sets (pipelines/qwen_pipeline.py:157-158): transformers' repetition
    penalty over prompt + generated ids (the oracle's restatement is pinned against ``RepetitionPenaltyLogitsProcessor`` and
    against ``generate(repetition_penalty=...)``) and a token budget per sequence.  float32: tokens identical, the log-probs
    are those of the penalised distribution."""
    d, w, oracle, model = _setup("float32", seed=10)
    rng = np.random.default_rng(8)
    prompts = _prompts(d, rng)
    budgets = [24, 3, 24, 7]
    packed, n = model.prompt_embeddings_many([p[0] for p in prompts], [p[1] for p in prompts])
    model.prefill_packed(packed, n)
    plain = model.generate(max_new_tokens=24)
    model.prefill_packed(packed, n)
    res = model.generate(max_new_tokens=24, repetition_penalty=1.6, prompt_ids=[p[0] for p in prompts], max_new_per_seq=budgets)
    assert model._lib.wj_qwen_last_used_graph(model.handle) == 1
    changed = 0
    with torch.no_grad():
        for b, (ids, audio) in enumerate(prompts):
            toks, lps = oracle.greedy(ids, audio, budgets[b], repetition_penalty=1.6)
            assert res.tokens[b] == toks, (b, res.tokens[b], toks)
            assert len(res.tokens[b]) <= budgets[b]
            assert len(res.token_logprob[b]) == len(lps)
            assert np.abs(np.array(res.token_logprob[b]) - np.array(lps)).max() < 1e-3
            changed += res.tokens[b] != plain.tokens[b][: len(res.tokens[b])]
    assert changed >= 1                      # the penalty changed at least one continuation on this seed
    with pytest.raises(ValueError, match="prompt_ids"):
        model.generate(max_new_tokens=4, repetition_penalty=1.2)
    model.close()


# ---------------------------------------------------------------------------------------------------------------------------
# generations that END (round 4): QwenEosRamp weights -- an EOS logit that rises with every generated token and crosses the
# text logits later for clips with more audio.  Rounds 1-3 never saw a sequence stop before its budget.
# ---------------------------------------------------------------------------------------------------------------------------
RAMP_CLIPS = ((26, 6), (78, 4), (40, 7), (52, 5), (0, 9), (104, 6))       # (audio tokens, text tokens after them)


def _ramp_setup(dtype, seed=7, max_seqs=6, max_ctx=256, exact=False):
    from whisperjav_amd import qwen
    d = qwen.Qwen3Dims(hidden=256, n_layer=3, n_head=4, n_kv_head=2, head_dim=128, ffn=640, vocab=4096, rope_theta=10000.0,
                       audio_token_id=9, eos_token_ids=(1, 2))
    ramp = qwen.QwenEosRamp()
    w = qwen.synth_weights(d, seed=seed, eos=ramp)
    if exact:       # fp16-representable matrices (what a published fp16 / bf16 checkpoint is): the engine's weight rounding is then exact
        w = {k: (v.astype(np.float16).astype(np.float32) if v.ndim == 2 else v) for k, v in w.items()}
    od = qwen3_ref.Qwen3AsrDims(d=d.hidden, layers=d.n_layer, heads=d.n_head, kv_heads=d.n_kv_head, head_dim=d.head_dim, ffn=d.ffn,
                                vocab=d.vocab, rope_theta=d.rope_theta, rms_eps=d.rms_eps, audio_token_id=d.audio_token_id,
                                eos_token_ids=d.eos_token_ids)
    return d, ramp, w, qwen3_ref.Qwen3AsrOracle(od, w), qwen.HipQwen3Decoder(d, w, dtype=dtype, max_seqs=max_seqs, max_ctx=max_ctx)


def _ramp_prompts(d, ramp, rng, clips=RAMP_CLIPS):
    from whisperjav_amd import qwen
    out = []
    for n_audio, n_text in clips:
        ids = [11, 12] + [d.audio_token_id] * n_audio + rng.integers(20, d.vocab, n_text).tolist()
        audio = None
        if n_audio:
            audio = qwen.plant_audio_rows(torch.from_numpy(rng.standard_normal((n_audio, d.hidden)).astype(np.float32)), d, ramp)
        out.append((ids, audio))
    return out


def test_generation_ends_on_eos_ragged_batch_matches_oracle(hip):
    """float32, six clips of different audio lengths in one batch, budget 120: every sequence must END ON EOS at its own
    length (the oracle's), the lengths must differ, the decode loop must leave early, the EOS token's log-prob is reported,
    and a batch must equal its sequences run alone -- each property asserted to have been exercised."""
    d, ramp, w, oracle, model = _ramp_setup("float32")
    prompts = _ramp_prompts(d, ramp, np.random.default_rng(3))
    embeds = [model.prompt_embeddings(ids, audio) for ids, audio in prompts]
    model.prefill(embeds)
    budget = 120
    res = model.generate(max_new_tokens=budget)
    lens = []
    with torch.no_grad():
        for b, (ids, audio) in enumerate(prompts):
            toks, lps = oracle.greedy(ids, audio, budget)
            assert len(toks) < budget and len(lps) == len(toks) + 1, "the oracle's sequence must end on EOS inside the budget"
            assert res.tokens[b] == toks, (b, res.tokens[b], toks)
            assert len(res.token_logprob[b]) == len(toks) + 1                   # + the EOS token's
            assert np.abs(np.array(res.token_logprob[b]) - np.array(lps)).max() < 1e-3
            lens.append(len(toks))
    assert len(set(lens)) >= 4 and min(lens) < max(lens) // 2, lens             # ragged finish inside one batch
    assert lens[1] > lens[3] > lens[2] > lens[0], lens                          # more audio, later EOS (78 > 52 > 40 > 26 audio tokens)
    assert max(lens) < res.steps <= max(lens) + 9 < budget, (res.steps, lens)   # early exit: the poll (every 8 iterations) saw all flags
    assert not any(res.context_limited)
    # finished sequences LEFT the batch (round 4): the batch was re-packed and did fewer row-iterations than rows x iterations;
    # without compaction (wj_tune qwen_compact_pct = 0) the same tokens and log-probs come out
    assert res.compactions >= 2 and res.row_steps < 0.75 * res.steps * len(prompts), (res.compactions, res.row_steps, res.steps)
    from whisperjav_amd import hipbind
    hipbind.tune("qwen_compact_pct", 0)
    try:
        model.prefill(embeds)
        flat = model.generate(max_new_tokens=budget)
    finally:
        hipbind.tune("qwen_compact_pct", 15)
    assert flat.compactions == 0 and flat.row_steps == flat.steps * len(prompts)
    assert flat.tokens == res.tokens and flat.token_logprob == res.token_logprob
    for b in (0, 1, 4):                                                         # batch == single, with EOS
        model.prefill([embeds[b]])
        one = model.generate(max_new_tokens=budget)
        assert one.tokens[0] == res.tokens[b] and one.steps < res.steps + 1
        assert np.abs(np.array(one.token_logprob[0]) - np.array(res.token_logprob[b])).max() < 1e-4
    model.close()


def test_budgets_and_eos_interact_per_sequence(hip):
    """Per-clip budgets (the reference's max_tokens_per_audio_second) together with EOS and the repetition penalty: some
    sequences hit their budget first, others end on EOS first; both kinds in ONE batch, each equal to the oracle run with that
    sequence's budget."""
    d, ramp, w, oracle, model = _ramp_setup("float32", seed=8)
    prompts = _ramp_prompts(d, ramp, np.random.default_rng(5))
    embeds = [model.prompt_embeddings(ids, audio) for ids, audio in prompts]
    budgets = [5, 200, 200, 12, 3, 30]
    model.prefill(embeds)
    res = model.generate(max_new_tokens=100, repetition_penalty=1.1, prompt_ids=[ids for ids, _ in prompts], max_new_per_seq=budgets)
    by_budget = by_eos = 0
    with torch.no_grad():
        for b, (ids, audio) in enumerate(prompts):
            lim = min(budgets[b], 100)
            toks, lps = oracle.greedy(ids, audio, lim, repetition_penalty=1.1)
            assert res.tokens[b] == toks, (b, res.tokens[b], toks)
            assert np.abs(np.array(res.token_logprob[b]) - np.array(lps)).max() < 1e-3
            if len(toks) == lim:
                by_budget += 1
            else:
                by_eos += 1
    assert by_budget >= 2 and by_eos >= 2, (by_budget, by_eos)
    model.close()


def test_budget_is_cut_to_the_room_left_in_the_kv_cache(hip):
    """prompt + budget beyond max_ctx used to overwrite the last cache position and decode garbage.  The budget is
    now cut to ``max_ctx - prompt`` and the cut is reported; inside the room the tokens equal the oracle's."""
    d, ramp, w, oracle, model = _ramp_setup("float32", max_seqs=2, max_ctx=96)
    prompts = _ramp_prompts(d, ramp, np.random.default_rng(9), clips=((78, 4), (26, 6)))       # 84 and 34 prompt tokens
    embeds = [model.prompt_embeddings(ids, audio) for ids, audio in prompts]
    model.prefill(embeds)
    res = model.generate(max_new_tokens=60)
    assert res.context_limited == [True, False]
    with torch.no_grad():
        full0, _ = oracle.greedy(*prompts[0], 60)
        full1, _ = oracle.greedy(*prompts[1], 60)
    assert len(full0) > 12 and res.tokens[0] == full0[:12]          # 96 - 84 positions were left
    assert res.tokens[1] == full1 and len(full1) < 60
    with pytest.raises(Exception, match="prefill first"):           # one generation per prefill: the decode state is consumed
        model.generate(max_new_tokens=4)
    model.embed([5, 6, 7])                                          # embedding between prefill and generate must not disturb the decode rows
    model.prefill(embeds)
    model.embed([5, 6, 7, 8])
    again = model.generate(max_new_tokens=60)
    assert again.tokens == res.tokens
    model.close()


@pytest.mark.parametrize("split", [5, 3, 2, 1, 0])
def test_float16_generation_to_eos_toy_model(hip, split):
    """float16 on fp16-representable weights, six sequences run to EOS on the TOY geometry (3 layers of 256).  With split
    activations (wj_tune qwen_split_act 2: o_proj / down_proj / LM head read [hi | lo] pairs, prompts included; 3: every projection input; 5, the default: 2 + the gate/up input) every sequence and its length must equal the fp32 oracle's and the per-token log-probs sit within
    4e-3 (measured 2.7e-3 at mode 2 against 1.3e-2 without the split).  The north-star's 1e-3 is asserted where the averaging
    over 2048 hidden units exists -- test_published_geometry_one_clip_within_the_north_star_bar (modes 5 -- the default -- and 3) -- a 256-wide
    model sums 8x fewer rounding errors per dot product and its logits are correspondingly noisier, as the Whisper toy model is."""
    from whisperjav_amd import hipbind
    hipbind.tune("qwen_split_act", split)
    try:
        d, ramp, w, oracle, model = _ramp_setup("float16", exact=True)      # noqa: F841
    finally:
        hipbind.tune("qwen_split_act", 5)
    prompts = _ramp_prompts(d, ramp, np.random.default_rng(3))
    embeds = [model.prompt_embeddings(ids, audio) for ids, audio in prompts]
    logits = model.prefill(embeds, want_logits=True).cpu()
    res = model.generate(max_new_tokens=120)
    worst_lp, worst_logit, same = 0.0, 0.0, 0
    with torch.no_grad():
        for b, (ids, audio) in enumerate(prompts):
            toks, lps = oracle.greedy(ids, audio, 120)
            ref = oracle.logits(oracle.embed(ids, audio))[-1]
            worst_logit = max(worst_logit, float((logits[b] - ref).abs().max()))
            n = next((i for i, (a, c) in enumerate(zip(res.tokens[b], toks)) if a != c), min(len(res.tokens[b]), len(toks)))
            same += res.tokens[b] == toks
            k = min(n + 1, len(lps), len(res.token_logprob[b]))
            worst_lp = max(worst_lp, float(np.abs(np.array(res.token_logprob[b][:k]) - np.array(lps[:k])).max()))
    import json, os
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/diag_qwen.jsonl", "a") as f:
        f.write(json.dumps({"test": "qwen_f16_eos", "split": split, "same_sequences": same, "of": len(prompts),
                            "max_logprob_err": worst_lp, "max_prompt_logit_err": worst_logit}) + "\n")
    if split >= 2:
        assert same >= len(prompts) - 1, same      # measured 6 / 6 at mode 2, 5 / 6 at mode 3: one EOS step is a near-tie on this model
        assert worst_lp < 4e-3, worst_lp
        assert worst_logit < 1e-2, worst_logit
    model.close()


# ---------------------------------------------------------------------------------------------------------------------------
# float8w (WJ_F8W): BASELINE cfg5's "fp8 MFMA" -- MX-fp8 projections in every decoder layer
# ---------------------------------------------------------------------------------------------------------------------------
def _mxq(x: torch.Tensor) -> torch.Tensor:
    """MX-fp8 round trip of the rows of ``x`` (OCP e4m3, one power-of-two scale per 32 elements): what the device's quantiser
    followed by the block-scaled MFMA sees of an operand."""
    shp = x.shape
    b = x.reshape(-1, shp[-1] // 32, 32).double()
    amax = b.abs().amax(-1, keepdim=True)
    e = torch.where(amax > 0, torch.floor(torch.log2(amax.clamp_min(1e-300))) - 8, torch.full_like(amax, -127.0)).clamp(-127, 127)
    q = (b / torch.pow(2.0, e)).clamp(-448, 448).float().to(torch.float8_e4m3fn).float().double() * torch.pow(2.0, e)
    return q.reshape(shp).float()


class _Mx8Oracle(qwen3_ref.Qwen3AsrOracle):
    """The fp32 oracle with the device's MX-fp8 rounding points: the inputs of the four projections of every layer and their
    weight matrices go through the MX round trip (fp16 rounding of the activations, which the device applies first, is left out)."""

    def __init__(self, dims, weights):
        super().__init__(dims, weights)
        for k in list(self.w):
            if k.startswith("model.language_model.layers.") and k.endswith("_proj.weight"):
                self.w[k] = _mxq(self.w[k].to(torch.float16).float())

    def decoder_layer(self, x, l, pos0, cache):
        import torch.nn.functional as F
        d, w = self.dims, self.w
        p = f"model.language_model.layers.{l}."
        T = x.shape[0]
        y = _mxq(qwen3_ref.rms_norm(x, w[p + "input_layernorm.weight"], d.rms_eps))
        q = (y @ w[p + "self_attn.q_proj.weight"].T).view(T, d.heads, d.head_dim)
        k = (y @ w[p + "self_attn.k_proj.weight"].T).view(T, d.kv_heads, d.head_dim)
        v = (y @ w[p + "self_attn.v_proj.weight"].T).view(T, d.kv_heads, d.head_dim)
        q = qwen3_ref.rms_norm(q, w[p + "self_attn.q_norm.weight"], d.rms_eps)
        k = qwen3_ref.rms_norm(k, w[p + "self_attn.k_norm.weight"], d.rms_eps)
        pos = torch.arange(pos0, pos0 + T)
        q, k = self._rope(q, pos), self._rope(k, pos)
        if cache is not None:
            if cache[l] is not None:
                k, v = torch.cat([cache[l][0], k], 0), torch.cat([cache[l][1], v], 0)
            cache[l] = (k, v)
        g = d.heads // d.kv_heads
        kk, vv = k.repeat_interleave(g, dim=1), v.repeat_interleave(g, dim=1)
        s_ = torch.einsum("qhd,khd->hqk", q, kk) * d.head_dim ** -0.5
        s_ = s_.masked_fill((torch.arange(kk.shape[0])[None, :] > pos[:, None])[None], float("-inf"))
        a = torch.einsum("hqk,khd->qhd", torch.softmax(s_, -1), vv).reshape(T, d.heads * d.head_dim)
        x = x + _mxq(a) @ w[p + "self_attn.o_proj.weight"].T
        y = _mxq(qwen3_ref.rms_norm(x, w[p + "post_attention_layernorm.weight"], d.rms_eps))
        y = F.silu(y @ w[p + "mlp.gate_proj.weight"].T) * (y @ w[p + "mlp.up_proj.weight"].T)
        return x + _mxq(y) @ w[p + "mlp.down_proj.weight"].T


def test_float8w_decoder_matches_the_mx_rounding_oracle_and_states_its_distance_from_fp32(hip):
    """dtype "float8w": every decoder layer's q/k/v, o, gate/up and down projections run as MX-fp8 GEMMs (weights quantised at
    create, activations per GEMM).  The arithmetic itself is pinned at kernel level (tests/test_gpu_kernels.py: quantiser bytes
    identical to the MX specification, product equal to the exact product of the dequantised operands).  Here: against the oracle
    WITH the same MX rounding points the first tokens are equal and the prompt logits sit as far from it (measured 0.64 of a ~1.5
    spread) as two independent realisations of the format's noise do -- the device quantises fp16-rounded activations, the oracle
    fp32 ones, and 3 mantissa bits turn such differences into different roundings -- and against the plain fp32
    oracle, where the distance is the FORMAT's (~0.9 of a 1.5 logit spread on this toy model, measured with the oracle alone): reported, and bounded loosely, orders of magnitude outside the
    1e-3 parity bar.  That is the qualification BASELINE cfg5's "fp8 MFMA" gets here: a throughput type, not a parity type."""
    from whisperjav_amd import qwen
    d = qwen.Qwen3Dims(hidden=256, n_layer=3, n_head=4, n_kv_head=2, head_dim=128, ffn=640, vocab=4096, rope_theta=10000.0,
                       audio_token_id=9, eos_token_ids=(1, 2))
    w = qwen.synth_weights(d, seed=7)
    od = qwen3_ref.Qwen3AsrDims(d=d.hidden, layers=d.n_layer, heads=d.n_head, kv_heads=d.n_kv_head, head_dim=d.head_dim, ffn=d.ffn,
                                vocab=d.vocab, rope_theta=d.rope_theta, rms_eps=d.rms_eps, audio_token_id=d.audio_token_id,
                                eos_token_ids=d.eos_token_ids)
    plain, mxo = qwen3_ref.Qwen3AsrOracle(od, w), _Mx8Oracle(od, w)
    model = qwen.HipQwen3Decoder(d, w, dtype="float8w", max_seqs=4, max_ctx=256)
    prompts = _prompts(d, np.random.default_rng(3))
    embeds = [model.prompt_embeddings(ids, audio) for ids, audio in prompts]
    logits = model.prefill(embeds, want_logits=True).cpu()
    res = model.generate(max_new_tokens=8)
    again_single = []
    for b in range(len(prompts)):
        model.prefill([embeds[b]])
        again_single.append(model.generate(max_new_tokens=8).tokens[0])
    assert again_single == res.tokens                                   # deterministic, batch == single
    e_mx = e_fp32 = 0.0
    first_same = 0
    with torch.no_grad():
        for b, (ids, audio) in enumerate(prompts):
            x = plain.embed(ids, audio)
            ref_mx, ref = mxo.logits(x.clone())[-1], plain.logits(x.clone())[-1]
            e_mx = max(e_mx, float((logits[b] - ref_mx).abs().max()))
            e_fp32 = max(e_fp32, float((logits[b] - ref).abs().max()))
            first_same += int(res.tokens[b][0] == int(torch.log_softmax(ref_mx, -1).argmax()))
    import json, os
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/diag_qwen.jsonl", "a") as f:
        f.write(json.dumps({"test": "qwen_float8w", "max_logit_err_vs_mx_oracle": e_mx, "max_logit_err_vs_fp32_oracle": e_fp32,
                            "first_token_same_as_mx_oracle": first_same, "of": len(prompts)}) + "\n")
    assert e_mx < 1.2, e_mx
    assert e_fp32 < 1.5, e_fp32
    assert first_same >= 3
    model.close()


@pytest.mark.parametrize("dtype", ["float16", "bfloat16"])
def test_prompt_attention_on_the_matrix_cores_equals_the_row_kernel(hip, dtype):
    """Prompt passes of the 16-bit types run their causal attention as MFMA tiles (csrc/qwen.hip prompt_attn_kernel; wj_tune
    qwen_prompt_mfma).  Sequence lengths on both sides of every tile edge (1, 63 / 64 / 65, 128 / 129, 200, 300 = three query
    blocks): the last-position logits, a classification over interior rows, and the generation that continues from the KV
    cache agree with the one-row-per-wave kernel (fp32 softmax weights) to the rounding of the probabilities, and with the
    fp32 oracle to the type's bound."""
    from whisperjav_amd import hipbind, qwen
    d = qwen.Qwen3Dims(hidden=256, n_layer=3, n_head=4, n_kv_head=2, head_dim=128, ffn=640, vocab=4096, rope_theta=10000.0,
                       audio_token_id=9, eos_token_ids=(1, 2))
    w = qwen.synth_weights(d, seed=11)
    od = qwen3_ref.Qwen3AsrDims(d=d.hidden, layers=d.n_layer, heads=d.n_head, kv_heads=d.n_kv_head, head_dim=d.head_dim, ffn=d.ffn,
                                vocab=d.vocab, rope_theta=d.rope_theta, rms_eps=d.rms_eps, audio_token_id=d.audio_token_id,
                                eos_token_ids=d.eos_token_ids)
    oracle = qwen3_ref.Qwen3AsrOracle(od, w)
    lengths = (1, 17, 63, 64, 65, 128, 129, 200, 300)
    rng = np.random.default_rng(21)
    model = qwen.HipQwen3Decoder(d, w, dtype=dtype, max_seqs=len(lengths), max_ctx=384)
    embeds = []
    for n in lengths:
        ids = rng.integers(20, d.vocab, n).tolist()
        embeds.append(model.prompt_embeddings(ids, None))
    rows = [sorted({0, n // 2, n - 1}) for n in lengths]
    head_w = torch.from_numpy(rng.standard_normal((24, d.hidden)).astype(np.float32) * 0.2)
    out = {}
    try:
        for mode in (0, 1):
            hipbind.tune("qwen_prompt_mfma", mode)
            _, cl = model.classify(embeds, rows, head_w, want_logits=True)
            logits = model.prefill(embeds, want_logits=True).cpu()
            gen = model.generate(max_new_tokens=6)
            out[mode] = (logits, cl.cpu(), gen)
    finally:
        hipbind.tune("qwen_prompt_mfma", 1)
    (l0, c0, g0), (l1, c1, g1) = out[0], out[1]
    assert torch.isfinite(l1).all() and torch.isfinite(c1).all()
    scale = float(l0.abs().max())
    kern = float((l1 - l0).abs().max()) / scale
    kern_c = float((c1 - c0).abs().max()) / max(1.0, float(c0.abs().max()))
    with torch.no_grad():
        ref = torch.stack([oracle.logits(e.cpu())[-1] for e in embeds])
    e0 = float((l0 - ref).abs().max()) / scale
    e1 = float((l1 - ref).abs().max()) / scale
    print(f"{dtype}: tile kernel vs row kernel {kern:.2e} (logits) {kern_c:.2e} (classifier); vs fp32 oracle: row {e0:.2e} tile {e1:.2e}")
    bound = {"float16": 4e-3, "bfloat16": 3e-2}[dtype]
    assert kern < bound and kern_c < bound, (kern, kern_c)
    assert e1 < max(1.5 * e0, bound), (e0, e1)
    same = sum(a == b for a, b in zip(g0.tokens, g1.tokens))
    assert same >= len(lengths) - (1 if dtype == "float16" else 3), (g0.tokens, g1.tokens)
    model.close()


@pytest.mark.parametrize("dtype", ["float16", "bfloat16"])
def test_decode_batches_split_k_of_the_residual_projections(hip, dtype):
    """Decode batches of 65 .. max_seqs rows (16-bit types) cut o_proj / down_proj into K slices whose raw fp32 sums the next
    RMSNorm adds into the residual stream in slice order (wj_tune qwen_splitk, csrc/qwen.hip resid_gemm): more workgroups for
    the 2048-column projections, the same arithmetic up to the fp32 summation order.  96 clips generated to EOS with the
    slices on (default 4; the toy ffn 640 admits 2 without split activations) and off: same tokens, log-probs within fp32
    noise of each other, same lengths; three of the sequences checked against the fp32 oracle."""
    from whisperjav_amd import hipbind
    n = 96
    d, ramp, w, oracle, model = _ramp_setup(dtype, seed=5, max_seqs=n, max_ctx=192)
    rng = np.random.default_rng(31)
    clips = [(24 + (7 * i) % 60, 3 + i % 4) for i in range(n)]
    prompts = _ramp_prompts(d, ramp, rng, clips)
    embeds = [model.prompt_embeddings(ids, audio) for ids, audio in prompts]
    ids = [p[0] for p in prompts]
    out = {}
    try:
        for ks in (1, 4):
            hipbind.tune("qwen_splitk", ks)
            model.prefill(embeds)
            out[ks] = model.generate(max_new_tokens=100, repetition_penalty=1.1, prompt_ids=ids)
    finally:
        hipbind.tune("qwen_splitk", 4)
    a, b = out[1], out[4]
    assert a.steps < 100 and abs(b.steps - a.steps) <= 8, (a.steps, b.steps)
    same = sum(x == y for x, y in zip(a.tokens, b.tokens))
    # a near-tie may flip under another fp32 summation order: one ulp of the residual stream moves a 16-bit rounding of the normed
    # activations, i.e. 5e-4 (float16) / 4e-3 (bfloat16) of a logit; over ~4000 generated tokens a few bfloat16 decisions sit that close
    assert same >= (n - 2 if dtype == "float16" else int(0.9 * n)), same
    worst = max(float(np.abs(np.array(x) - np.array(y)).max()) for x, y, p, q in zip(a.token_logprob, b.token_logprob, a.tokens, b.tokens)
                if p == q)
    print(f"{dtype}: {same}/{n} sequences identical, log-prob difference {worst:.2e}, lengths {min(map(len, b.tokens))}..{max(map(len, b.tokens))}")
    assert worst < {"float16": 2e-3, "bfloat16": 6e-2}[dtype], worst       # measured 1.1e-3 / 3.0e-2: one 16-bit ulp of a normed activation
    assert len({len(t) for t in b.tokens}) > 3
    agree = 0
    with torch.no_grad():
        for i in (0, 37, 95):
            toks, _ = oracle.greedy(prompts[i][0], prompts[i][1], 100, repetition_penalty=1.1)
            agree += toks == b.tokens[i]
    assert agree >= (2 if dtype == "float16" else 1), agree
    model.close()
