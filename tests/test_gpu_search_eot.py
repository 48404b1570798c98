"""Searches that END, on a real MI355X through the C ABI.

The plain synthetic weights never emit the end-of-text token, so in tests/test_gpu_pipeline.py every hypothesis runs to
``max_new_tokens``.  Here the weights are ``weights.SPEECHLIKE`` (end-of-text ramp, duration cue, peaked cross-attention)
and the windows are clips of 0.8 .. 6 s of synthetic speech (log-mel by the HIP extractor): windows -- and the beams of
one window -- finish at different steps, after a number of tokens that grows with the clip.  Against the oracle (``oracle/decoding.py``: ``greedy_decode``, the literal
CTranslate2 ``beam_search`` with its branch trace) and against the host-driven search over the step API:

  (a) hypotheses of different lengths, ranked by ``cum / len ** length_penalty``;
  (b) windows stopping on ``len(finished) >= round(beam * patience)`` before the length limit;
  (c) finished slots re-filled from the candidates ``beam .. 2 * beam``;
  (d) a 16-window batch with ragged finish == per-window decodes;
  (e) the greedy loop's early exit (``csrc/engine.hip`` polls the finished flags every 16 steps) and the beam loop's
      (done counter polled every 8 steps).

Every test asserts that these branches actually ran (oracle trace / realised lengths / ``last_decode_info()["steps"]``).
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import decoding
from tests import helpers

pytestmark = pytest.mark.gpu

DIAG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SMALL = dict(n_mels=80, d_model=128, heads=2, layers=2, n_vocab=51865)


def _diag(name, payload):
    os.makedirs(DIAG, exist_ok=True)
    with open(os.path.join(DIAG, "diag_search_eot.jsonl"), "a") as f:
        f.write(json.dumps({"test": name, **payload}) + "\n")


CLIP_SECONDS = (0.8, 6.0, 2.5, 4.0, 1.5, 5.0, 3.0, 3.5)


def _setup(dtype, n_windows, max_beam, seed=33, clip_seed=19):
    from whisperjav_amd import engine, synth, weights as pweights
    d = helpers.small_dims(**SMALL)
    oracle, w = helpers.make_oracle(d, seed=seed, emulate=dtype, **pweights.SPEECHLIKE)
    model = engine.HipWhisper(d, w, dtype=dtype, max_batch=n_windows, max_beam=max_beam)
    clips = [synth.speech_like(CLIP_SECONDS[i % len(CLIP_SECONDS)] + 0.1 * (i // len(CLIP_SECONDS)), seed=clip_seed + i)
             for i in range(n_windows)]
    mel = engine.HipLogMel(d.n_mels, "fw")(clips)          # the product's own features feed both sides
    model.encode(mel)
    with torch.no_grad():
        xa = oracle.encode(mel.cpu())
    return d, oracle, model, xa


def _common(a, b):
    return next((i for i, (x, y) in enumerate(zip(a, b)) if x != y), min(len(a), len(b)))


@pytest.mark.parametrize("dtype", ["float32", "float16", "bfloat16"])
def test_device_greedy_ends_with_eot(hip, dtype):
    """(e) + ragged rows: every row stops at its own EOT, the cumulative log-prob includes the EOT token, rows that are
    done are padded with EOT and not advanced, the loop leaves before ``max_new_tokens``."""
    from whisperjav_amd import engine
    B, max_new = 6, 64
    d, oracle, model, xa = _setup(dtype, B, 1)
    prompt = model.sot_prompt("ja", "transcribe")
    suppress = (1, 2, 7, 8, 9, 10, 14, 25, 50256, 50360, 50361)
    res = model.decode_greedy(np.tile(np.array(prompt, dtype=np.int32), (B, 1)),
                              engine.DecodeOptions(max_new_tokens=max_new, suppress_tokens=suppress, max_initial_timestamp=1.0))
    info = model.last_decode_info()
    cfg = decoding.FilterConfig(suppress_tokens=suppress, max_initial_timestamp_index=50)
    ref = decoding.greedy_decode(oracle, xa, prompt, max_new, cfg)
    ref_len = [len(t) for t in ref.tokens]
    assert max(ref_len) < max_new - 16 and len(set(ref_len)) > 1, ref_len         # the oracle's rows end, raggedly
    worst, same = 0.0, 0
    for r in range(B):
        n = int(res.n_tokens[r])
        got = res.tokens[r, :n].tolist()
        assert model.tokens.eot not in got
        assert (res.tokens[r, n:] == model.tokens.eot).all()                        # padded, not advanced
        c = _common(got, ref.tokens[r])
        same += got == ref.tokens[r]
        lp_n = min(c + 1, len(ref.token_logprob[r])) if got == ref.tokens[r] else c  # + the EOT token's own log-prob
        worst = max(worst, float(np.abs(res.token_logprob[r, :lp_n] - np.array(ref.token_logprob[r][:lp_n])).max()) if lp_n else 0.0)
        if dtype == "float32":
            assert got == ref.tokens[r], (r, got, ref.tokens[r])
            assert abs(float(res.sum_logprob[r]) - float(ref.sum_logprob[r])) < 1e-3 * (n + 1)
            assert abs(float(res.sum_logprob[r]) - float(res.token_logprob[r, : n + 1].sum())) < 1e-4   # EOT included
        else:
            assert c >= min(4, ref_len[r]), (r, got, ref.tokens[r])
    _diag("greedy_eot", {"dtype": dtype, "ref_len": ref_len, "got_len": res.n_tokens.tolist(), "rows_identical": same,
                         "max_logprob_diff": worst, "steps": info["steps"]})
    assert worst < {"float32": 1e-3, "float16": 0.02, "bfloat16": 0.15}[dtype], worst
    assert info["steps"] < max_new and info["steps"] >= int(res.n_tokens.max()) + 1, info        # (e) early exit
    assert len(set(res.n_tokens.tolist())) > 1
    model.close()


@pytest.mark.parametrize("dtype", ["float32", "float16", "bfloat16"])
@pytest.mark.parametrize("beam,patience,lpen,rep,ngram,max_new", [
    (2, 1.2, 1.0, 1.5, 3, 64),     # the reference's "balanced" defaults
    (5, 1.2, 1.0, 1.5, 3, 64),     # BASELINE cfg3
    (3, 1.0, 0.0, 1.0, 0, 64),     # allow-early-exit shape of CTranslate2 (patience 1, no length penalty)
    (4, 2.0, 1.0, 1.3, 2, 64),
    (5, 1.2, 1.0, 1.5, 3, 14),     # some windows stop on patience, others at the length limit
    (8, 1.0, 1.0, 1.0, 0, 40),
])
def test_device_beam_search_ends_with_eot(hip, dtype, beam, patience, lpen, rep, ngram, max_new):
    """(a) (b) (c) + ragged windows, device loop == host-driven search (same engine numerics, every compute type) ==
    the oracle (float32: exact hypotheses; 16-bit: winner's cumulative log-prob within the type's bound)."""
    from whisperjav_amd import engine, search
    B = 6
    d, oracle, model, xa = _setup(dtype, B, beam)
    toks = model.tokens
    prompt = model.sot_prompt("ja", "transcribe")
    suppress = (toks.sot, toks.translate, toks.transcribe, toks.sot_lm, toks.sot_prev, toks.no_speech)
    dopt = engine.DecodeOptions(max_new_tokens=max_new, suppress_tokens=suppress, max_initial_timestamp=0.0,
                                repetition_penalty=rep, no_repeat_ngram_size=ngram)
    res = model.decode_beam(np.tile(np.array(prompt, dtype=np.int32), (B, 1)), dopt, beam_size=beam, patience=patience,
                            length_penalty=lpen)
    info = model.last_decode_info()
    assert info["hip_graph"]
    opts = search.SearchOptions(beam_size=beam, patience=patience, length_penalty=lpen, repetition_penalty=rep,
                                no_repeat_ngram_size=ngram, suppress_tokens=suppress, max_initial_timestamp_index=0,
                                max_new_tokens=max_new)
    host = search.beam_search(search.HipStepScorer(model, opts), [prompt] * B, opts, eot=toks.eot,
                              timestamp_begin=toks.timestamp_begin)
    fcfg = decoding.FilterConfig(suppress_tokens=suppress, max_initial_timestamp_index=0)
    bcfg = decoding.BeamConfig(beam, patience, lpen, rep, ngram, max_new)
    stops, steps, refills, lens, worst, same = set(), [], 0, set(), 0.0, 0
    for w in range(B):
        got = res.tokens[w, : res.n_tokens[w]].tolist()
        assert toks.eot not in got and (res.tokens[w, res.n_tokens[w]:] == toks.eot).all()
        assert got == host[w].sequences[0], (w, got, host[w].sequences[0])
        assert abs(float(res.sum_logprob[w]) - host[w].cum_logprobs[0]) < 1e-3
        tr = {}
        ref, nsp = decoding.beam_search(oracle, xa[w:w + 1], prompt, bcfg, fcfg, trace=tr)
        stops.add(tr["stop"]); steps.append(tr["steps"]); refills += tr["refills"]
        lens.update(len(t) for t, _, _ in ref)
        same += got == ref[0][0]
        if dtype == "float32":
            assert got == ref[0][0], (w, got, ref[0][0])
            assert abs(float(res.sum_logprob[w]) - ref[0][2]) < 1e-3
            assert abs(float(res.token_logprob[w, 0]) - ref[0][1]) < 1e-3           # normalised score
            assert abs(float(res.no_speech_prob[w]) - nsp) < 1e-5
        if got == ref[0][0]:
            worst = max(worst, abs(float(res.sum_logprob[w]) - ref[0][2]))
    _diag("device_beam_eot", {"dtype": dtype, "beam": beam, "patience": patience, "max_new": max_new, "stops": sorted(stops),
                              "oracle_steps": steps, "refills": refills, "lens": sorted(lens), "winners_identical": same,
                              "cum_logprob_diff": worst, "device_steps": info["steps"]})
    # the oracle ran the branches this test is about (so the equalities above cover them)
    assert min(lens) < max_new and len(lens) > 1, lens                              # (a)
    assert "patience" in stops and refills > 0, (stops, refills)                    # (b) (c)
    if max_new == 14:
        assert stops == {"patience", "length"}, stops
    else:
        assert len(set(steps)) > 1, steps                                           # ragged finish inside the batch
        assert info["steps"] < max_new, info                                        # (e) the done counter ended the loop
    assert info["steps"] >= max(steps)
    if dtype != "float32":
        assert same >= B - 2 and worst < {"float16": 0.05, "bfloat16": 0.3}[dtype], (same, worst)    # cumulative, up to 25 tokens
    model.close()


@pytest.mark.parametrize("dtype", ["float32", "float16"])
@pytest.mark.parametrize("beam,patience,lpen,max_new", [
    (2, 1.2, None, 64),      # the reference's fidelity defaults (config/components/asr/openai_whisper.py:229-255)
    (5, 2.0, None, 64),
    (3, 1.0, 1.0, 64),       # GNMT length penalty
    (5, 1.0, None, 15),      # length limit: windows holding fewer than `beam` finished sequences are topped up with live beams
    (8, 1.0, None, 40),
])
def test_device_openai_beam_search_ends_with_eot(hip, dtype, beam, patience, lpen, max_new):
    """Fidelity mode's search on the device (``wj_whisper_decode_beam_openai``: whisper's ``BeamSearchDecoder`` +
    ``MaximumLikelihoodRanker``) == the host-driven restatement over the step API (same engine numerics, every compute
    type) == ``oracle.decoding.beam_search_openai`` (float32: exact), on searches that end at different lengths."""
    from whisperjav_amd import engine, search
    B = 6
    d, oracle, model, xa = _setup(dtype, B, beam)
    toks = model.tokens
    prompt = model.sot_prompt("ja", "transcribe")
    suppress = (toks.sot, toks.translate, toks.transcribe, toks.sot_lm, toks.sot_prev, toks.no_speech)
    dopt = engine.DecodeOptions(max_new_tokens=max_new, suppress_tokens=suppress, max_initial_timestamp=1.0)
    P = np.tile(np.array(prompt, dtype=np.int32), (B, 1))
    res = model.decode_beam(P, dopt, beam_size=beam, patience=patience, length_penalty=lpen, flavor="openai")
    info = model.last_decode_info()
    assert info["hip_graph"]
    sub = model.decode_beam(P[:2], dopt, beam_size=beam, patience=patience, length_penalty=lpen, flavor="openai", slots=[4, 1])
    for row, w in enumerate([4, 1]):
        assert sub.tokens[row, : sub.n_tokens[row]].tolist() == res.tokens[w, : res.n_tokens[w]].tolist()
    opts = search.SearchOptions(beam_size=beam, patience=patience, length_penalty=-1 if lpen is None else lpen,
                                suppress_tokens=suppress, max_initial_timestamp_index=50, max_new_tokens=max_new)
    host = search.beam_search_openai(search.HipStepScorer(model, opts), [prompt] * B, opts, eot=toks.eot,
                                     timestamp_begin=toks.timestamp_begin)
    fcfg = decoding.FilterConfig(suppress_tokens=suppress, max_initial_timestamp_index=50)
    lens, same, worst = set(), 0, 0.0
    for w in range(B):
        got = res.tokens[w, : res.n_tokens[w]].tolist()
        assert toks.eot not in got
        assert got == host[w].sequences[0], (w, got, host[w].sequences[0])
        assert abs(float(res.sum_logprob[w]) - host[w].cum_logprobs[0]) < 1e-3
        seq, total, avg, nsp = decoding.beam_search_openai(oracle, xa[w:w + 1], prompt, beam, patience, lpen, max_new, fcfg)
        lens.add(len(seq))
        same += got == seq
        if dtype == "float32":
            assert got == seq, (w, got, seq)
            assert abs(float(res.sum_logprob[w]) - total) < 1e-3
            assert abs(float(res.sum_logprob[w]) / (len(got) + 1) - avg) < 1e-4
            assert abs(float(res.no_speech_prob[w]) - nsp) < 1e-5
        if got == seq:
            worst = max(worst, abs(float(res.sum_logprob[w]) - total))
    _diag("device_beam_openai", {"dtype": dtype, "beam": beam, "patience": patience, "max_new": max_new, "lens": sorted(lens),
                                 "winners_identical": same, "cum_logprob_diff": worst, "device_steps": info["steps"]})
    assert len(lens) > 1, lens
    if max_new >= 40:
        assert max(lens) < max_new and info["steps"] < max_new, (lens, info)
    if dtype != "float32":
        assert same >= B - 2 and worst < 0.05, (same, worst)
    model.close()


def test_ragged_16_window_batch_equals_per_window_decodes(hip):
    """(d): 16 windows decoded as ONE batch (windows finish at different steps, the rest keep running) give the same
    hypotheses as 16 single-window calls through the slot map, and as the oracle window by window."""
    from whisperjav_amd import engine
    B, beam, max_new = 16, 5, 64
    d, oracle, model, xa = _setup("float32", B, beam, seed=7, clip_seed=3)
    toks = model.tokens
    prompt = model.sot_prompt("ja", "transcribe")
    suppress = (toks.sot, toks.translate, toks.transcribe, toks.sot_lm, toks.sot_prev, toks.no_speech)
    dopt = engine.DecodeOptions(max_new_tokens=max_new, suppress_tokens=suppress, max_initial_timestamp=0.0,
                                repetition_penalty=1.5, no_repeat_ngram_size=3)
    P = np.array(prompt, dtype=np.int32)
    res = model.decode_beam(np.tile(P, (B, 1)), dopt, beam_size=beam, patience=1.2, length_penalty=1.0)
    batch_steps = model.last_decode_info()["steps"]
    gres = model.decode_greedy(np.tile(P, (B, 1)), dopt)
    fcfg = decoding.FilterConfig(suppress_tokens=suppress, max_initial_timestamp_index=0)
    bcfg = decoding.BeamConfig(beam, 1.2, 1.0, 1.5, 3, max_new)
    gref = decoding.greedy_decode(oracle, xa, prompt, max_new, fcfg, processors=bcfg)
    steps = []
    for w in range(B):
        one = model.decode_beam(P[None], dopt, beam_size=beam, patience=1.2, length_penalty=1.0, slots=[w])
        steps.append(model.last_decode_info()["steps"])
        got = res.tokens[w, : res.n_tokens[w]].tolist()
        assert got == one.tokens[0, : one.n_tokens[0]].tolist(), w
        assert abs(float(res.sum_logprob[w]) - float(one.sum_logprob[0])) < 1e-4
        ref, _ = decoding.beam_search(oracle, xa[w:w + 1], prompt, bcfg, fcfg)
        assert got == ref[0][0], (w, got, ref[0][0])
        assert abs(float(res.sum_logprob[w]) - ref[0][2]) < 1e-3
        assert gres.tokens[w, : gres.n_tokens[w]].tolist() == gref.tokens[w], w
    lens = res.n_tokens.tolist()
    _diag("ragged16", {"beam_len": lens, "greedy_len": gres.n_tokens.tolist(), "batch_steps": batch_steps, "single_steps": steps})
    assert len(set(lens)) >= 3 and max(lens) < max_new, lens
    assert len(set(steps)) >= 2 and batch_steps == max(steps), (batch_steps, steps)     # both polled every 8 steps
    model.close()


@pytest.mark.parametrize("dtype", ["float32", "float16"])
def test_batch_compaction_is_transparent(hip, dtype):
    """The device beam search re-packs its batch when windows finish (``wj_tune beam_compact``): forced here on every
    poll (``beam_poll=1, beam_compact_min=1, beam_compact_pct=1``), with a slot map, against the un-compacted loop and the
    oracle.  KV-cache rows and finished lists never move, so the hypotheses must be the same ones."""
    from whisperjav_amd import engine, hipbind
    B, beam, max_new = 8, 5, 64
    d, oracle, model, xa = _setup(dtype, B, beam, seed=11, clip_seed=5)
    toks = model.tokens
    prompt = model.sot_prompt("ja", "transcribe")
    suppress = (toks.sot, toks.translate, toks.transcribe, toks.sot_lm, toks.sot_prev, toks.no_speech)
    dopt = engine.DecodeOptions(max_new_tokens=max_new, suppress_tokens=suppress, max_initial_timestamp=0.0,
                                repetition_penalty=1.5, no_repeat_ngram_size=3)
    P = np.tile(np.array(prompt, dtype=np.int32), (B, 1))
    order = [5, 0, 7, 2, 1, 6, 3, 4]                                     # windows addressed through the slot map too
    try:
        hipbind.tune("beam_compact", 0)
        plain = model.decode_beam(P, dopt, beam_size=beam, patience=1.2, length_penalty=1.0)
        assert model.last_decode_info()["compactions"] == 0
        for k, v in (("beam_compact", 1), ("beam_poll", 1), ("beam_compact_min", 1), ("beam_compact_pct", 1)):
            hipbind.tune(k, v)
        packed = model.decode_beam(P, dopt, beam_size=beam, patience=1.2, length_penalty=1.0)
        info = model.last_decode_info()
        shuffled = model.decode_beam(P, dopt, beam_size=beam, patience=1.2, length_penalty=1.0, slots=order)
    finally:
        for k, v in (("beam_compact", 1), ("beam_poll", 4), ("beam_compact_min", 8), ("beam_compact_pct", 12)):
            hipbind.tune(k, v)
    assert info["compactions"] >= 3 and info["hip_graph"], info
    assert info["window_steps"] < info["steps"] * B                      # less work than the un-compacted loop
    fcfg = decoding.FilterConfig(suppress_tokens=suppress, max_initial_timestamp_index=0)
    bcfg = decoding.BeamConfig(beam, 1.2, 1.0, 1.5, 3, max_new)
    tol = 1e-4 if dtype == "float32" else 2e-2      # smaller batches pick other GEMM kernels: summation order differs
    for w in range(B):
        a = plain.tokens[w, : plain.n_tokens[w]].tolist()
        b = packed.tokens[w, : packed.n_tokens[w]].tolist()
        c = shuffled.tokens[order.index(w), : shuffled.n_tokens[order.index(w)]].tolist()
        if dtype == "float32":
            ref, nsp = decoding.beam_search(oracle, xa[w:w + 1], prompt, bcfg, fcfg)
            assert a == b == c == ref[0][0], (w, a, b, c, ref[0][0])
            assert abs(float(packed.sum_logprob[w]) - ref[0][2]) < 1e-3
            assert abs(float(packed.no_speech_prob[w]) - nsp) < 1e-5
        else:
            assert a == b == c, (w, a, b, c)
        assert abs(float(plain.sum_logprob[w]) - float(packed.sum_logprob[w])) < tol
        assert abs(float(shuffled.sum_logprob[order.index(w)]) - float(packed.sum_logprob[w])) < tol
    _diag("compaction", {"dtype": dtype, "lens": packed.n_tokens.tolist(), **info})
    model.close()


@pytest.mark.parametrize("dtype", ["float32", "float16"])
def test_beam_winner_per_token_logprobs(hip, dtype):
    """``decode_beam(token_logprobs=True)`` (wj_tune beam_token_logprobs): the search carries the cumulative log-prob of every
    hypothesis through its history gathers; the winner's per-token values equal the oracle's (its beam search records the
    same differences in ``trace["token_logprobs"]``) -- 1e-4 in float32, the type's per-token bound in float16 on this toy
    model -- sum to ``sum_logprob``, and the winner itself is the one the plain search returns.  CTranslate2 rules stopped by
    patience and by the length limit (where the last token is appended and nothing is added for an EOT), and the
    openai-whisper rules including the top-up from live beams."""
    from whisperjav_amd import engine, hipbind
    B = 6
    d, oracle, model, xa = _setup(dtype, B, 5)
    toks = model.tokens
    prompt = model.sot_prompt("ja", "transcribe")
    P = np.tile(np.array(prompt, dtype=np.int32), (B, 1))
    suppress = (toks.sot, toks.translate, toks.transcribe, toks.sot_lm, toks.sot_prev, toks.no_speech)
    fcfg = decoding.FilterConfig(suppress_tokens=suppress, max_initial_timestamp_index=0)
    tol = {"float32": 1e-4, "float16": 8e-3}[dtype]      # d_model 128 toy: the EOT step carries the largest error (PARITY.md); the 1e-3 bar is asserted at the large-v3 geometry
    seen_length_stop = seen_eot = 0
    for max_new in (64, 14):
        dopt = engine.DecodeOptions(max_new_tokens=max_new, suppress_tokens=suppress, max_initial_timestamp=0.0,
                                    repetition_penalty=1.5, no_repeat_ngram_size=3)
        plain = model.decode_beam(P, dopt, beam_size=5, patience=1.2, length_penalty=1.0)
        res = model.decode_beam(P, dopt, beam_size=5, patience=1.2, length_penalty=1.0, token_logprobs=True)
        assert model.last_decode_info()["compactions"] == 0
        assert np.array_equal(plain.tokens, res.tokens) and np.array_equal(plain.sum_logprob, res.sum_logprob)
        assert res.token_logprob.shape == (B, max_new + 1)
        bcfg = decoding.BeamConfig(5, 1.2, 1.0, 1.5, 3, max_new)
        for w in range(B):
            n = int(res.n_tokens[w])
            lp = res.token_logprob[w]
            assert np.isfinite(lp[: n + 1]).all() and np.isnan(lp[n + 1:]).all()
            assert abs(float(lp[: n + 1].sum()) - float(res.sum_logprob[w])) < 1e-4 * (n + 1)
            tr = {}
            ref, _ = decoding.beam_search(oracle, xa[w:w + 1], prompt, bcfg, fcfg, trace=tr)
            if res.tokens[w, :n].tolist() != ref[0][0]:
                assert dtype != "float32"
                continue
            want = np.array(tr["token_logprobs"][0])
            assert len(want) == n + 1
            assert np.abs(lp[: n + 1] - want).max() < tol, (w, lp[: n + 1], want)
            if n == max_new:
                seen_length_stop += 1
                assert lp[n] == 0.0                     # ended by the length limit: no EOT was scored
            else:
                seen_eot += 1
                assert lp[n] < 0.0
    assert seen_length_stop and seen_eot, (seen_length_stop, seen_eot)
    # openai-whisper rules (no processors), a length limit short enough that windows are topped up with live beams
    for max_new in (64, 15):
        dopt = engine.DecodeOptions(max_new_tokens=max_new, suppress_tokens=suppress, max_initial_timestamp=0.0)
        plain = model.decode_beam(P, dopt, beam_size=5, patience=1.0, length_penalty=None, flavor="openai")
        res = model.decode_beam(P, dopt, beam_size=5, patience=1.0, length_penalty=None, flavor="openai", token_logprobs=True)
        assert np.array_equal(plain.tokens, res.tokens) and np.array_equal(plain.sum_logprob, res.sum_logprob)
        for w in range(B):
            n = int(res.n_tokens[w])
            lp = res.token_logprob[w]
            assert np.isfinite(lp[: min(n + 1, max_new + 1)]).all()
            assert abs(float(np.nansum(lp)) - float(res.sum_logprob[w])) < 1e-4 * (n + 1), (w, lp, res.sum_logprob[w])
            assert (lp[:n] <= 1e-6).all()
    with pytest.raises(hipbind.WjError, match="beam_token_logprobs"):
        model.decode_beam(P, dopt, beam_size=5)                                          # a plain search ...
        out = np.empty((B, dopt.max_new_tokens + 1), dtype=np.float32)
        import ctypes as C
        hipbind.check(model._lib.wj_whisper_last_beam_token_logprobs(model.handle, B, dopt.max_new_tokens + 1,
                                                                    out.ctypes.data_as(C.POINTER(C.c_float))))   # ... carries none
    model.close()


def test_short_kv_cache_and_encoder_slices_change_nothing(hip):
    """``HipWhisper(kv_len=..., enc_batch=...)``: a KV cache sized for prompt + max_new_tokens and an encoder run in slices
    give bit-identical encoder outputs and the same hypotheses as the full-size engine; a decode that does not fit the
    cache is refused."""
    from whisperjav_amd import engine, hipbind, synth, weights as pweights
    d = helpers.small_dims(**SMALL)
    w = pweights.synth_weights(d, seed=33, **pweights.SPEECHLIKE)
    clips = [synth.speech_like(CLIP_SECONDS[i], seed=19 + i) for i in range(7)]
    mel = engine.HipLogMel(d.n_mels, "fw")(clips)
    full = engine.HipWhisper(d, w, dtype="float16", max_batch=7, max_beam=5)
    small = engine.HipWhisper(d, w, dtype="float16", max_batch=7, max_beam=5, kv_len=3 + 40, enc_batch=3)
    assert small.kv_len == 48 and small.enc_batch == 3 and small.workspace_bytes < full.workspace_bytes
    e1 = full.encode(mel, want_output=True)
    e2 = small.encode(mel, want_output=True)                    # slices of 3 + 3 + 1 windows
    assert torch.equal(e1, e2)
    toks = full.tokens
    prompt = np.tile(np.array(full.sot_prompt("ja", "transcribe"), dtype=np.int32), (7, 1))
    suppress = (toks.sot, toks.translate, toks.transcribe, toks.sot_lm, toks.sot_prev, toks.no_speech)
    dopt = engine.DecodeOptions(max_new_tokens=40, suppress_tokens=suppress, max_initial_timestamp=0.0, repetition_penalty=1.5,
                                no_repeat_ngram_size=3)
    a = full.decode_beam(prompt, dopt, beam_size=5, patience=1.2)
    b = small.decode_beam(prompt, dopt, beam_size=5, patience=1.2)
    assert np.array_equal(a.tokens, b.tokens) and np.array_equal(a.n_tokens, b.n_tokens)
    assert np.array_equal(a.sum_logprob, b.sum_logprob)
    ga, gb = full.decode_greedy(prompt, dopt), small.decode_greedy(prompt, dopt)
    assert np.array_equal(ga.tokens, gb.tokens) and np.array_equal(ga.token_logprob, gb.token_logprob)
    with pytest.raises(hipbind.WjError, match="KV cache"):
        small.decode_beam(prompt, engine.DecodeOptions(max_new_tokens=60), beam_size=5)
    full.close(); small.close()


@pytest.mark.parametrize("dtype", ["float32", "float16"])
def test_golden_large_v3_r3_searches_that_end(hip, dtype):
    """Large-v3 geometry, fp16-representable ``SPEECHLIKE`` weights, two windows (a 6 s and a 2.5 s clip): greedy until
    EOT and the cfg3 beam search until patience stops it, against the committed oracle vectors
    (tests/golden/make_golden.py --large-r3).  Bars: tokens / winning hypothesis identical, per-token log-probs within
    1e-3 (the north-star bar) in both compute types -- for the greedy rows AND for the beam winner (its per-token values
    come from the cumulative histories the search carries under wj_tune beam_token_logprobs)."""
    from whisperjav_amd import dims as pdims, engine, synth, weights as pweights
    g = np.load(os.path.join(GOLDEN, "golden_large_v3_r3_eot.npz"))
    dims = pdims.dims_for("large-v3")
    clips = [synth.speech_like(float(s), seed=int(k)) for s, k in zip(g["clip_seconds"], g["clip_seeds"])]
    mel = engine.HipLogMel(128, "fw")(clips)
    assert str(g["weights"]) == "SPEECHLIKE"
    w = helpers.cached_weights(dims, int(g["seed"]), str(g["exact"]), pweights.SPEECHLIKE["eot"], pweights.SPEECHLIKE["cross_gain"],
                               pweights.SPEECHLIKE["logit_std"])
    model = engine.HipWhisper(dims, w, dtype=dtype, max_batch=2, max_beam=5)
    del w
    enc = model.encode(mel, want_output=True).cpu()
    d_probe = float(np.abs(enc[0][g["probe_t"]][:, g["probe_d"]].numpy() - g["enc_probe"]).max())
    sup = tuple(int(t) for t in g["suppress"])
    prompts = np.tile(np.array(g["prompt"], dtype=np.int32), (2, 1))
    beam, patience, lpen, rep, ngram, max_new = (float(x) for x in g["beam"])
    max_new = int(max_new)
    res = model.decode_greedy(prompts, engine.DecodeOptions(max_new_tokens=max_new, suppress_tokens=sup, max_initial_timestamp=1.0))
    g_steps = model.last_decode_info()["steps"]
    br = model.decode_beam(prompts, engine.DecodeOptions(max_new_tokens=max_new, suppress_tokens=sup, max_initial_timestamp=0.0,
                                                        repetition_penalty=rep, no_repeat_ngram_size=int(ngram)),
                           beam_size=int(beam), patience=patience, length_penalty=lpen)
    b_steps = model.last_decode_info()["steps"]
    # the same search carrying the cumulative score of every hypothesis (wj_tune beam_token_logprobs): same winner, and its
    # per-token log-probs are read back
    bl = model.decode_beam(prompts, engine.DecodeOptions(max_new_tokens=max_new, suppress_tokens=sup, max_initial_timestamp=0.0,
                                                        repetition_penalty=rep, no_repeat_ngram_size=int(ngram)),
                           beam_size=int(beam), patience=patience, length_penalty=lpen, token_logprobs=True)
    assert np.array_equal(bl.tokens, br.tokens) and np.array_equal(bl.n_tokens, br.n_tokens)
    assert np.array_equal(bl.sum_logprob, br.sum_logprob) and np.array_equal(bl.beam_score, br.beam_score)
    model.close()
    bar = 1e-3
    out = {"dtype": dtype, "probe_max_abs": d_probe, "greedy_steps": g_steps, "beam_steps": b_steps}
    ok = True
    for b in range(2):
        ref_t = g[f"greedy{b}_tokens"].tolist()
        ref_lp = g[f"greedy{b}_logprob"]
        n = int(res.n_tokens[b])
        got = res.tokens[b, :n].tolist()
        c = _common(got, ref_t)
        lp_n = min(c + 1, len(ref_lp)) if got == ref_t else c
        d_lp = float(np.abs(res.token_logprob[b, :lp_n] - ref_lp[:lp_n]).max()) if lp_n else 0.0
        b_got = br.tokens[b, : int(br.n_tokens[b])].tolist()
        b_ref = g[f"beam{b}_tokens"][0, : int(g[f"beam{b}_len"][0])].tolist()
        d_cum = abs(float(br.sum_logprob[b]) - float(g[f"beam{b}_cum"][0]))
        ref_blp = g[f"beam{b}_token_logprobs"]                   # the winner's tokens, then the EOT
        got_blp = bl.token_logprob[b, : int(bl.n_tokens[b]) + 1]
        assert np.isnan(bl.token_logprob[b, int(bl.n_tokens[b]) + 1:]).all()
        assert abs(float(got_blp.sum()) - float(bl.sum_logprob[b])) < 1e-4 * max(1, len(got_blp))
        d_blp = float(np.abs(got_blp - ref_blp).max()) if b_got == b_ref else float("inf")
        norm = g[f"beam{b}_norm"]
        out.update({f"w{b}_greedy_len": len(ref_t), f"w{b}_greedy_common": c, f"w{b}_greedy_lp_max_abs": d_lp,
                    f"w{b}_beam_same": b_got == b_ref, f"w{b}_beam_len": len(b_ref), f"w{b}_beam_cum_abs": d_cum, f"w{b}_beam_token_lp_max_abs": d_blp,
                    f"w{b}_beam_lens": g[f"beam{b}_len"].tolist(), f"w{b}_beam_margin": float(norm[0] - norm[1]),
                    f"w{b}_oracle_stop": str(g[f"beam{b}_stop"]), f"w{b}_oracle_refills": int(g[f"beam{b}_refills"])})
        ok = ok and got == ref_t and d_lp < bar and b_got == b_ref and d_cum < bar * max(1, len(b_ref)) and d_blp < bar
        assert len(ref_t) < max_new and str(g[f"beam{b}_stop"]) == "patience" and int(g[f"beam{b}_refills"]) > 0
        assert len(set(g[f"beam{b}_len"].tolist())) > 1
    _diag("golden_large_v3_r3", out)
    assert d_probe < (5e-3 if dtype == "float32" else 0.05), d_probe
    assert ok, out
    assert g_steps < max_new and b_steps < max_new, (g_steps, b_steps)
