"""Parity of the HIP hot path against the CPU oracle on a real MI355X (through the C ABI).

Bars (BASELINE.json north_star):
  * log-mel: frame counts / indices exact; values within 5e-5 of the oracle's fp32 restatement (measured 2.1e-5)
    (the two differ only in fp32 summation order; float64 yardstick in tests/test_oracle_logmel.py);
  * compute_type float32: greedy token ids identical and per-token log-probs within 1e-3 of the fp32
    oracle; encoder activations within 2e-3;
  * compute_type bfloat16: NOT a parity type (README "Compute types"): checked against the oracle run with the engine's
    bf16 rounding points, tolerance stated per assertion (bf16 has 8 mantissa bits: per-token log-probs 1e-2-class,
    8-25x outside the north-star's 1e-3 bar; kept as a throughput / A-B type only).
"""
import json
import os

import numpy as np
import pytest
import torch

from tests import helpers
from oracle import decoding, logmel as olm, whisper_ref

pytestmark = pytest.mark.gpu

DIAG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")


def _tol(dtype, f32, bf16, f16=None):
    """Per-compute-type bound: float32 carries the north-star bar, the 16-bit types are compared with the oracle run
    at their own rounding points (fp16 has 3 more mantissa bits than bf16: 1/8 of its bound unless stated)."""
    return {"float32": f32, "bfloat16": bf16, "float16": bf16 / 8.0 if f16 is None else f16}[dtype]


def _diag(name, payload):
    os.makedirs(DIAG, exist_ok=True)
    with open(os.path.join(DIAG, "diag_pipeline.jsonl"), "a") as f:
        f.write(json.dumps({"test": name, **payload}) + "\n")


# ---------------------------------------------------------------------------------------------
# log-mel
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mode", ["fw", "ow"])
@pytest.mark.parametrize("n_mels", [80, 128])
def test_logmel_matches_oracle(hip, mode, n_mels):
    from whisperjav_amd import engine, synth
    audio = synth.speech_like(30.0, seed=1234)
    rng = np.random.default_rng(3)
    clips = [audio, audio[: 16000 * 6], audio[5000: 5000 + 16000 * 11 + 77],
             (rng.standard_normal(4000) * 0.1).astype(np.float32), np.zeros(16000, dtype=np.float32)]
    fe = engine.HipLogMel(n_mels, mode)
    got = fe(clips).cpu().numpy()
    assert got.shape == (len(clips), n_mels, 3000)
    worst = 0.0
    for i, c in enumerate(clips):
        ref = olm.window_features(c, n_mels, mode)
        assert fe.frames(len(c)) == (len(c) + (160 if mode == "fw" else 480000)) // 160
        d = np.abs(got[i] - ref).max()
        worst = max(worst, float(d))
        assert d < 5e-5, (i, d)            # measured 2.1e-5 on MI355X (profiles/): the bar is 2.4x that, not 10x
        if mode == "fw":  # zero padding of the frame axis is exact
            nf = min(3000, fe.frames(len(c)))
            assert np.all(got[i][:, nf:] == 0.0)
    _diag("logmel", {"mode": mode, "n_mels": n_mels, "max_abs": worst})


def test_logmel_long_clip_global_max(hip):
    """The clamp floor depends on the maximum over the WHOLE clip (non-local), also past out_frames."""
    from whisperjav_amd import engine, synth
    audio = synth.speech_like(42.0, seed=5)
    audio[16000 * 40: 16000 * 40 + 800] = 0.95  # loud burst after the 30 s mark
    fe = engine.HipLogMel(128, "fw")
    got = fe([audio], out_frames=fe.frames(len(audio))).cpu().numpy()[0]
    ref = olm.logmel_fw(audio, 128)
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() < 5e-5
    first = fe([audio], out_frames=3000).cpu().numpy()[0]
    assert np.abs(first - ref[:, :3000]).max() < 5e-5


def test_logmel_bit_reproducible(hip):
    from whisperjav_amd import engine, synth
    audio = synth.speech_like(20.0, seed=9)
    fe = engine.HipLogMel(128, "fw")
    a = fe([audio, audio[:50000]]).cpu().numpy()
    b = fe([audio[:50000], audio]).cpu().numpy()
    assert np.array_equal(a[0], b[1]) and np.array_equal(a[1], b[0])


# ---------------------------------------------------------------------------------------------
# Whisper engine, small model (oracle finishes in seconds)
# ---------------------------------------------------------------------------------------------
SMALL = dict(n_mels=80, d_model=128, heads=2, layers=2, n_vocab=51865)


def _engine_and_oracle(dtype, dims_kw=SMALL, seed=21, max_batch=3, max_beam=1):
    from whisperjav_amd import engine
    d = helpers.small_dims(**dims_kw)
    oracle, w = helpers.make_oracle(d, seed=seed, emulate=dtype)
    model = engine.HipWhisper(d, w, dtype=dtype, max_batch=max_batch, max_beam=max_beam)
    return d, oracle, model


@pytest.mark.parametrize("dtype", ["float32", "bfloat16", "float16"])
def test_encoder_layer_bisection(hip, dtype):
    """Stem only, one block, full stack: localises a kernel bug to a launch group."""
    d, oracle, model = _engine_and_oracle(dtype)
    mel = torch.from_numpy(helpers.synth_mel(2, d.n_mels, seed=4))
    tol = _tol(dtype, 2e-3, 6e-2, 8e-3)
    for n_layers in (0, 1, -1):
        with torch.no_grad():
            ref = oracle.encode(mel, n_layers=None if n_layers < 0 else n_layers, final_ln=n_layers < 0)
        got = model.encode(mel.cuda(), n_layers=n_layers, want_output=True).cpu()
        d_abs = (got - ref).abs()
        _diag("encoder_bisect", {"dtype": dtype, "n_layers": n_layers, "max_abs": float(d_abs.max()),
                                 "mean_abs": float(d_abs.mean()), "ref_rms": float(ref.pow(2).mean().sqrt())})
        assert float(d_abs.max()) < tol * max(1.0, float(ref.abs().max())), (n_layers, float(d_abs.max()))
    model.close()


@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
def test_encoder_blocked_operands_equal_row_major(hip, dtype):
    """Round 4: models with d_model % 256 == 0 run the encoder's big GEMMs on BLOCKED operands (weights copied into
    [N/256][K/32][256][32] at create, LayerNorm / attention / fc1 write the activations in that layout, the head-split
    epilogues take the window from the flat row index).  Same kernels' arithmetic in the same order: the encoder output and
    everything decoded from the cross K/V must be BIT-identical to the row-major path (wj_tune enc_blocked=0), and both
    within the 16-bit bound of the oracle.  3 windows of 1500 positions = 4500 rows: row blocks straddle windows and the
    last block is partial."""
    from whisperjav_amd import engine, hipbind
    dims_kw = dict(n_mels=80, d_model=256, heads=4, layers=2, n_vocab=51865)
    d = helpers.small_dims(**dims_kw)
    oracle, w = helpers.make_oracle(d, seed=33, emulate=dtype)
    mel = torch.from_numpy(helpers.synth_mel(3, d.n_mels, seed=9))
    toks = __import__("whisperjav_amd.dims", fromlist=["x"]).special_tokens(d.n_vocab)
    prompt = np.array([[toks.sot, toks.language_token(0), toks.transcribe]] * 3, dtype=np.int32)
    outs = {}
    try:
        for blocked in (1, 0):
            hipbind.tune("enc_blocked", blocked)
            model = engine.HipWhisper(d, w, dtype=dtype, max_batch=3)
            enc = model.encode(mel.cuda(), want_output=True).cpu()
            res = model.decode_greedy(prompt, engine.DecodeOptions(max_new_tokens=6))
            outs[blocked] = (enc, res.tokens.copy(), res.sum_logprob.copy())
            model.close()
    finally:
        hipbind.tune("enc_blocked", 1)
    assert torch.equal(outs[1][0], outs[0][0])
    assert np.array_equal(outs[1][1], outs[0][1]) and np.array_equal(outs[1][2], outs[0][2])
    with torch.no_grad():
        ref = oracle.encode(mel)
    tol = _tol(dtype, 2e-3, 6e-2, 8e-3)
    err = float((outs[1][0] - ref).abs().max())
    _diag("encoder_blocked", {"dtype": dtype, "max_abs": err, "ref_max": float(ref.abs().max())})
    assert err < tol * max(1.0, float(ref.abs().max())), err


@pytest.mark.parametrize("dtype", ["float32", "bfloat16", "float16"])
@pytest.mark.parametrize("without_timestamps", [False, True])
def test_greedy_decode_matches_oracle(hip, dtype, without_timestamps):
    from whisperjav_amd import engine
    d, oracle, model = _engine_and_oracle(dtype)
    mel = torch.from_numpy(helpers.synth_mel(3, d.n_mels, seed=8))
    suppress = (1, 2, 7, 8, 9, 10, 14, 25, 50256, 50360, 50361)
    n_new = 24
    prompt = model.sot_prompt("ja", "transcribe", without_timestamps)
    model.encode(mel.cuda())
    res = model.decode_greedy(np.tile(np.array(prompt, dtype=np.int32), (3, 1)),
                              engine.DecodeOptions(max_new_tokens=n_new, suppress_tokens=suppress,
                                                   without_timestamps=without_timestamps, max_initial_timestamp=1.0))
    cfg = decoding.FilterConfig(suppress_tokens=suppress, without_timestamps=without_timestamps,
                                max_initial_timestamp_index=50)
    with torch.no_grad():
        xa = oracle.encode(mel)
        ref = decoding.greedy_decode(oracle, xa, prompt, n_new, cfg)
    lp_tol = _tol(dtype, 1e-3, 0.15, 0.02)
    agree = 0
    worst = 0.0
    for r in range(3):
        got_t = res.tokens[r, : res.n_tokens[r]].tolist()
        n_common = 0
        for a, b in zip(got_t, ref.tokens[r]):
            if a != b:
                break
            n_common += 1
        agree += n_common
        # log-probs are compared on the common prefix plus the first diverging position's context
        for j in range(n_common):
            worst = max(worst, abs(float(res.token_logprob[r, j]) - ref.token_logprob[r][j]))
        if dtype == "float32":
            assert got_t == ref.tokens[r], (r, got_t, ref.tokens[r])
            assert abs(float(res.sum_logprob[r]) - float(ref.sum_logprob[r])) < 1e-3 * max(1, len(got_t))
        else:
            assert n_common >= min(4, len(ref.tokens[r])), (r, got_t, ref.tokens[r])
    _diag("greedy", {"dtype": dtype, "without_timestamps": without_timestamps, "common_tokens": agree,
                     "max_logprob_diff": worst, "no_speech_diff": float(np.abs(res.no_speech_prob - ref.no_speech_prob).max())})
    assert worst < lp_tol, worst
    assert np.abs(res.no_speech_prob - ref.no_speech_prob).max() < _tol(dtype, 1e-5, 1e-3, 2e-4)
    if not without_timestamps:  # structural timestamp rules: first token is a timestamp <= 1.0 s
        tb = model.tokens.timestamp_begin
        assert all(tb <= res.tokens[r, 0] <= tb + 50 for r in range(3))
    model.close()


def test_greedy_teacher_forced_bf16_logprobs(hip):
    """bf16 throughput mode: per-step log-probs of the ORACLE's token sequence, step API."""
    d, oracle, model = _engine_and_oracle("bfloat16")
    mel = torch.from_numpy(helpers.synth_mel(2, d.n_mels, seed=12))
    prompt = model.sot_prompt("ja", "transcribe", True)
    cfg = decoding.FilterConfig(without_timestamps=True, suppress_blank=False)
    with torch.no_grad():
        xa = oracle.encode(mel)
        ref = decoding.greedy_decode(oracle, xa, prompt, 12, cfg)
    model.encode(mel.cuda())
    model.open(2, 1)
    for p in prompt[:-1]:
        model.step(np.full(2, p, dtype=np.int32), want_logits=False)
    feed = np.full(2, prompt[-1], dtype=np.int32)
    worst = 0.0
    for i in range(8):
        model.step(feed)
        lp = torch.log_softmax(model.logits().cpu(), dim=-1)
        for r in range(2):
            tok = ref.tokens[r][i]
            worst = max(worst, abs(float(lp[r, tok]) - ref.token_logprob[r][i]))
        feed = np.array([ref.tokens[r][i] for r in range(2)], dtype=np.int32)
    _diag("teacher_forced_bf16", {"max_logprob_diff": worst})
    assert worst < 0.1, worst
    model.close()


def test_step_api_equals_greedy_and_beam_rebinding(hip):
    """The host-driven step API reproduces the device-resident greedy loop, and re-binding rows to a
    parent (beam search) reads the parent's KV history."""
    from whisperjav_amd import engine
    d, oracle, model = _engine_and_oracle("float32", max_batch=2)
    model2 = engine.HipWhisper(d, helpers.make_oracle(d, seed=21)[1], dtype="float32", max_batch=2, max_beam=2)
    mel = torch.from_numpy(helpers.synth_mel(2, d.n_mels, seed=15))
    prompt = model.sot_prompt("ja", "transcribe", True)
    opts = engine.DecodeOptions(max_new_tokens=6, without_timestamps=True, suppress_blank=False)
    model.encode(mel.cuda())
    g = model.decode_greedy(np.tile(np.array(prompt, dtype=np.int32), (2, 1)), opts)
    # beam=2 rows per window: row 2w follows greedy, row 2w+1 follows the 2nd best; then swap parents
    model2.encode(mel.cuda())
    model2.open(2, 2)
    for p in prompt[:-1]:
        model2.step(np.full(4, p, dtype=np.int32), want_logits=False)
    model2.step(np.full(4, prompt[-1], dtype=np.int32))
    ids, lps, _ = model2.topk(2)
    assert ids[0, 0] == g.tokens[0, 0] and ids[2, 0] == g.tokens[1, 0]
    assert abs(lps[0, 0] - g.token_logprob[0, 0]) < 1e-4
    # rows 0/2 take best, rows 1/3 take second best
    model2.step(np.array([ids[0, 0], ids[0, 1], ids[2, 0], ids[2, 1]], dtype=np.int32))
    ids2, lps2, _ = model2.topk(1)
    assert ids2[0, 0] == g.tokens[0, 1] and ids2[2, 0] == g.tokens[1, 1]
    # swap: new row 1 continues old row 0's history (parent 0), new row 0 continues old row 1
    model2.step(np.array([ids2[1, 0], ids2[0, 0], ids2[3, 0], ids2[2, 0]], dtype=np.int32),
                parents=np.array([1, 0, 3, 2], dtype=np.int32))
    ids3, lps3, _ = model2.topk(1)
    assert ids3[1, 0] == g.tokens[0, 2] and ids3[3, 0] == g.tokens[1, 2]
    assert abs(lps3[1, 0] - g.token_logprob[0, 2]) < 1e-4
    model.close()
    model2.close()


@pytest.mark.parametrize("dtype", ["float32", "bfloat16", "float16"])
def test_whisper_tiny_shape(hip, dtype):
    """The published 'tiny' geometry (384 / 6 heads / 4 layers), BASELINE config #1's model size."""
    from whisperjav_amd import dims as pdims, engine
    d = pdims.dims_for("tiny")
    oracle, w = helpers.make_oracle(d, seed=2, emulate=dtype)
    model = engine.HipWhisper(d, w, dtype=dtype, max_batch=1)
    mel = torch.from_numpy(helpers.synth_mel(1, d.n_mels, seed=1))
    with torch.no_grad():
        ref = oracle.encode(mel)
    got = model.encode(mel.cuda(), want_output=True).cpu()
    tol = _tol(dtype, 2e-3, 8e-2, 1e-2)
    assert float((got - ref).abs().max()) < tol * max(1.0, float(ref.abs().max()))
    model.close()


# ---------------------------------------------------------------------------------------------
# committed golden vectors (tests/golden/make_golden.py)
# ---------------------------------------------------------------------------------------------
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _golden_run(name, dims, dtype, n_mels):
    from whisperjav_amd import engine, synth, weights as pweights
    g = np.load(os.path.join(GOLDEN, name))
    audio = synth.speech_like(30.0, seed=1234)
    mel = engine.HipLogMel(n_mels, "fw")([audio])
    w = helpers.cached_weights(dims, int(g["seed"]))
    model = engine.HipWhisper(dims, w, dtype=dtype, max_batch=1)
    del w
    enc = model.encode(mel, want_output=True).cpu()
    n_new = len(g["tokens"])
    res = model.decode_greedy(np.array([g["prompt"]], dtype=np.int32),
                              engine.DecodeOptions(max_new_tokens=n_new, suppress_tokens=tuple(int(t) for t in g["suppress"]),
                                                   max_initial_timestamp=1.0))
    model.close()
    probe = enc[0][g["probe_t"]][:, g["probe_d"]].numpy()
    return g, enc, probe, res


@pytest.mark.parametrize("dtype", ["float32", "bfloat16", "float16"])
def test_golden_small_end_to_end(hip, dtype):
    """audio -> HIP log-mel -> encoder -> greedy decode against the committed fp32 oracle vectors."""
    g, enc, probe, res = _golden_run("golden_small.npz", helpers.small_dims(), dtype, 80)
    d_probe = float(np.abs(probe - g["enc_probe"]).max())
    n = int(res.n_tokens[0])
    got = res.tokens[0, :n].tolist()
    ref = g["tokens"].tolist()
    common = 0
    for a, b in zip(got, ref):
        if a != b:
            break
        common += 1
    d_lp = float(np.abs(res.token_logprob[0, :common] - g["token_logprob"][:common]).max()) if common else 0.0
    _diag("golden_small", {"dtype": dtype, "probe_max_abs": d_probe, "common": common, "lp_max_abs": d_lp})
    if dtype == "float32":
        assert d_probe < 2e-3 and got == ref and d_lp < 1e-3
    elif dtype == "float16":      # weights are bf16-exact here, so only the activations are rounded (2^-11)
        assert d_probe < 1.5e-2 and common >= 8 and d_lp < 2e-2
    else:
        assert d_probe < 0.1 and common >= 3 and d_lp < 0.15


@pytest.mark.parametrize("dtype", ["bfloat16", "float32"])
def test_golden_large_v3_window(hip, dtype):
    """BASELINE cfg2 geometry: Whisper large-v3 shape, one 30 s window, mel + encoder + greedy decode."""
    from whisperjav_amd import dims as pdims
    g, enc, probe, res = _golden_run("golden_large_v3.npz", pdims.dims_for("large-v3"), dtype, 128)
    d_probe = float(np.abs(probe - g["enc_probe"]).max())
    d_mean = abs(float(enc.abs().mean()) - float(g["enc_abs_mean"]))
    n = int(res.n_tokens[0])
    got = res.tokens[0, :n].tolist()
    ref = g["tokens"].tolist()
    common = 0
    for a, b in zip(got, ref):
        if a != b:
            break
        common += 1
    d_lp = float(np.abs(res.token_logprob[0, :common] - g["token_logprob"][:common]).max()) if common else 0.0
    _diag("golden_large_v3", {"dtype": dtype, "probe_max_abs": d_probe, "abs_mean_diff": d_mean, "common": common,
                              "lp_max_abs": d_lp, "tokens": got})
    if dtype == "float32":
        assert d_probe < 5e-3 and d_mean < 1e-4
        assert got == ref, (got, ref)
        assert d_lp < 1e-3, d_lp
    else:
        assert d_probe < 0.25 and d_mean < 5e-3
        assert common >= 3 and d_lp < 0.2


# Round-2 goldens: large-v3 geometry, weights NOT pre-rounded to the engine's storage type, 32 greedy tokens + the
# BASELINE cfg3 search (beam 5, patience 1.2, repetition penalty 1.5, no-repeat-3-gram).  Bars:
#   * float32: tokens identical, per-token log-probs within 1e-3 (the north-star bar; measured ~1e-5);
#   * float16 (the reference's GPU arithmetic, split-activation decode GEMMs): tokens and the winning beam identical
#     (zero flips); per-token log-probs within 1e-3 when the weights are fp16-representable (as the published
#     checkpoints are) and within 2.5e-3 on the raw fp32 draws, where storing the weights in fp16 alone moves the
#     oracle's log-probs by 1.05e-3 (profiles/r02_precision_ablation_cpu.json, case only_weights16);
#   * bfloat16: reported, bounded loosely (8 mantissa bits; not the default compute type any more).
R2_LP_BAR = {("float32", "none"): 1e-3, ("float32", "float16"): 1e-3, ("float16", "float16"): 1e-3,
             ("float16", "none"): 2.5e-3, ("bfloat16", "none"): 0.1, ("bfloat16", "float16"): 0.1}


@pytest.mark.parametrize("exact", ["float16", "none"])
@pytest.mark.parametrize("dtype", ["float16", "float32", "bfloat16"])
def test_golden_large_v3_r2_greedy_and_beam(hip, dtype, exact):
    from whisperjav_amd import dims as pdims, engine, synth, weights as pweights
    g = np.load(os.path.join(GOLDEN, f"golden_large_v3_r2_{exact}.npz"))
    dims = pdims.dims_for("large-v3")
    audio = synth.speech_like(30.0, seed=1234)
    mel = engine.HipLogMel(128, "fw")([audio])
    w = helpers.cached_weights(dims, int(g["seed"]), exact)
    model = engine.HipWhisper(dims, w, dtype=dtype, max_batch=1, max_beam=5)
    del w
    enc = model.encode(mel, want_output=True).cpu()
    probe = enc[0][g["probe_t"]][:, g["probe_d"]].numpy()
    d_probe = float(np.abs(probe - g["enc_probe"]).max())
    sup = tuple(int(t) for t in g["suppress"])
    prompt = np.array([g["prompt"]], dtype=np.int32)
    ref_t = g["tokens"].tolist()
    res = model.decode_greedy(prompt, engine.DecodeOptions(max_new_tokens=len(ref_t), suppress_tokens=sup, max_initial_timestamp=1.0))
    got = res.tokens[0, : int(res.n_tokens[0])].tolist()
    common = next((i for i, (a, b) in enumerate(zip(got, ref_t)) if a != b), min(len(got), len(ref_t)))
    d_lp = float(np.abs(res.token_logprob[0, :common] - g["token_logprob"][:common]).max()) if common else 0.0
    beam, patience, lpen, rep, ngram, n_new = (float(x) for x in g["beam"])
    br = model.decode_beam(prompt, engine.DecodeOptions(max_new_tokens=int(n_new), suppress_tokens=sup, max_initial_timestamp=0.0,
                                                       repetition_penalty=rep, no_repeat_ngram_size=int(ngram)),
                           beam_size=int(beam), patience=patience, length_penalty=lpen)
    model.close()
    b_got = br.tokens[0, : int(br.n_tokens[0])].tolist()
    bar = R2_LP_BAR[(dtype, exact)]
    margin = float(g["beam_norm"][0] - g["beam_norm"][1]) if len(g["beam_norm"]) > 1 else float("inf")
    # The oracle's own winner beats its runner-up by 1.6e-4 in normalised score on this golden -- a tie inside the per-token
    # bar.  (A golden whose margin is FAR outside the bar, where a real flip cannot hide behind this rule, is the first clip of
    # golden_large_v3_r3_eot.npz: margin 0.53 in normalised score, winner required identical in tests/test_gpu_search_eot.py.)  A 16-bit engine may land on either side of it (round 4: a different fp32 summation order in the encoder's
    # LayerNorm moved the fp16 run across); every oracle hypothesis within the bar of the best is an acceptable winner, anything
    # else is a flip.
    accept = [j for j in range(len(g["beam_len"])) if float(g["beam_norm"][0] - g["beam_norm"][j]) <= bar] if dtype != "float32" else [0]
    j_hit = next((j for j in accept if b_got == g["beam_tokens"][j, : int(g["beam_len"][j])].tolist()), 0)
    b_ref = g["beam_tokens"][j_hit, : int(g["beam_len"][j_hit])].tolist()
    d_cum = abs(float(br.sum_logprob[0]) - float(g["beam_cum"][j_hit]))
    _diag("golden_large_v3_r2", {"dtype": dtype, "weights": exact, "probe_max_abs": d_probe, "greedy_common": common,
                                 "greedy_n": len(ref_t), "greedy_lp_max_abs": d_lp, "beam_same": b_got == b_ref,
                                 "beam_cum_abs": d_cum, "beam_len": len(b_ref), "beam_norm_margin_to_runner_up": margin,
                                 "beam_hypothesis_matched": j_hit, "beam_hypotheses_inside_the_bar": len(accept),
                                 "greedy_sum_abs": abs(float(res.sum_logprob[0]) - float(g["sum_logprob"][0]))})
    if dtype == "bfloat16":
        assert common >= 3 and d_lp < bar
        return
    assert got == ref_t, (common, got, ref_t)                      # zero token flips over 32 greedy tokens
    assert d_lp < bar, d_lp
    assert b_got == b_ref, (b_got, b_ref)                          # the same winning hypothesis
    assert d_cum < bar * max(1, len(b_ref)), d_cum                 # cumulative log-prob: the per-token bar x length
    assert abs(float(br.no_speech_prob[0]) - float(g["beam_no_speech"])) < (1e-5 if dtype == "float32" else 1e-3)


# ---------------------------------------------------------------------------------------------
# beam search through the device scorer (wj_decode_topk_rules) vs the oracle's CTranslate2 restatement
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", ["float32", "bfloat16", "float16"])
@pytest.mark.parametrize("beam,patience,rep,ngram", [(2, 1.2, 1.5, 3), (5, 1.2, 1.5, 3), (3, 1.0, 1.0, 0)])
def test_beam_search_matches_oracle(hip, dtype, beam, patience, rep, ngram):
    from whisperjav_amd import engine, search
    d = helpers.small_dims()
    oracle, w = helpers.make_oracle(d, seed=33, emulate=dtype)
    model = engine.HipWhisper(d, w, dtype=dtype, max_batch=2, max_beam=beam)
    mel = torch.from_numpy(helpers.synth_mel(2, d.n_mels, seed=19))
    toks = model.tokens
    prompt = model.sot_prompt("ja", "transcribe")
    suppress = (toks.sot, toks.translate, toks.transcribe, toks.sot_lm, toks.sot_prev, toks.no_speech)
    opts = search.SearchOptions(beam_size=beam, patience=patience, length_penalty=1.0, repetition_penalty=rep,
                                no_repeat_ngram_size=ngram, suppress_tokens=suppress, max_initial_timestamp_index=0,
                                max_new_tokens=16)
    model.encode(mel.cuda())
    got = search.beam_search(search.HipStepScorer(model, opts), [prompt, prompt], opts, eot=toks.eot,
                             timestamp_begin=toks.timestamp_begin)
    fcfg = decoding.FilterConfig(suppress_tokens=suppress, max_initial_timestamp_index=0)
    bcfg = decoding.BeamConfig(beam, patience, 1.0, rep, ngram, 16)
    with torch.no_grad():
        xa = oracle.encode(mel)
    worst = 0.0
    for wdx in range(2):
        ref, nsp = decoding.beam_search(oracle, xa[wdx:wdx + 1], prompt, bcfg, fcfg)
        if dtype == "float32":
            assert got[wdx].sequences[0] == ref[0][0], (got[wdx].sequences[0], ref[0][0])
            assert abs(got[wdx].cum_logprobs[0] - ref[0][2]) < 1e-3
            assert abs(got[wdx].no_speech_prob - nsp) < 1e-5
        else:
            common = 0
            for a, b in zip(got[wdx].sequences[0], ref[0][0]):
                if a != b:
                    break
                common += 1
            assert common >= min(3, len(ref[0][0]))
        worst = max(worst, abs(got[wdx].cum_logprobs[0] - ref[0][2]))
    _diag("beam", {"dtype": dtype, "beam": beam, "cum_logprob_diff": worst})
    model.close()


# ---------------------------------------------------------------------------------------------
# end-to-end through the reference-shaped surfaces
# ---------------------------------------------------------------------------------------------
def test_whisper_model_shim_end_to_end(hip):
    """HipWhisperModel.transcribe (faster-whisper's call contract) == oracle greedy on the same window,
    and transcribe_many == per-clip transcribe."""
    from whisperjav_amd import synth, weights as pweights, whisper_model as wm
    d = helpers.small_dims()
    w = pweights.synth_weights(d, seed=21)
    oracle = whisper_ref.WhisperOracle(helpers.oracle_dims(d), w)
    model = wm.HipWhisperModel("tiny", compute_type="float32", weights=w, dims=d, max_batch=4, max_beam=2)
    audio = synth.speech_like(9.0, seed=31)
    kw = dict(task="transcribe", language="ja", beam_size=1, temperature=0.0, condition_on_previous_text=False,
              suppress_tokens=[], max_new_tokens=20, no_speech_threshold=None, max_initial_timestamp=0.0,
              word_timestamps=False, log_progress=False, vad_filter=False)
    segs, info = model.transcribe(audio, **kw)
    segs = list(segs)
    assert info.duration == pytest.approx(9.0) and len(segs) >= 1
    got = [t for s in segs if s.seek == 0 for t in s.tokens]     # first window
    # oracle on faster-whisper's window: content_frames = frames - 1, zero padded features
    feat = olm.logmel_fw(audio, d.n_mels)
    win = np.zeros((1, d.n_mels, 3000), dtype=np.float32)
    win[0, :, : feat.shape[1] - 1] = feat[:, :-1]
    toks = model.tokens
    sup = tuple(sorted({toks.transcribe, toks.translate, toks.sot, toks.sot_prev, toks.sot_lm}))
    cfg = decoding.FilterConfig(suppress_tokens=sup, max_initial_timestamp_index=0)
    with torch.no_grad():
        ref = decoding.greedy_decode(oracle, oracle.encode(torch.from_numpy(win)), model.model.sot_prompt("ja"), 20, cfg)
    assert ref.tokens[0][: len(got)] == got or got == ref.tokens[0]
    assert abs(segs[0].avg_logprob - float(ref.avg_logprob()[0])) < 1e-3
    many, _ = model.transcribe_many([audio, audio[: 16000 * 4]], **kw)
    assert [t for s in many[0] if s.seek == 0 for t in s.tokens] == got
    beam, _ = model.transcribe(audio, **dict(kw, beam_size=2, patience=1.2, repetition_penalty=1.5, no_repeat_ngram_size=3))
    assert len(list(beam)) >= 1
    # the temperature ladder: an unreachable log-prob bar makes every rung fail -> best average log-prob is kept
    # (never worse than the zero-temperature one) and the LAST temperature is reported; without bars nothing falls back
    # condition_on_previous_text=True makes prompt lengths differ between clips from the second window on: the batched
    # path then decodes per length against slot-addressed resident windows and must equal the per-clip results
    long_a, long_b = synth.speech_like(41.0, seed=3), synth.speech_like(35.0, seed=4)
    ckw = dict(kw, condition_on_previous_text=True, max_new_tokens=12)
    both, _ = model.transcribe_many([long_a, long_b], **ckw)
    for clip, got_segs in zip((long_a, long_b), both):
        alone, _ = model.transcribe(clip, **ckw)
        assert [(s.seek, s.tokens) for s in alone] == [(s.seek, s.tokens) for s in got_segs]
    lad, _ = model.transcribe(audio, **dict(kw, temperature=(0.0, 0.5, 1.0), best_of=2, log_prob_threshold=-1e-3,
                                             compression_ratio_threshold=None))
    lad = list(lad)
    assert lad and all(s.temperature == 1.0 for s in lad)
    assert lad[0].avg_logprob >= segs[0].avg_logprob - 1e-6
    calm, _ = model.transcribe(audio, **dict(kw, temperature=(0.0, 0.5, 1.0), best_of=2, log_prob_threshold=None,
                                              compression_ratio_threshold=None))
    assert [t for s in calm if s.seek == 0 for t in s.tokens] == got
    model.close()


def test_asr_adapter_with_hip_vad_writes_srt(hip, tmp_path):
    """Scene WAV -> HIP Silero-class segmenter -> grouped windows -> batched decode -> SRT."""
    import wave
    from whisperjav_amd import asr, segmenters, synth, weights as pweights, whisper_model as wm
    d = helpers.small_dims()
    audio = synth.speech_like(20.0, seed=4)
    path = tmp_path / "scene_0000.wav"
    with wave.open(str(path), "wb") as wf:
        wf.setnchannels(1); wf.setsampwidth(2); wf.setframerate(16000)
        wf.writeframes((np.clip(audio, -1, 1) * 32767).astype("<i2").tobytes())
    model = wm.HipWhisperModel("tiny", compute_type="bfloat16", weights=pweights.synth_weights(d, seed=21), dims=d,
                               max_batch=8, max_beam=2)
    seg = segmenters.HipSileroV6SpeechSegmenter(threshold=0.5, speech_pad_ms=100, max_group_duration_s=6.0, weights="synthetic")
    params = {"decoder": {"task": "transcribe", "language": "ja", "beam_size": 2, "patience": 1.2, "suppress_tokens": None,
                          "temperature": [0.0], "max_initial_timestamp": 0.0, "no_speech_threshold": None,
                          "logprob_threshold": -1.0, "condition_on_previous_text": False, "max_new_tokens": 24},
              "provider": {"repetition_penalty": 1.5, "no_repeat_ngram_size": 3}, "vad": {"threshold": 0.5},
              "speech_segmenter": {"backend": "silero-v6.2-hip"}}
    a = asr.HipFasterWhisperProASR({"model_name": "tiny"}, params, "transcribe", whisper_model=model, segmenter=seg)
    out = a.transcribe_to_srt(path, tmp_path / "scene_0000.srt")
    text = out.read_text(encoding="utf-8")
    vad = a.get_last_vad_segments()
    _diag("asr_adapter", {"vad_segments": len(vad), "srt_bytes": len(text)})
    assert len(vad) >= 1 and all(0 <= v["start_sec"] < v["end_sec"] <= 20.0 for v in vad)
    assert "-->" in text
    a.cleanup()


def test_decode_chains_are_equivalent(hip, monkeypatch):
    """The greedy loop splits the rows into concurrent chains on forked streams (captured into one
    hipGraph); results must not depend on the number of chains."""
    from whisperjav_amd import engine
    d, oracle, model = _engine_and_oracle("float32", max_batch=6)
    mel = torch.from_numpy(helpers.synth_mel(6, d.n_mels, seed=23))
    prompt = np.tile(np.array(model.sot_prompt("ja"), dtype=np.int32), (6, 1))
    opts = engine.DecodeOptions(max_new_tokens=20, max_initial_timestamp=1.0)
    model.encode(mel.cuda())
    outs = []
    for chains in ("1", "2", "3"):
        monkeypatch.setenv("WJ_DECODE_CHAINS", chains)
        res = model.decode_greedy(prompt, opts)
        info = model.last_decode_info()
        assert info["chains"] == int(chains) and info["hip_graph"] is True, info
        outs.append(res)
    for res in outs[1:]:
        assert np.array_equal(res.tokens, outs[0].tokens)
        assert np.array_equal(res.token_logprob, outs[0].token_logprob)
        assert np.array_equal(res.sum_logprob, outs[0].sum_logprob)
    with torch.no_grad():
        ref = decoding.greedy_decode(oracle, oracle.encode(mel), model.sot_prompt("ja"), 20,
                                     decoding.FilterConfig(max_initial_timestamp_index=50))
    for r in range(6):
        assert outs[1].tokens[r, : outs[1].n_tokens[r]].tolist() == ref.tokens[r]
    model.close()


def test_fidelity_flavour_end_to_end(hip):
    """openai-whisper call contract on the engine: 'ow' log-mel semantics + BeamSearchDecoder-style search,
    checked against the oracle's literal restatement on the same window."""
    from whisperjav_amd import synth, weights as pweights, whisper_model as wm
    d = helpers.small_dims()
    w = pweights.synth_weights(d, seed=21)
    oracle = whisper_ref.WhisperOracle(helpers.oracle_dims(d), w)
    model = wm.HipOpenAIWhisperModel("tiny", compute_type="float32", weights=w, dims=d, max_batch=2, max_beam=3)
    audio = synth.speech_like(7.0, seed=13)
    res = model.transcribe(audio, verbose=None, fp16=False, temperature=(0.0,), beam_size=3, patience=None,
                           length_penalty=None, suppress_tokens=[], condition_on_previous_text=False,
                           no_speech_threshold=None, logprob_threshold=None, language="ja", task="transcribe",
                           sample_len=18)
    assert set(res) == {"text", "segments", "language"} and len(res["segments"]) >= 1
    got = [t for s in res["segments"] for t in s["tokens"]]
    mel = olm.logmel_ow(audio, d.n_mels)
    content = mel.shape[1] - 3000
    win = np.zeros((1, d.n_mels, 3000), dtype=np.float32)
    win[0, :, :content] = mel[:, :content]
    toks = model.tokens
    sup = tuple(sorted({toks.transcribe, toks.translate, toks.sot, toks.sot_prev, toks.sot_lm, toks.no_speech}))
    cfg = decoding.FilterConfig(suppress_tokens=sup, max_initial_timestamp_index=50)
    with torch.no_grad():
        xa = oracle.encode(torch.from_numpy(win))
        seq, total, avg, nsp = decoding.beam_search_openai(oracle, xa, model.model.sot_prompt("ja"), 3, None, None, 18, cfg)
    first_window = [t for s in res["segments"] if s["seek"] == 0 for t in s["tokens"]]
    assert seq[: len(first_window)] == first_window or first_window == seq, (first_window, seq)
    assert abs(res["segments"][0]["avg_logprob"] - avg) < 1e-3
    _diag("fidelity", {"tokens": len(got), "avg_logprob_diff": abs(res["segments"][0]["avg_logprob"] - avg)})
    model.close()


# ---------------------------------------------------------------------------------------------
# edge cases, full-size properties, determinism
# ---------------------------------------------------------------------------------------------
def test_logmel_full_size_properties(hip):
    """BASELINE-scale input: 10 min of noisy audio in one launch -- frame count, agreement with the oracle on a
    sampled set of frames, global-max clamp floor, and bit-identical repeat."""
    from whisperjav_amd import engine, synth
    audio = synth.speech_like(600.0, seed=1234, noisy=True)
    fe = engine.HipLogMel(128, "fw")
    n_frames = fe.frames(len(audio))
    assert n_frames == (len(audio) + 160) // 160 == 60001
    got = fe([audio], out_frames=n_frames).cpu().numpy()[0]
    again = fe([audio], out_frames=n_frames).cpu().numpy()[0]
    assert np.array_equal(got, again)
    ref = olm.logmel_fw(audio, 128)
    assert got.shape == ref.shape
    cols = np.r_[0:50, 29990:30010, 59950:60001, np.arange(100, 60000, 997)]
    assert np.abs(got[:, cols] - ref[:, cols]).max() < 5e-5
    assert abs(float(got.min()) - float(ref.min())) < 1e-6            # clamp floor = (global max - 8 + 4) / 4
    assert abs(float(got.astype(np.float64).sum()) - float(ref.astype(np.float64).sum())) < 1e-4 * abs(float(ref.sum()))


def test_logmel_minimum_clip_and_rejects_too_short(hip):
    from whisperjav_amd import engine, hipbind
    fe = engine.HipLogMel(80, "fw")
    rng = np.random.default_rng(0)
    clip = (rng.standard_normal(201) * 0.1).astype(np.float32)
    got = fe([clip]).cpu().numpy()[0]
    ref = olm.window_features(clip, 80, "fw")
    assert np.abs(got - ref).max() < 5e-5
    with pytest.raises(hipbind.WjError):
        fe([clip[:200]])


def test_vad_degenerate_streams(hip):
    from oracle import silero_ref
    from whisperjav_amd import vad, vad_weights
    w = vad_weights.synth_weights()
    scorer = vad.HipSileroScorer(w)
    rng = np.random.default_rng(1)
    clips = [np.zeros(0, np.float32), (rng.standard_normal(1) * 0.1).astype(np.float32),
             (rng.standard_normal(512) * 0.1).astype(np.float32), (rng.standard_normal(513) * 0.1).astype(np.float32)]
    got = scorer.scores(clips)
    assert [len(g) for g in got] == [0, 1, 1, 2]
    oracle = silero_ref.SileroOracle(w)
    for c, g in zip(clips[1:], got[1:]):
        assert np.abs(g - oracle.probs(c)).max() < 1e-5
    assert vad.get_speech_timestamps(np.zeros(0, np.float32), scorer) == []
    scorer.close()


def test_decode_is_deterministic_and_respects_context_limit(hip):
    from whisperjav_amd import engine, hipbind
    d, oracle, model = _engine_and_oracle("bfloat16", max_batch=2)
    mel = torch.from_numpy(helpers.synth_mel(2, d.n_mels, seed=41))
    model.encode(mel.cuda())
    prompt = np.tile(np.array(model.sot_prompt("ja"), dtype=np.int32), (2, 1))
    a = model.decode_greedy(prompt, engine.DecodeOptions(max_new_tokens=445))     # 3 + 445 = n_text_ctx
    b = model.decode_greedy(prompt, engine.DecodeOptions(max_new_tokens=445))
    assert np.array_equal(a.tokens, b.tokens) and np.array_equal(a.token_logprob, b.token_logprob)
    assert a.tokens.shape == (2, 445)
    with pytest.raises(hipbind.WjError):
        model.decode_greedy(prompt, engine.DecodeOptions(max_new_tokens=446))
    with pytest.raises(hipbind.WjError):
        model.encode(torch.zeros((3, d.n_mels, 3000), device="cuda"))               # more windows than max_batch
    model.close()


# ---------------------------------------------------------------------------------------------
# device sampler: logits processors and temperature sampling (the fallback ladder's rungs)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("rep,ngram", [(1.5, 0), (1.0, 2), (1.3, 3)])
def test_greedy_device_processors_match_oracle(hip, rep, ngram):
    """beam_size=1 with repetition_penalty / no_repeat_ngram_size stays in the device loop: token ids identical,
    per-token log-probs within 1e-3 (fp32) of the oracle's ctranslate2-style processors + Whisper rules."""
    from whisperjav_amd import engine
    d, oracle, model = _engine_and_oracle("float32")
    mel = torch.from_numpy(helpers.synth_mel(3, d.n_mels, seed=31))
    suppress = (1, 2, 7, 8, 9, 10, 14, 25, 50256, 50360, 50361)
    n_new = 28
    prompt = model.sot_prompt("ja", "transcribe", False)
    model.encode(mel.cuda())
    res = model.decode_greedy(np.tile(np.array(prompt, dtype=np.int32), (3, 1)),
                              engine.DecodeOptions(max_new_tokens=n_new, suppress_tokens=suppress, max_initial_timestamp=1.0,
                                                   repetition_penalty=rep, no_repeat_ngram_size=ngram))
    cfg = decoding.FilterConfig(suppress_tokens=suppress, max_initial_timestamp_index=50)
    with torch.no_grad():
        ref = decoding.greedy_decode(oracle, oracle.encode(mel), prompt, n_new, cfg,
                                     processors=decoding.BeamConfig(repetition_penalty=rep, no_repeat_ngram_size=ngram))
        plain = decoding.greedy_decode(oracle, oracle.encode(mel), prompt, n_new, cfg)
    worst = 0.0
    for r in range(3):
        got = res.tokens[r, : res.n_tokens[r]].tolist()
        assert got == ref.tokens[r], (r, got, ref.tokens[r])
        for j in range(len(got)):
            worst = max(worst, abs(float(res.token_logprob[r, j]) - ref.token_logprob[r][j]))
    assert worst < 1e-3, worst
    assert any(ref.tokens[r] != plain.tokens[r] for r in range(3)), "processors had no effect: test is vacuous"
    _diag("greedy_processors", {"rep": rep, "ngram": ngram, "max_logprob_diff": worst})
    model.close()


def test_sampling_small_temperature_equals_greedy_and_is_seeded(hip):
    from whisperjav_amd import engine
    d, oracle, model = _engine_and_oracle("float32", max_batch=4, max_beam=8)
    mel = torch.from_numpy(helpers.synth_mel(4, d.n_mels, seed=32))
    prompt = np.tile(np.array(model.sot_prompt("ja", "transcribe", False), dtype=np.int32), (4, 1))
    o = engine.DecodeOptions(max_new_tokens=20, max_initial_timestamp=1.0)
    model.encode(mel.cuda())
    g = model.decode_greedy(prompt, o)
    # group = 1, temperature 0 through the general entry == greedy, bit for bit
    z = model.decode_sample(prompt, o, temperature=0.0, best_of=1)
    assert np.array_equal(z.tokens, g.tokens) and np.array_equal(z.sum_logprob, g.sum_logprob)
    # T -> 0+: the Gumbel noise is scaled out (logit gaps / 1e-4 dwarf it)
    c = model.decode_sample(prompt, o, temperature=1e-4, best_of=1, seed=5)
    assert np.array_equal(c.tokens, g.tokens)
    assert np.abs(c.sum_logprob - g.sum_logprob).max() < 1e-4
    # a subset of the resident windows through the slot map, two samples each: rows are window-major
    s = model.decode_sample(prompt[[3, 1]], o, temperature=0.0, best_of=2, slots=[3, 1])
    for row, w in enumerate([3, 3, 1, 1]):
        assert np.array_equal(s.tokens[row], g.tokens[w]), (row, w)
        assert abs(float(s.sum_logprob[row]) - float(g.sum_logprob[w])) < 1e-4
        assert abs(float(s.no_speech_prob[row]) - float(g.no_speech_prob[w])) < 1e-6
    # seeded: same seed -> same draw, different seed -> a different one; rows of one window differ from each other
    a = model.decode_sample(prompt, o, temperature=1.0, best_of=4, seed=11)
    b = model.decode_sample(prompt, o, temperature=1.0, best_of=4, seed=11)
    c2 = model.decode_sample(prompt, o, temperature=1.0, best_of=4, seed=12)
    assert np.array_equal(a.tokens, b.tokens) and np.array_equal(a.sum_logprob, b.sum_logprob)
    assert not np.array_equal(a.tokens, c2.tokens)
    assert any(not np.array_equal(a.tokens[4 * w], a.tokens[4 * w + 1]) for w in range(4))
    model.close()


def test_sampled_sequences_obey_rules_and_report_unscaled_logprobs(hip):
    """Every sampled sequence must be possible under the Whisper rules, and its reported per-token log-probs are
    those of the UNSCALED filtered distribution: the fp32 oracle teacher-forced on the sampled tokens gives the
    same numbers (1e-3)."""
    from whisperjav_amd import engine
    d, oracle, model = _engine_and_oracle("float32", max_batch=4, max_beam=8)
    mel = torch.from_numpy(helpers.synth_mel(2, d.n_mels, seed=33))
    suppress = (1, 2, 7, 8, 9, 10, 14, 25, 50256, 50360, 50361)
    p = model.sot_prompt("ja", "transcribe", False)
    o = engine.DecodeOptions(max_new_tokens=16, suppress_tokens=suppress, max_initial_timestamp=1.0)
    model.encode(mel.cuda())
    G = 5
    s = model.decode_sample(np.tile(np.array(p, dtype=np.int32), (2, 1)), o, temperature=0.8, best_of=G, seed=77)
    cfg = decoding.FilterConfig(suppress_tokens=suppress, max_initial_timestamp_index=50)
    worst = 0.0
    with torch.no_grad():
        xa = oracle.encode(mel)
        forced = [s.tokens[r, : min(16, s.n_tokens[r] + 1)].tolist() for r in range(2 * G)]
        ref = decoding.greedy_decode(oracle, xa.repeat_interleave(G, dim=0), p, 16, cfg, forced=forced)
    for r in range(2 * G):
        n = len(forced[r])
        for j in range(n):
            assert np.isfinite(ref.token_logprob[r][j]), (r, j, forced[r])      # the rules allow the sampled token
            worst = max(worst, abs(float(s.token_logprob[r, j]) - ref.token_logprob[r][j]))
        assert abs(float(s.sum_logprob[r]) - float(np.sum(ref.token_logprob[r][:n]))) < 1e-3 * n
    _diag("sampling_forced", {"max_logprob_diff": worst})
    assert worst < 1e-3, worst
    model.close()


def test_sampling_first_token_frequencies_follow_softmax(hip):
    """Statistical check of the Gumbel-max draw: 8 samples/window/call x 96 calls of ONE step; the empirical
    distribution of the first token matches softmax(filtered logits / T) from the oracle (chi-square style bound)."""
    from whisperjav_amd import engine
    d, oracle, model = _engine_and_oracle("float32", max_batch=4, max_beam=8)
    mel = torch.from_numpy(helpers.synth_mel(1, d.n_mels, seed=34))
    p = model.sot_prompt("ja", "transcribe", False)
    T = 2.5   # flatten the synthetic model's first-step distribution over the 51 allowed initial timestamps
    o = engine.DecodeOptions(max_new_tokens=1, max_initial_timestamp=1.0)
    model.encode(mel.cuda())
    counts = {}
    n_draws = 0
    for call in range(96):
        s = model.decode_sample(np.array([p], dtype=np.int32), o, temperature=T, best_of=8, seed=1000 + call)
        for r in range(8):
            counts[int(s.tokens[r, 0])] = counts.get(int(s.tokens[r, 0]), 0) + 1
            n_draws += 1
    with torch.no_grad():
        xa = oracle.encode(mel)
        dec = whisper_ref.CachedDecoder(oracle, xa)
        for t in p:
            lg = dec.step(torch.tensor([[t]]))
        filt = decoding.filter_logits(lg, [list(p)], len(p), decoding.TokenLayout.for_vocab(d.n_vocab),
                                      decoding.FilterConfig(max_initial_timestamp_index=50))
        probs = torch.softmax(filt[0].double() / T, dim=-1).numpy()
    assert all(probs[t] > 0 for t in counts)                       # nothing outside the allowed set was drawn
    support = np.nonzero(probs > 0)[0]
    chi2 = sum((counts.get(int(t), 0) - n_draws * probs[t]) ** 2 / (n_draws * probs[t]) for t in support if n_draws * probs[t] >= 5)
    dof = sum(1 for t in support if n_draws * probs[t] >= 5) - 1
    _diag("sampling_chi2", {"chi2": float(chi2), "dof": int(dof), "draws": n_draws, "distinct": len(counts)})
    assert dof >= 3, "distribution too peaked for the test to mean anything"
    assert chi2 < dof + 5 * np.sqrt(2 * dof), (chi2, dof)         # > 5 sigma of the chi-square would be a broken sampler
    model.close()


# ---------------------------------------------------------------------------------------------
# word-timestamp alignment (wj_whisper_align vs oracle/alignment.py = whisper/timing.py restated)
# ---------------------------------------------------------------------------------------------

def test_alignment_subcalls_over_similar_lengths_equal_one_call(hip):
    """Round 6: ``HipWhisper.align`` runs a pooled call as sub-calls over windows of similar token count (a single call pads every
    window to the longest).  A window's result does not depend on its neighbours: 96 resident windows with 2 ... 40 text tokens,
    float32 -- text / frame indices and token probabilities identical to the one-call pass, window by window, in caller order."""
    d, _, model = _engine_and_oracle("float32", max_batch=96)
    mel = torch.from_numpy(helpers.synth_mel(96, d.n_mels, seed=43))
    model.encode(mel.cuda())
    t = model.tokens
    sot_seq = model.sot_prompt("ja", "transcribe", True)[:-1]
    rng = np.random.default_rng(12)
    texts = [rng.integers(10, 2000, size=int(n)).tolist() for n in rng.integers(2, 41, size=96)]
    frames = [int(f) for f in rng.integers(400, 3001, size=96)]
    heads = [(0, 1), (1, 0), (1, 1)]
    rows = [[*sot_seq, t.no_timestamps, *tx, t.eot] for tx in texts]
    slots = list(rng.permutation(96))
    model.align_waste = None
    one = model.align(rows, len(sot_seq) + 1, heads, frames, slots=slots)
    model.align_waste, model.align_min_rows = 0.25, 16
    sub = model.align(rows, len(sot_seq) + 1, heads, frames, slots=slots)
    model.close()
    assert len(one) == len(sub) == 96
    for (a_t, a_f, a_p), (b_t, b_f, b_p) in zip(one, sub):
        assert np.array_equal(a_t, b_t) and np.array_equal(a_f, b_f) and np.allclose(a_p, b_p, atol=1e-6)


@pytest.mark.parametrize("prefill", [1, 0])
@pytest.mark.parametrize("dtype", ["float32", "bfloat16", "float16"])
def test_alignment_matches_oracle(hip, dtype, prefill):
    """Teacher-forced pass + softmax / normalise / median filter / head mean / DTW on the device.  float32: the DTW
    path (integer token / frame indices) is identical to the oracle's and the text-token probabilities agree to
    1e-4; bfloat16 (throughput mode): the path may wander by a few 20 ms frames."""
    from oracle import alignment
    from whisperjav_amd import hipbind
    hipbind.tune("align_prefill", prefill)      # 1: one full-sequence decoder pass (default), 0: token by token
    d, oracle, model = _engine_and_oracle(dtype, max_batch=3)
    mel = torch.from_numpy(helpers.synth_mel(3, d.n_mels, seed=41))
    model.encode(mel.cuda())
    t = model.tokens
    sot_seq = model.sot_prompt("ja", "transcribe", True)[:-1]          # sot, language, task
    texts = [[11, 500, 7, 7, 1234, 42, 9, 300, 301], [5], [900, 901, 902, 903, 904, 905, 906, 907, 908, 909, 910, 911, 912, 913]]
    frames = [3000, 1200, 2001]
    heads = [(0, 1), (1, 0), (1, 1)]
    rows = [[*sot_seq, t.no_timestamps, *tx, t.eot] for tx in texts]
    # windows 2 and 0 through the slot map, then all three in order
    got = model.align([rows[2], rows[0]], len(sot_seq) + 1, heads, [frames[2], frames[0]], slots=[2, 0])
    got = [got[1], model.align([rows[1]], len(sot_seq) + 1, heads, [frames[1]], slots=[1])[0], got[0]]
    worst_p, worst_t, worst_m, worst_x = 0.0, 0, 0.0, 0.0
    with torch.no_grad():
        xa = oracle.encode(mel)
    dev_m = {}
    if dtype != "float32":      # the matrices the device ran its DTWs on (one call per window: the export serves the LAST call)
        for w in range(3):
            model.align([rows[w]], len(sot_seq) + 1, heads, [frames[w]], slots=[w])
            dev_m[w] = model.last_align_matrix(1, len(texts[w]) + 1, frames[w] // 2)[0]
    for w in range(3):
        ti, fi, probs, matrix = alignment.find_alignment(oracle, xa[w:w + 1], sot_seq, t.no_timestamps, texts[w], t.eot, frames[w], heads)
        g_ti, g_fi, g_p = got[w]
        assert g_p.shape == probs.shape
        worst_p = max(worst_p, float(np.abs(g_p - probs).max()))
        assert g_ti[0] == 0 and g_fi[0] == 0 and g_ti[-1] == len(texts[w]) and g_fi[-1] == frames[w] // 2 - 1
        assert np.all(np.diff(g_ti) >= 0) and np.all(np.diff(g_fi) >= 0) and np.all(np.diff(g_ti) + np.diff(g_fi) >= 1)
        if dtype == "float32":
            assert np.array_equal(g_ti, ti) and np.array_equal(g_fi, fi), (w, len(ti), len(g_ti))
        else:   # first frame of every token row, compared in 20 ms units
            first = lambda a, b: np.array([b[np.argmax(a == k)] for k in range(len(texts[w]) + 1)])   # noqa: E731
            worst_t = max(worst_t, int(np.abs(first(g_ti, g_fi) - first(ti, fi)).max()))
            worst_m = max(worst_m, float(np.abs(dev_m[w] - matrix).max()))
            worst_x = max(worst_x, _path_excess(matrix, (g_ti, g_fi), (ti, fi)))
    hipbind.tune("align_prefill", 1)
    _diag("alignment", {"dtype": dtype, "prefill": prefill, "max_prob_diff": worst_p, "max_token_start_shift_frames": worst_t,
                        "max_matrix_diff": worst_m, "path_excess_per_cell_on_oracle_matrix": worst_x})
    assert worst_p < _tol(dtype, 1e-4, 2e-2, 4e-3), worst_p
    # integer outputs get an integer bar: float16 (the default compute type) reproduces the fp32 oracle's token start frames
    # exactly on this fixture.  bfloat16 (8 mantissa bits): the z-scored matrix itself agrees with the oracle's (bound below) but
    # where it is flat along a token row the optimal path is not unique to that precision; what IS asserted is that the device's
    # path, priced on the ORACLE's matrix, costs no more than the bound per path cell above the oracle's optimum (a near-tie),
    # and the start frames are recorded.  r05 root cause run: scripts/diag_align.py, profiles/r05_diag_align.jsonl
    if dtype == "float16":
        assert worst_t == 0, worst_t
        assert worst_m < 0.02 and worst_x < 1e-6, (worst_m, worst_x)        # measured 3.9e-3 / 0 (profiles/r05_parity_diag_pipeline.jsonl)
    elif dtype == "bfloat16":
        assert worst_m < 0.15 and worst_x < 1e-4, (worst_m, worst_x)        # measured 4.0e-2 / 4.3e-6
        assert worst_t <= 25, worst_t                                       # measured 10: recorded, the near-tie bound above is the bar
    model.close()


def _path_cost(matrix, path):
    ti, fi = path
    return float(-matrix[np.asarray(ti), np.asarray(fi)].astype(np.float64).sum())


def _path_excess(matrix, path, optimal):
    """How much dearer ``path`` is than ``optimal`` under the DTW objective of ``matrix`` (sum of negated entries along the path),
    per cell of the path: 0 for the optimum itself, ~ the matrix noise for a near-tie, O(1) for a wrong alignment."""
    return max(0.0, _path_cost(matrix, path) - _path_cost(matrix, optimal)) / max(1, len(path[0]))


def test_word_timestamps_through_the_shim(hip):
    """word_timestamps=True end to end (faster-whisper call contract): every segment carries words whose times are
    monotone, inside the clip, in 0.01 s units, and whose token count adds up to the segment's text tokens."""
    from whisperjav_amd import synth, weights as pweights, whisper_model as wm
    d = helpers.small_dims()
    model = wm.HipWhisperModel("tiny", compute_type="float32", weights=pweights.synth_weights(d, seed=21), dims=d,
                               max_batch=4, max_beam=2)
    audio = synth.speech_like(9.0, seed=31)
    kw = dict(language="ja", beam_size=1, temperature=0.0, condition_on_previous_text=False, suppress_tokens=[],
              max_new_tokens=20, no_speech_threshold=None, max_initial_timestamp=0.0, word_timestamps=True)
    segs, _ = model.transcribe(audio, **kw)
    segs = list(segs)
    plain, _ = model.transcribe(audio, **dict(kw, word_timestamps=False))
    plain = list(plain)
    assert segs and segs[0].tokens == plain[0].tokens          # the alignment pass does not disturb decoding
    eot = model.tokens.eot
    for s in segs:
        assert s.words is not None
        assert len(s.words) == len([t for t in s.tokens if t < eot])      # IdTokenizer: one word per text token
        last = 0.0
        for w in s.words:
            assert 0.0 <= w.start <= w.end <= 30.0 and w.start >= last - 1e-9 and 0.0 <= w.probability <= 1.0
            assert abs(w.start * 100 - round(w.start * 100)) < 1e-6
            last = w.start
        if s.words:
            assert s.start <= s.words[0].end and s.end >= s.words[-1].start
    many, _ = model.transcribe_many([audio, audio[: 16000 * 4]], **kw)
    assert [[(w.start, w.end) for w in s.words] for s in many[0]] == [[(w.start, w.end) for w in s.words] for s in segs]
    model.close()


# ---------------------------------------------------------------------------------------------
# device-resident beam search (wj_whisper_decode_beam) vs the oracle's CTranslate2 restatement and vs the
# host-driven search over the step API
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", ["float32", "bfloat16", "float16"])
@pytest.mark.parametrize("beam,patience,rep,ngram,n_new", [(2, 1.2, 1.5, 3, 16), (5, 1.2, 1.5, 3, 16), (3, 1.0, 1.0, 0, 16),
                                                           (5, 2.0, 1.1, 2, 40), (8, 1.0, 1.0, 0, 9)])
def test_device_beam_search_matches_oracle(hip, dtype, beam, patience, rep, ngram, n_new):
    from whisperjav_amd import engine, search
    d = helpers.small_dims()
    oracle, w = helpers.make_oracle(d, seed=33, emulate=dtype)
    model = engine.HipWhisper(d, w, dtype=dtype, max_batch=3, max_beam=beam)
    mel = torch.from_numpy(helpers.synth_mel(3, d.n_mels, seed=19))
    toks = model.tokens
    prompt = model.sot_prompt("ja", "transcribe")
    suppress = (toks.sot, toks.translate, toks.transcribe, toks.sot_lm, toks.sot_prev, toks.no_speech)
    model.encode(mel.cuda())
    res = model.decode_beam(np.tile(np.array(prompt, dtype=np.int32), (3, 1)),
                            engine.DecodeOptions(max_new_tokens=n_new, suppress_tokens=suppress, max_initial_timestamp=0.0,
                                                 repetition_penalty=rep, no_repeat_ngram_size=ngram),
                            beam_size=beam, patience=patience, length_penalty=1.0)
    assert model.last_decode_info()["hip_graph"]
    # a subset of the resident windows through the slot map gives the same hypotheses
    sub = model.decode_beam(np.tile(np.array(prompt, dtype=np.int32), (2, 1)),
                            engine.DecodeOptions(max_new_tokens=n_new, suppress_tokens=suppress, max_initial_timestamp=0.0,
                                                 repetition_penalty=rep, no_repeat_ngram_size=ngram),
                            beam_size=beam, patience=patience, length_penalty=1.0, slots=[2, 0])
    for row, w in enumerate([2, 0]):
        assert sub.tokens[row, : sub.n_tokens[row]].tolist() == res.tokens[w, : res.n_tokens[w]].tolist()
        assert abs(float(sub.sum_logprob[row]) - float(res.sum_logprob[w])) < 1e-4
    # the host-driven restatement over the step API must agree with the device loop (same engine numerics)
    opts = search.SearchOptions(beam_size=beam, patience=patience, length_penalty=1.0, repetition_penalty=rep,
                                no_repeat_ngram_size=ngram, suppress_tokens=suppress, max_initial_timestamp_index=0,
                                max_new_tokens=n_new)
    host = search.beam_search(search.HipStepScorer(model, opts), [prompt] * 3, opts, eot=toks.eot,
                              timestamp_begin=toks.timestamp_begin)
    fcfg = decoding.FilterConfig(suppress_tokens=suppress, max_initial_timestamp_index=0)
    bcfg = decoding.BeamConfig(beam, patience, 1.0, rep, ngram, n_new)
    with torch.no_grad():
        xa = oracle.encode(mel)
    worst = 0.0
    for wdx in range(3):
        got = res.tokens[wdx, : res.n_tokens[wdx]].tolist()
        assert got == host[wdx].sequences[0], (wdx, got, host[wdx].sequences[0])
        assert abs(float(res.sum_logprob[wdx]) - host[wdx].cum_logprobs[0]) < 1e-3
        ref, nsp = decoding.beam_search(oracle, xa[wdx:wdx + 1], prompt, bcfg, fcfg)
        if dtype == "float32":
            assert got == ref[0][0], (wdx, got, ref[0][0])
            assert abs(float(res.sum_logprob[wdx]) - ref[0][2]) < 1e-3
            assert abs(float(res.token_logprob[wdx, 0]) - ref[0][1]) < 1e-3       # normalised score
            assert abs(float(res.no_speech_prob[wdx]) - nsp) < 1e-5
        worst = max(worst, abs(float(res.sum_logprob[wdx]) - ref[0][2]))
    _diag("device_beam", {"dtype": dtype, "beam": beam, "patience": patience, "cum_logprob_diff": worst})
    model.close()


# ---------------------------------------------------------------------------------------------
# scene detection: device frame energies + host tokenizer vs the reference driver's fixtures
# ---------------------------------------------------------------------------------------------
def test_scene_detection_matches_reference_fixtures(hip, tmp_path):
    """HipAuditokSceneDetector.split_clip reproduces, bit for bit (float equality of the boundaries), the scenes the
    REFERENCE's two-pass driver produced for the committed fixtures; detect_scenes writes the PCM16 WAVs."""
    import wave
    from whisperjav_amd import scenes, synth
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_scenes.json")))
    n_scenes = 0
    for case in gold["cases"]:
        audio = synth.speech_like(case["seconds"], seed=case["seed"], noisy=case["noisy"])
        det = scenes.HipAuditokSceneDetector(config=scenes.AuditokSceneConfig(**case["cfg"]))
        got, story = det.split_clip(audio, 16000)
        assert [list(x) for x in story] == case["story"], case["seed"]
        assert [[a, b, p, m.get("split_method")] for a, b, p, m in got] == case["scenes"], case["seed"]
        n_scenes += len(got)
    assert n_scenes > 100
    # file interface: WAVs on disk, legacy tuples, metadata
    case = gold["cases"][8]
    audio = synth.speech_like(case["seconds"], seed=case["seed"], noisy=case["noisy"])
    src = tmp_path / "clip.wav"
    with wave.open(str(src), "wb") as wf:
        wf.setnchannels(1); wf.setsampwidth(2); wf.setframerate(16000)
        wf.writeframes((np.clip(audio, -1, 1) * 32767).astype("<i2").tobytes())
    det = scenes.HipAuditokSceneDetector(**{"max_duration_s": 10.0, "pass1_energy_threshold": 55, "pass2_energy_threshold": 60,
                                            "pass1_max_silence_s": 1.0, "pass2_max_silence_s": 0.4})
    res = det.detect_scenes(src, tmp_path / "scenes", "clip")
    assert res.method == "auditok-hip" and res.num_scenes >= 10 and res.audio_duration_sec == pytest.approx(case["seconds"], abs=0.01)
    for path, s, e, dur in res.to_legacy_tuples():
        with wave.open(str(path), "rb") as wf:
            assert wf.getframerate() == 16000 and wf.getnframes() == int(e * 16000) - int(s * 16000)
        assert 0 <= s < e <= case["seconds"] + 1e-6 and dur <= 10.0 + 1e-9
    det.cleanup()


@pytest.mark.parametrize("dtype", ["float16", "bfloat16"])
def test_large_v3_geometry_beam_and_alignment_consistency(hip, dtype):
    """BASELINE geometry (large-v3 shape; float16 = the default compute type, bfloat16 = the throughput type): properties that
    need no CPU oracle run at full size -- the device-resident beam search equals the host-driven restatement over the step API, a
    window's result does not depend on its batch neighbours, and the full-sequence alignment pass agrees with the token-by-token
    one.

    Alignment bar (r05 root cause of the r04 driver failure, scripts/diag_align.py -> profiles/r05_diag_align.jsonl): both passes
    are deterministic and state-free (bit-identical on repetition and in a fresh engine), they differ in fp32 summation order
    (tile GEMMs / MFMA attention vs row kernels with split-K), and whisper's (w - mean) / std over the token axis amplifies
    that noise wherever a frame's weights barely vary over the tokens.  float16: token start frames IDENTICAL (0 frames, also on
    weights.SPEECHLIKE).  bfloat16: any change of summation order moves the path through flat regions (9 frames with the
    vectorised LayerNorm, 3 without, 26 on peaked weights): the paths are compared through the DTW objective instead -- each
    pass's path priced on the OTHER pass's matrix is a near-tie of that matrix's optimum, and the matrices agree."""
    from whisperjav_amd import dims as pdims, engine, hipbind, search, synth, weights as pweights
    dims = pdims.dims_for("large-v3")
    model = engine.HipWhisper(dims, helpers.cached_weights(dims, 1234), dtype=dtype, max_batch=3, max_beam=5)
    fe = engine.HipLogMel(128, "fw")
    clips = [synth.speech_like(30.0, seed=1234), synth.speech_like(11.0, seed=77), synth.speech_like(30.0, seed=5)]
    model.encode(fe(clips))
    toks = model.tokens
    prompt = model.sot_prompt("ja", "transcribe")
    suppress = (toks.sot, toks.translate, toks.transcribe, toks.sot_lm, toks.sot_prev, toks.no_speech)
    o = engine.DecodeOptions(max_new_tokens=14, suppress_tokens=suppress, max_initial_timestamp=0.0, repetition_penalty=1.5,
                             no_repeat_ngram_size=3)
    P = np.tile(np.array(prompt, dtype=np.int32), (3, 1))
    dev = model.decode_beam(P, o, beam_size=5, patience=1.2, length_penalty=1.0)
    so = search.SearchOptions(beam_size=5, patience=1.2, length_penalty=1.0, repetition_penalty=1.5, no_repeat_ngram_size=3,
                              suppress_tokens=suppress, max_initial_timestamp_index=0, max_new_tokens=14)
    host = search.beam_search(search.HipStepScorer(model, so), [prompt] * 3, so, eot=toks.eot, timestamp_begin=toks.timestamp_begin)
    for w in range(3):
        assert dev.tokens[w, : dev.n_tokens[w]].tolist() == host[w].sequences[0], w
        assert abs(float(dev.sum_logprob[w]) - host[w].cum_logprobs[0]) < 2e-2
    # batch invariance: window 1 alone (slot map) == window 1 inside the batch of three
    g3 = model.decode_greedy(P, o)
    g1 = model.decode_sample(P[:1], o, temperature=0.0, best_of=1, slots=[1])
    assert g1.tokens[0].tolist() == g3.tokens[1].tolist()
    # alignment: full-sequence pass vs token-by-token pass
    sot_seq = prompt[:3]
    rows = [[*sot_seq, toks.no_timestamps, *[int(t) for t in g3.tokens[w, : g3.n_tokens[w]] if t < toks.eot], toks.eot] for w in range(3)]
    heads = [(7, 0), (10, 17), (12, 18), (13, 12), (16, 1), (17, 14), (19, 11), (21, 4), (24, 1), (25, 6)]
    frames = [3000, 1100, 3000]
    n_rows = max(len(r) for r in rows) - 4
    a = model.align(rows, 4, heads, frames)
    ma = model.last_align_matrix(3, n_rows, 1500)
    a2 = model.align(rows, 4, heads, frames)
    hipbind.tune("align_prefill", 0)
    b = model.align(rows, 4, heads, frames)
    mb = model.last_align_matrix(3, n_rows, 1500)
    b2 = model.align(rows, 4, heads, frames)
    hipbind.tune("align_prefill", 1)
    worst, worst_m, worst_x = 0, 0.0, 0.0
    for w, ((ti_a, fi_a, p_a), (ti_b, fi_b, p_b), row, nf) in enumerate(zip(a, b, rows, frames)):
        n_text = len(row) - 5
        assert ti_a[-1] == n_text and fi_a[-1] == nf // 2 - 1 and ti_b[-1] == n_text
        for x, y in ((a[w], a2[w]), (b[w], b2[w])):        # each pass reproduces itself bit for bit
            assert all(np.array_equal(u, v) for u, v in zip(x, y))
        first = lambda t, f: np.array([f[np.argmax(t == k)] for k in range(n_text + 1)])   # noqa: E731
        worst = max(worst, int(np.abs(first(ti_a, fi_a) - first(ti_b, fi_b)).max()))
        assert np.abs(p_a - p_b).max() < 5e-3
        wa, wb = ma[w, : n_text + 1, : nf // 2], mb[w, : n_text + 1, : nf // 2]
        worst_m = max(worst_m, float(np.abs(wa - wb).max()))
        worst_x = max(worst_x, _path_excess(wa, (ti_b, fi_b), (ti_a, fi_a)), _path_excess(wb, (ti_a, fi_a), (ti_b, fi_b)))
    _diag("large_v3_consistency", {"dtype": dtype, "align_shift_frames_prefill_vs_steps": worst, "max_matrix_diff": worst_m,
                                   "path_excess_per_cell_on_the_other_matrix": worst_x})
    if dtype == "float16":
        assert worst == 0, worst
    assert worst_m < (0.02 if dtype == "float16" else 0.15), worst_m         # measured 5.7e-3 / 4.4e-2
    assert worst_x < (1e-6 if dtype == "float16" else 1e-3), worst_x         # measured 0 / 1.3e-4
    model.close()


def test_language_detection_matches_oracle(hip):
    """detect_language: one decoder step on <|sot|>, softmax over the language tokens -- probabilities against the
    fp32 oracle (1e-5) and the winner through the shim (language=None)."""
    from whisperjav_amd import dims as pdims, synth, weights as pweights, whisper_model as wm
    d, oracle, model = _engine_and_oracle("float32", max_batch=3)
    mel = torch.from_numpy(helpers.synth_mel(3, d.n_mels, seed=51))
    model.encode(mel.cuda())
    got = model.language_probs(3)
    t = model.tokens
    with torch.no_grad():
        lg = oracle.decoder_logits(torch.full((3, 1), t.sot), oracle.encode(mel))[:, 0]
    ref = torch.softmax(lg[:, t.sot + 1: t.sot + 1 + t.num_languages].double(), dim=-1).numpy()
    assert got.shape == ref.shape == (3, 99) and np.abs(got - ref).max() < 1e-5
    assert np.array_equal(got.argmax(1), ref.argmax(1))
    model.close()
    w = pweights.synth_weights(d, seed=21)
    shim = wm.HipWhisperModel("tiny", compute_type="float32", weights=w, dims=d, max_batch=2, max_beam=1)
    audio = synth.speech_like(5.0, seed=8)
    segs, info = shim.transcribe(audio, language=None, beam_size=1, temperature=0.0, max_new_tokens=8, word_timestamps=False,
                                 no_speech_threshold=None, language_detection_threshold=0.0)
    segs = list(segs)
    assert info.language in pdims.LANGUAGE_CODES and info.all_language_probs[0][0] == info.language
    assert abs(sum(p for _, p in info.all_language_probs) - 1.0) < 1e-4
    shim.close()


@pytest.mark.parametrize("mode", ["balanced", "fidelity"])
def test_sharded_transcribe_cli_single_rank(hip, tmp_path, mode):
    """The cfg4 driver end to end on one rank: Hugging Face checkpoint directory (written by transformers, random
    weights) -> blob -> scenes -> VAD groups -> batched beam search with word timestamps -> SRT.  ``--mode fidelity`` is
    BASELINE cfg4 as written: the openai-whisper contract (HipWhisperProASR over HipOpenAIWhisperModel, device-resident
    BeamSearchDecoder search)."""
    import subprocess
    import sys
    import wave
    from transformers import WhisperConfig, WhisperForConditionalGeneration
    from whisperjav_amd import synth
    torch.manual_seed(5)
    cfg = WhisperConfig(vocab_size=51865, num_mel_bins=80, d_model=128, encoder_layers=2, decoder_layers=2,
                        encoder_attention_heads=2, decoder_attention_heads=2, encoder_ffn_dim=512, decoder_ffn_dim=512,
                        max_source_positions=1500, max_target_positions=448)
    WhisperForConditionalGeneration(cfg).save_pretrained(tmp_path / "model", safe_serialization=True)
    audio = synth.speech_like(40.0, seed=12)
    wav = tmp_path / "rec.wav"
    with wave.open(str(wav), "wb") as wf:
        wf.setnchannels(1); wf.setsampwidth(2); wf.setframerate(16000)
        wf.writeframes((np.clip(audio, -1, 1) * 32767).astype("<i2").tobytes())
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, "-m", "whisperjav_amd.sharded_transcribe", str(wav), str(tmp_path / "out" / "rec.srt"),
                          "--model", str(tmp_path / "model"), "--compute-type", "float32", "--batch", "8", "--beam-size", "2",
                          "--max-new-tokens", "12", "--scene-energy-db", "52", "--vad-weights", "synthetic", "--mode", mode,
                          "--logprob-threshold", "-30"],      # random weights: the fidelity post-model gate (-1.0) would drop every segment
                         cwd=root, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    text = (tmp_path / "out" / "rec.srt").read_text(encoding="utf-8")
    assert text.count("-->") >= 2 and text.startswith("1\n")


def test_shim_opens_a_ctranslate2_model_directory(hip, tmp_path):
    """``HipWhisperModel("<dir with model.bin>")`` = what ``faster_whisper.WhisperModel(model_size_or_path=...)`` opens
    (faster_whisper_pro_asr.py:246-253): the directory is written here in CTranslate2's format (float16 and int8 storage,
    whisperjav_amd/ct2_format.py) from seeded EOT-bearing weights; the float16 one must transcribe exactly as the same weights
    handed over directly, the int8 one must load and run (its weights are the dequantised ones), config.json's alignment heads
    arrive."""
    from whisperjav_amd import ct2_format, synth, weights as pweights, whisper_model as wm
    d = helpers.small_dims()
    w = pweights.synth_weights(d, seed=33, **pweights.SPEECHLIKE)
    w = {k: a.astype(np.float16).astype(np.float32) for k, a in w.items()}         # what a float16 conversion stores
    ct2_format.write_ct2_whisper(str(tmp_path / "f16"), d, w, dtype="float16", alignment_heads=[(1, 0), (1, 1)])
    ct2_format.write_ct2_whisper(str(tmp_path / "i8"), d, w, dtype="float16", quantization="int8")
    clips = [synth.speech_like(sec, seed=60 + i) for i, sec in enumerate((1.5, 4.0, 7.0))]
    kw = dict(task="transcribe", language="ja", beam_size=5, patience=1.2, repetition_penalty=1.5, no_repeat_ngram_size=3,
              temperature=0.0, condition_on_previous_text=False, max_new_tokens=48, no_speech_threshold=None,
              max_initial_timestamp=0.0, word_timestamps=False, log_prob_threshold=None, compression_ratio_threshold=None)
    key = lambda per_clip: [[(s.seek, tuple(s.tokens), round(s.avg_logprob, 5)) for s in segs] for segs in per_clip]     # noqa: E731
    direct = wm.HipWhisperModel("tiny", compute_type="float32", weights=w, dims=d, max_batch=4, max_beam=5)
    want = key(direct.transcribe_many(clips, **kw)[0])
    direct.close()
    opened = wm.HipWhisperModel(str(tmp_path / "f16"), compute_type="float32", max_batch=4, max_beam=5)
    assert opened.dims == d and opened._alignment_heads == [(1, 0), (1, 1)]
    got = key(opened.transcribe_many(clips, **kw)[0])
    opened.close()
    assert got == want and any(len(c) for c in got)
    q = wm.HipWhisperModel(str(tmp_path / "i8"), compute_type="int8_float16", max_batch=4, max_beam=5)
    assert q.dims == d and q.compute_type == "float16"
    res = q.transcribe_many(clips, **kw)[0]
    q.close()
    assert len(res) == 3


# ---------------------------------------------------------------------------------------------
# cross-scene pooling behind the seams (round 2): pooled == per scene, and the priming path of the reference's loop
# ---------------------------------------------------------------------------------------------
def test_pooled_scenes_equal_per_scene_and_priming_serves_the_loop(hip, tmp_path):
    """``RecordingTranscriber`` (steps 2-4 of BalancedPipeline.process) with all scenes pooled into ONE segmentation
    launch + ONE engine call gives exactly the segments of the reference's call pattern (one call per scene); and the
    reference's loop shape -- ``prime_scenes`` then ``transcribe_to_srt`` scene by scene -- is served from one pooled
    pass and writes the same per-scene SRT files."""
    import wave
    from whisperjav_amd import asr, pipeline, scenes as scn, segmenters, sharded_transcribe, synth, weights as pweights, whisper_model as wm
    d = helpers.small_dims()
    audio = synth.speech_like(150.0, seed=9, noisy=True)
    model = wm.HipWhisperModel("tiny", compute_type="float32", weights=pweights.synth_weights(d, seed=21), dims=d,
                               max_batch=16, max_beam=2)
    params = sharded_transcribe.balanced_params("ja", 2, 12, word_timestamps=False)
    seg = segmenters.HipSileroV6SpeechSegmenter(weights="synthetic", **params["vad"])
    module = asr.HipFasterWhisperProASR({"model_name": "tiny"}, params, "transcribe", whisper_model=model, segmenter=seg)
    calls = []
    real = model.transcribe_many
    model.transcribe_many = lambda clips, **kw: (calls.append(len(clips)) or real(clips, **kw))
    det = scn.HipAuditokSceneDetector(pass1_energy_threshold=52, pass2_energy_threshold=56)
    runner = pipeline.RecordingTranscriber(module, det)
    pooled = runner.transcribe(audio, 16000, pooled=True)
    n_pooled_calls = len(calls)
    single = runner.transcribe(audio, 16000, pooled=False)
    assert len(pooled["scenes"]) >= 4 and pooled["scenes"] == single["scenes"]
    assert n_pooled_calls == 1 and len(calls) - 1 >= len([r for r in single["per_scene"] if r["segments"]])
    key = lambda segs: [(round(x["start"], 3), round(x["end"], 3), x["text"], round(x["avg_logprob"], 4)) for x in segs]   # noqa: E731
    assert key(pooled["segments"]) == key(single["segments"]) and len(pooled["segments"]) >= 4
    for a, b in zip(pooled["per_scene"], single["per_scene"]):
        assert key(a["segments"]) == key(b["segments"])
    # clips as views of the recording resident in HBM (default) == numpy clips on the host (the reference's contract)
    host = pipeline.RecordingTranscriber(module, det, device_resident=False).transcribe(audio, 16000, pooled=True)
    assert key(host["segments"]) == key(pooled["segments"]) and host["vad_segments"] == pooled["vad_segments"]
    # the reference's loop: scene files on disk, announced once, then one transcribe_to_srt per scene
    paths = []
    for i, sc in enumerate(pooled["scenes"]):
        path = tmp_path / f"rec_scene_{i:04d}.wav"
        chunk = audio[int(sc[0] * 16000): int(sc[1] * 16000)]
        with wave.open(str(path), "wb") as wf:
            wf.setnchannels(1); wf.setsampwidth(2); wf.setframerate(16000)
            wf.writeframes(pipeline.pcm16_encode(chunk).astype("<i2").tobytes())
        paths.append(path)
    del calls[:]
    module.prime_scenes(paths)
    srt_primed = [module.transcribe_to_srt(p, tmp_path / "primed" / (p.stem + ".srt")).read_text(encoding="utf-8") for p in paths]
    vad_primed = module.get_last_vad_segments()
    assert len(calls) == 1 and module.pooled_calls == 1          # one pooled engine call served every scene
    del calls[:]
    srt_loop = [module.transcribe_to_srt(p, tmp_path / "loop" / (p.stem + ".srt")).read_text(encoding="utf-8") for p in paths]
    assert srt_primed == srt_loop and any("-->" in t for t in srt_loop)
    assert vad_primed == module.get_last_vad_segments()
    _diag("pooled_scenes", {"scenes": len(paths), "segments": len(pooled["segments"]), "per_scene_engine_calls": len(calls)})
    module.cleanup()


def test_cabi_weight_broadcast_single_rank(hip):
    """``wj_comm_*`` / ``wj_bcast_weights`` (RCCL resolved at run time) with a one-rank communicator: the blob comes back
    unchanged.  More ranks need more GPUs than this environment leases; the call sequence is the same."""
    import ctypes as C
    from whisperjav_amd import hipbind, sharding
    blob = torch.arange(1 << 20, dtype=torch.int32).view(torch.uint8)
    offsets = np.arange(0, blob.numel(), 256, dtype=np.int64)
    out, offs = sharding.broadcast_blob_cabi(blob, offsets, torch.device("cuda", 0))
    assert out.is_cuda and torch.equal(out.cpu(), blob) and np.array_equal(offs, offsets)
    uid = C.create_string_buffer(128)
    hipbind.check(hip.wj_comm_unique_id(uid), "wj_comm_unique_id")
    assert any(b != 0 for b in uid.raw)
    comm = C.c_void_p()
    assert hip.wj_comm_init(hipbind.context(0).handle, 2, 5, uid.raw, C.byref(comm)) != 0          # rank outside the communicator


_BCAST_WORKER = """
import os, sys
import numpy as np, torch
sys.path.insert(0, os.environ["WJ_REPO"])
from whisperjav_amd import sharding
info = sharding.init_distributed()
dev = torch.device("cuda", info.local_rank)
torch.cuda.set_device(dev)
blob = offsets = None
if info.rank == 0:
    blob = torch.arange(1 << 22, dtype=torch.int32).view(torch.uint8)
    offsets = np.arange(0, blob.numel(), 4096, dtype=np.int64)
out, offs = sharding.broadcast_blob_cabi(blob, offsets, dev)
want = torch.arange(1 << 22, dtype=torch.int32).view(torch.uint8)
ok = bool(torch.equal(out.cpu(), want)) and len(offs) == want.numel() // 4096
print(f"rank {info.rank} ok={ok} comm_ranks={sharding.LAST_COMM_RANKS}", flush=True)
torch.distributed.destroy_process_group()
sys.exit(0 if ok else 1)
"""


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="the multi-rank C-ABI broadcast needs two GPUs (the gpurun boxes lease one)")
def test_cabi_weight_broadcast_two_ranks(hip, tmp_path):
    """Two processes, one GPU each: the communicator id travels through the process group as an object, ``wj_comm_init`` /
    ``wj_bcast_weights`` move the blob over RCCL on the library's own stream (this path was single-rank only)."""
    import os
    import subprocess
    import sys
    worker = tmp_path / "bcast_worker.py"
    worker.write_text(_BCAST_WORKER)
    env = dict(os.environ, WJ_REPO=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), HSA_ENABLE_IPC_MODE_LEGACY="0")
    run = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29517", str(worker)], env=env, capture_output=True, text=True, timeout=600)
    # With two GPUs visible (the skip above) everything short of both ranks reporting success is a FAILURE -- a crash in
    # wj_comm_init, a hang, a rendezvous problem print nothing and used to read as "x" 
    tail = run.stdout[-2000:] + run.stderr[-2000:]
    assert run.returncode == 0, tail
    assert "ok=False" not in run.stdout, tail
    for r in (0, 1):
        assert f"rank {r} ok=True comm_ranks=2" in run.stdout, tail             # RCCL itself reports a 2-rank communicator


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (the gpurun boxes lease one)")
def test_bench_two_gpus_strong_scaling_equals_one_gpu(hip):
    """``bench.py --gpus 2 --strong`` (one recording, scenes LPT-sharded over two ranks, RCCL weight broadcast) must report
    n_gpus = 2 and the SAME transcript as the 1-GPU run of the same recording (merged CRC), on a 10-minute recording."""
    import json as _json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", WJ_BCAST_CABI="1", WJ_BENCH_SEGMENT_HASHES="1")
    common = ["--steps", "1", "--warmup", "0", "--minutes", "10", "--no-extras", "--no-cpu-baseline", "--no-profile",
              "--tune", "batch_invariant=1"]       # one kernel family whatever the row count: a window's result does not depend on its shard
    lines = {}
    for n in (1, 2):
        run = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), "--strong", *common], env=env,
                             capture_output=True, text=True, timeout=1500, cwd=root)
        assert run.returncode == 0, run.stdout[-1500:] + run.stderr[-1500:]
        lines[n] = _json.loads([l for l in run.stdout.splitlines() if l.startswith("{")][-1])
    assert lines[2]["n_gpus"] == 2 and lines[2]["scaling"] == "strong"
    ranks1, ranks2 = lines[1]["config"]["per_rank_last_step"], lines[2]["config"]["per_rank_last_step"]
    assert len(ranks2) == 2 and all(r["scenes"] > 0 for r in ranks2)
    assert sum(r["scenes"] for r in ranks2) == ranks1[0]["scenes"]                        # every scene transcribed exactly once
    one = sorted(ranks1[0]["segment_hashes"])
    two = sorted(h for r in ranks2 for h in r["segment_hashes"])
    assert one == two                                                                     # the SAME transcript, segment for segment
    assert sum(r["segment_digest"] for r in ranks2) % (1 << 32) == ranks1[0]["segment_digest"]


def test_batch_invariant_mode_makes_a_window_independent_of_its_batch(hip):
    """``wj_tune("batch_invariant", 1)`` (one GEMM kernel family, fixed split-K factors whatever the row count): the same 22 clips
    transcribed (a) in one pool through an engine of 16 resident windows and (b) as two shards in another order through an engine
    of 4 (20 decode rows: the default mode's one-wave-per-row kernels; 80: its skinny / tile kernels) give BIT-IDENTICAL tokens,
    log-probs and timestamps per clip, float16, beam 5, searches that end, word timestamps on.  This is what lets a multi-GPU run
    reproduce the single-GPU transcript exactly (test_bench_two_gpus_strong_scaling_equals_one_gpu runs in this mode); the
    default mode picks the fastest kernel per row count and is recorded beside it."""
    from whisperjav_amd import hipbind, synth, weights as pweights, whisper_model as wm
    d = helpers.small_dims()
    w = pweights.synth_weights(d, seed=33, exact="float16", **pweights.SPEECHLIKE)
    clips = [synth.speech_like(0.8 + 0.45 * i, seed=300 + i) for i in range(22)]
    kw = dict(task="transcribe", language="ja", beam_size=5, patience=1.2, repetition_penalty=1.5, no_repeat_ngram_size=3,
              temperature=0.0, condition_on_previous_text=False, max_new_tokens=48, no_speech_threshold=None,
              max_initial_timestamp=0.0, word_timestamps=True, log_prob_threshold=None, compression_ratio_threshold=None)
    key = lambda segs: [(s.seek, s.start, s.end, tuple(s.tokens), float(s.avg_logprob),     # noqa: E731
                         tuple((x.start, x.end, float(x.probability)) for x in (s.words or []))) for s in segs]
    order = list(range(1, 22, 2)) + list(range(0, 22, 2))                  # shard 1 then shard 0 of an alternating plan

    def run(mode, max_batch, idx_lists):
        hipbind.tune("batch_invariant", mode)
        try:
            model = wm.HipWhisperModel("tiny", compute_type="float16", weights=w, dims=d, max_batch=max_batch, max_beam=5, kv_len=3 + 48 + 5)
            model.word_reseek = False
            out = {}
            for idx in idx_lists:
                per_clip, _ = model.transcribe_many([clips[i] for i in idx], **kw)
                for i, segs in zip(idx, per_clip):
                    out[i] = key(segs)
            model.close()
            return out
        finally:
            hipbind.tune("batch_invariant", 0)

    pooled = run(1, 16, [list(range(22))])
    sharded = run(1, 4, [order[:11], order[11:]])
    assert sum(len(v) for v in pooled.values()) >= 22
    diff = [i for i in range(22) if pooled[i] != sharded[i]]
    fast_pooled, fast_sharded = run(0, 16, [list(range(22))]), run(0, 4, [order[:11], order[11:]])
    same_tokens = sum([t[3] for t in fast_pooled[i]] == [t[3] for t in fast_sharded[i]] for i in range(22))
    _diag("batch_invariant", {"clips": 22, "invariant_mode_clips_that_differ": diff, "default_mode_clips_with_identical_tokens": same_tokens,
                              "default_mode_bit_identical_clips": sum(fast_pooled[i] == fast_sharded[i] for i in range(22)),
                              "invariant_equals_default_tokens": sum([t[3] for t in pooled[i]] == [t[3] for t in fast_pooled[i]] for i in range(22))})
    assert not diff, diff
    assert same_tokens >= 20            # default mode: near-tie flips only


@pytest.mark.parametrize("flavour", ["fw", "ow"])
def test_encoder_of_the_next_chunk_overlaps_the_decode_of_this_one(hip, flavour):
    """``transcribe_many`` encodes chunk i + 1 on a second stream into the other half of the resident window slots while
    chunk i decodes (``wj_whisper_encode_at``): the segments must be the ones the sequential schedule produces, for the
    CTranslate2-flavoured model (beam 5) and the openai-flavoured one (beam 2), on searches that end."""
    from whisperjav_amd import synth, weights as pweights, whisper_model as wm
    d = helpers.small_dims()
    w = pweights.synth_weights(d, seed=33, **pweights.SPEECHLIKE)
    cls = wm.HipWhisperModel if flavour == "fw" else wm.HipOpenAIWhisperModel
    model = cls("tiny", compute_type="float32", weights=w, dims=d, max_batch=4, max_beam=5, kv_len=3 + 48 + 5)
    clips = [synth.speech_like(0.7 + 0.55 * i, seed=200 + i) for i in range(11)]
    if flavour == "fw":
        kw = dict(task="transcribe", language="ja", beam_size=5, patience=1.2, repetition_penalty=1.5, no_repeat_ngram_size=3,
                  temperature=0.0, condition_on_previous_text=False, max_new_tokens=48, no_speech_threshold=None,
                  max_initial_timestamp=0.0, word_timestamps=False, log_prob_threshold=None, compression_ratio_threshold=None)
    else:
        kw = dict(task="transcribe", language="ja", beam_size=2, patience=1.2, temperature=0.0, condition_on_previous_text=False,
                  sample_len=48, no_speech_threshold=None, logprob_threshold=None, compression_ratio_threshold=None, fp16=False)
    key = lambda per_clip: [[(s.seek, tuple(s.tokens), round(s.avg_logprob, 4)) for s in segs] for segs in per_clip]     # noqa: E731
    model.overlap_encode = False
    if flavour == "fw":
        seq, _ = model.transcribe_many(clips, **kw)
    else:
        seq = [model.transcribe(c, **kw)["segments"] for c in clips]
    model.overlap_encode = True          # an opt-in schedule (measured: no gain on MI355X), kept correct
    if flavour == "fw":
        ovl, _ = model.transcribe_many(clips, **kw)
        assert key(ovl) == key(seq)
        assert any(len(s) for s in seq) and len({sum(len(x.tokens) for x in s) for s in seq}) > 2      # ragged, non-trivial
    else:
        many, _ = model.transcribe_many(clips, **{k: v for k, v in kw.items() if k != "fp16"})
        got = [[(s.seek, tuple(s.tokens)) for s in segs] for segs in many]
        ref = [[(s["seek"], tuple(s["tokens"])) for s in segs] for segs in seq]
        assert got == ref
    assert model._overlapped                                             # the overlapped path ran
    # ... and with the pair on disjoint compute units (CU-masked streams)
    model.encoder_cus = 160
    if flavour == "fw":
        cus, _ = model.transcribe_many(clips, **kw)
        assert key(cus) == key(seq) and model.model._split is not None
    model.close()
