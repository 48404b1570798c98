"""Root-cause run for the r04 driver failure of test_large_v3_geometry_beam_and_alignment_consistency (bf16, prefill vs
step-wise alignment 9 frames apart).  Writes one JSON line per experiment to gpurun_out/diag_align.jsonl:

  * determinism: the same align call repeated (both paths), in the failing test's order (beam, greedy, sample, align);
  * a FRESH engine whose first call after encode is align (no decode state before it);
  * ln_vec / enc_blocked toggles (the r04 summation-order changes);
  * float16 beside bfloat16, flat weights beside weights.SPEECHLIKE (peaked cross attention).
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from whisperjav_amd import dims as pdims, engine, hipbind, synth, weights as pweights  # noqa: E402

OUT = os.path.join(ROOT, "gpurun_out", "diag_align.jsonl")
HEADS = [(7, 0), (10, 17), (12, 18), (13, 12), (16, 1), (17, 14), (19, 11), (21, 4), (24, 1), (25, 6)]
FRAMES = [3000, 1100, 3000]


def note(**kw):
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(OUT, "a") as f:
        f.write(json.dumps(kw) + "\n")
    print(json.dumps(kw), flush=True)


def first_frames(ti, fi, n_text):
    return np.array([fi[np.argmax(ti == k)] for k in range(n_text + 1)])


def shift(a, b, rows):
    worst, dp = 0, 0.0
    per = []
    for (ti_a, fi_a, p_a), (ti_b, fi_b, p_b), row in zip(a, b, rows):
        n_text = len(row) - 5
        s = int(np.abs(first_frames(ti_a, fi_a, n_text) - first_frames(ti_b, fi_b, n_text)).max())
        per.append(s)
        worst = max(worst, s)
        dp = max(dp, float(np.abs(p_a - p_b).max()))
    return worst, per, dp


def same(a, b):
    return all(np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1]) and np.array_equal(x[2], y[2]) for x, y in zip(a, b))


def run(dtype, wkind, w, d, mels, clips_tag):
    model = engine.HipWhisper(d, w, dtype=dtype, max_batch=3, max_beam=5)
    model.encode(mels)
    toks = model.tokens
    prompt = model.sot_prompt("ja", "transcribe")
    suppress = (toks.sot, toks.translate, toks.transcribe, toks.sot_lm, toks.sot_prev, toks.no_speech)
    o = engine.DecodeOptions(max_new_tokens=14, suppress_tokens=suppress, max_initial_timestamp=0.0, repetition_penalty=1.5,
                             no_repeat_ngram_size=3)
    P = np.tile(np.array(prompt, dtype=np.int32), (3, 1))
    model.decode_beam(P, o, beam_size=5, patience=1.2, length_penalty=1.0)
    g3 = model.decode_greedy(P, o)
    model.decode_sample(P[:1], o, temperature=0.0, best_of=1, slots=[1])
    rows = [[*prompt[:3], toks.no_timestamps, *[int(t) for t in g3.tokens[w_, : g3.n_tokens[w_]] if t < toks.eot], toks.eot] for w_ in range(3)]

    def both():
        a = model.align(rows, 4, HEADS, FRAMES)
        hipbind.tune("align_prefill", 0)
        b = model.align(rows, 4, HEADS, FRAMES)
        hipbind.tune("align_prefill", 1)
        return a, b

    a0, b0 = both()
    w0, per0, dp0 = shift(a0, b0, rows)
    note(exp="in_test_order", dtype=dtype, weights=wkind, shift=w0, per_window=per0, dprob=dp0, row_lens=[len(r) for r in rows])
    rep_a, rep_b = True, True
    for _ in range(3):
        a, b = both()
        rep_a &= same(a, a0)
        rep_b &= same(b, b0)
    note(exp="repeat3", dtype=dtype, weights=wkind, prefill_bit_identical=bool(rep_a), steps_bit_identical=bool(rep_b))
    for key in ("ln_vec",):
        hipbind.tune(key, 0)
        a, b = both()
        hipbind.tune(key, 1)
        note(exp=f"{key}=0", dtype=dtype, weights=wkind, shift=shift(a, b, rows)[0], prefill_vs_default=shift(a, a0, rows)[0],
             steps_vs_default=shift(b, b0, rows)[0])
    # dec_split_act only matters for fp16; dec_rows (row kernel) vs tile kernels on the step path
    hipbind.tune("dec_rows", 0)
    _, b = both()
    hipbind.tune("dec_rows", 1)
    note(exp="dec_rows=0", dtype=dtype, weights=wkind, steps_vs_default=shift(b, b0, rows)[0], shift=shift(a0, b, rows)[0])
    hipbind.tune("dec_fuse_reduce", 0)
    _, b = both()
    hipbind.tune("dec_fuse_reduce", 1)
    note(exp="dec_fuse_reduce=0", dtype=dtype, weights=wkind, steps_vs_default=shift(b, b0, rows)[0], shift=shift(a0, b, rows)[0])
    model.close()
    # a fresh engine: align is the first call after encode
    model = engine.HipWhisper(d, w, dtype=dtype, max_batch=3, max_beam=5)
    model.encode(mels)
    a = model.align(rows, 4, HEADS, FRAMES)
    hipbind.tune("align_prefill", 0)
    b = model.align(rows, 4, HEADS, FRAMES)
    hipbind.tune("align_prefill", 1)
    note(exp="fresh_engine_align_first", dtype=dtype, weights=wkind, shift=shift(a, b, rows)[0], prefill_same_as_in_test_order=same(a, a0),
         steps_same_as_in_test_order=same(b, b0))
    model.close()
    return rows


def main():
    d = pdims.dims_for("large-v3")
    fe = engine.HipLogMel(128, "fw")
    clips = [synth.speech_like(30.0, seed=1234), synth.speech_like(11.0, seed=77), synth.speech_like(30.0, seed=5)]
    mels = fe(clips)
    for wkind, kw in (("flat", {}), ("speechlike", pweights.SPEECHLIKE)):
        w = pweights.synth_weights(d, seed=1234, **kw)
        for dtype in ("bfloat16", "float16"):
            run(dtype, wkind, w, d, mels, "t")
        del w


if __name__ == "__main__":
    main()
