"""Decode-step tuning sweep on one MI355X: one model load, many (batch, split-K, kernel-choice) settings.
Times `tokens` greedy steps per setting with wall clock around the synchronous C-ABI call."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whisperjav_amd import dims as pdims, engine, hipbind, weights as pweights

TOK = int(os.environ.get("SWEEP_TOKENS", "48"))
dims = pdims.dims_for("large-v3")
w = pweights.synth_weights(dims, seed=1234)
BMAX = int(os.environ.get('SWEEP_BMAX', '384'))
model = engine.HipWhisper(dims, w, dtype="bfloat16", max_batch=BMAX)
del w
g = torch.Generator(device="cuda").manual_seed(1)
mel = (torch.randn((BMAX, dims.n_mels, 3000), device="cuda", generator=g) * 0.4).clamp(-1, 1.5)
model.encode(mel)
toks = model.tokens
suppress = (toks.sot, toks.translate, toks.transcribe, toks.sot_lm, toks.sot_prev, toks.no_speech)
opts = engine.DecodeOptions(max_new_tokens=TOK, suppress_tokens=suppress, max_initial_timestamp=1.0)
rows = []
# (batch, dec_tile_reg)
configs = [(384, 0), (384, 1), (128, 0), (128, 1), (256, 0), (256, 1)]
ref_tokens = {}
for B, tr in configs:
    hipbind.tune("dec_tile_reg", tr)
    prompt = np.tile(np.array(model.sot_prompt("ja"), dtype=np.int32), (B, 1))
    model.decode_greedy(prompt, engine.DecodeOptions(max_new_tokens=4, suppress_tokens=suppress))   # warm
    best = 1e9
    for _ in range(2):
        t0 = time.perf_counter(); res = model.decode_greedy(prompt, opts); best = min(best, time.perf_counter() - t0)
    same = None
    if B in ref_tokens:
        same = float((res.tokens == ref_tokens[B]).mean())
    else:
        ref_tokens[B] = res.tokens.copy()
    rows.append({"B": B, "tile_reg": tr, "ms_per_step": round(1e3 * best / (TOK + 2), 3), "token_agreement_vs_first_config": same})
    print(rows[-1], flush=True)
hipbind.tune("dec_tile_reg", 0)
json.dump(rows, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "decode_sweep.json"), "w"), indent=1)
