"""CPU study (oracle only, no GPU): how far do 16-bit operand types move Whisper's token log-probs?

The oracle (oracle/whisper_ref.py) is run three times on the large-v3 geometry with seeded weights -- plain fp32,
and with every GEMM / attention operand rounded to fp16 resp. bf16 at the engine's rounding points (fp32
accumulation, fp32 residual stream, LayerNorm, softmax and logits, exactly the split libwjhip's 16-bit compute
types use) -- and the teacher-forced log-probs of the fp32 greedy tokens are compared.  Written to
profiles/r02_precision_study_cpu.json.  Usage: python scripts/precision_study.py [n_tokens]
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import decoding, logmel, whisper_ref  # noqa: E402
from tests import helpers  # noqa: E402
from whisperjav_amd import dims as pdims, synth, weights as pweights  # noqa: E402


def f16_round(x):
    return x.to(torch.float16).to(torch.float32)


def main():
    n_new = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    dims = pdims.dims_for("large-v3")
    audio = synth.speech_like(30.0, seed=1234)
    mel = torch.from_numpy(logmel.window_features(audio, 128, "fw")[None])
    toks = pdims.special_tokens(dims.n_vocab)
    prompt = [toks.sot, toks.language_token(pdims.language_index("ja")), toks.transcribe]
    suppress = (1, 2, 7, 8, 9, 10, 14, 25, toks.sot, toks.translate, toks.transcribe, toks.sot_lm, toks.sot_prev, toks.no_speech)
    cfg = decoding.FilterConfig(suppress_tokens=suppress, max_initial_timestamp_index=50)
    out = {"what": __doc__.split("\n\n")[0], "n_tokens": n_new, "cases": []}
    for exact in ("none", "float16"):
        w = pweights.synth_weights(dims, seed=1234, exact=exact)
        ref = whisper_ref.WhisperOracle(helpers.oracle_dims(dims), w)
        with torch.no_grad():
            t0 = time.time()
            enc = ref.encode(mel)
            res = decoding.greedy_decode(ref, enc, prompt, n_new, cfg)
            seq = torch.tensor([prompt + res.tokens[0]], dtype=torch.int64)
            lp_ref = torch.log_softmax(ref.decoder_logits(seq, enc)[0], dim=-1)
            print(f"[{exact}] fp32 reference {time.time() - t0:.0f}s tokens {res.tokens[0][:8]}...", flush=True)
            P = len(prompt)
            idx = seq[0, P:]
            pos = torch.arange(P - 1, P - 1 + len(idx))
            for name, rnd in (("float16", f16_round), ("bfloat16", whisper_ref.bf16_round)):
                t0 = time.time()
                em = whisper_ref.WhisperOracle(helpers.oracle_dims(dims), w, act_round=rnd)
                enc_e = em.encode(mel)
                lp_e = torch.log_softmax(em.decoder_logits(seq, enc_e)[0], dim=-1)
                d_tok = (lp_e[pos, idx] - lp_ref[pos, idx]).abs()
                d_all = (lp_e[pos] - lp_ref[pos]).abs()
                flips = int((lp_e[pos].argmax(-1) != lp_ref[pos].argmax(-1)).sum())
                case = {"weights": exact, "operands": name, "enc_max_abs": float((enc_e - enc).abs().max()),
                        "token_logprob_max_abs": float(d_tok.max()), "token_logprob_mean_abs": float(d_tok.mean()),
                        "any_logprob_max_abs_top100": float(d_all.gather(1, lp_ref[pos].topk(100).indices).max()),
                        "argmax_flips": flips, "seconds": round(time.time() - t0, 1)}
                out["cases"].append(case)
                print(json.dumps(case), flush=True)
    with open(os.path.join(ROOT, "profiles", "r02_precision_study_cpu.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
