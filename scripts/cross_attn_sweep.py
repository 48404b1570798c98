"""Stand-alone timing of the decode cross-attention kernels at production shape (gpurun).
Compares the matrix-core kernel (V transposed) with the vector kernel (row-major V) and sweeps the
loads-in-flight variants; prints achieved GB/s against the algorithmic K+V bytes."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whisperjav_amd import engine, hipbind  # noqa: E402

out = []
H, n_keys = 20, 1500
for G, nb in [(384, 1), (128, 1), (128, 5)]:
    g = torch.Generator().manual_seed(1)
    q = torch.randn(G, nb, H * 64, generator=g).cuda()
    k = torch.randn(G, H, n_keys, 64, generator=g).cuda()
    v = torch.randn(G, H, n_keys, 64, generator=g).cuda()
    bytes_ = 2.0 * G * H * n_keys * 64 * 2
    ref, ms_old = engine.k_attention_dec_timed(q, k, v, "bfloat16", layout=1, reps=30)
    row = {"G": G, "nb": nb, "valu_ms": ms_old, "valu_GBs": bytes_ / ms_old / 1e6}
    for nt in (0, 1):
        hipbind.tune("dec_cross_nt", nt)
        for u in (0, 1, 2, 3):
            hipbind.tune("dec_cross_u", u)
            got, ms = engine.k_attention_dec_timed(q, k, v, "bfloat16", layout=0, reps=30)
            row[f"mfma_nt{nt}_u{u}_GBs"] = round(bytes_ / ms / 1e6, 1)
            row[f"mfma_nt{nt}_u{u}_maxdiff"] = float((got - ref).abs().max())
    hipbind.tune("dec_cross_u", 0); hipbind.tune("dec_cross_nt", 0)
    print(json.dumps(row), flush=True)
    out.append(row)
    del q, k, v
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/cross_attn_sweep.json", "w"), indent=1)
