"""Micro-benchmark of the bf16 GEMM tile kernel variants on the encoder's shapes (MI355X only).
HIP-event timed on the engine stream, uniform random operands (not zeros: DVFS)."""
import json
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whisperjav_amd import engine, hipbind

SHAPES = [("enc_qk", 96000, 2560, 1280), ("enc_out", 96000, 1280, 1280), ("enc_fc1", 96000, 5120, 1280),
          ("enc_fc2", 96000, 1280, 5120), ("cube4096", 4096, 4096, 4096), ("cube8192", 8192, 8192, 8192)]
rows = []
for name, M, N, K in SHAPES:
    for variant, label in ((6, "big256"), (6, "big256_sched")):
        hipbind.tune("gemm_big", 2 if label.endswith("sched") else 1)
        if label == "skinny" and M > 512:
            continue
        if label != "skinny" and M <= 512:
            continue
        ms = min(engine.k_gemm_timed(M, N, K, "bfloat16", variant, reps=20) for _ in range(4))
        tf = 2.0 * M * N * K / ms / 1e9
        gbs = (N * K * 2) / ms / 1e6
        rows.append({"shape": name, "M": M, "N": N, "K": K, "variant": label, "ms": round(ms, 4), "TFLOPs": round(tf, 1),
                     "weight_GBs": round(gbs, 1)})
        print(rows[-1], flush=True)
hipbind.tune('gemm_big', 1)
json.dump(rows, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "gemm_sweep.json"), "w"), indent=1)
