"""Decode-step GEMMs of the Qwen3-ASR-1.7B decoder (hidden 2048, ffn 6144, 16 q + 8 kv heads of 128) at the row counts a
batched greedy step has: which kernel of csrc/gemm.hip is fastest per shape.  Variants: 0 = dispatcher default (skinny up to
512 rows, 256-tile ping-pong from 1024), 2 = skinny, 3 = 128-tile LDS-DMA, 73..75 = 128-tile multi-stage, 86 = 256-tile pp64.
    gpurun -- 'bash scripts/gpu_run.sh py scripts/qwen_gemm_sweep.py'  ->  gpurun_out/qwen_gemm_sweep.log (JSON lines)"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from whisperjav_amd import engine, hipbind

SHAPES = {"qkv": (4096, 2048), "o": (2048, 2048), "gate_up": (12288, 2048), "down": (2048, 6144)}
ROWS = [int(a) for a in sys.argv[1:]] or [64, 256, 512, 1024, 2048]
VARIANTS = [0, 2, 3, 73, 74, 75, 86]

for M in ROWS:
    for name, (N, K) in SHAPES.items():
        row = {"M": M, "gemm": name, "N": N, "K": K}
        for v in VARIANTS:
            try:
                ms = engine.k_gemm_timed(M, N, K, "float16", variant=v, reps=20)
                row[f"v{v}_us"] = round(1e3 * ms, 1)
            except hipbind.WjError:
                row[f"v{v}_us"] = None
        best = min((t, k) for k, t in row.items() if k.endswith("_us") and t is not None)
        row["best"] = best[1]
        row["best_tflops"] = round(2.0 * M * N * K / (best[0] * 1e-6) / 1e12, 1)
        row["best_weight_gbs"] = round(2.0 * N * K / (best[0] * 1e-6) / 1e9, 1)
        print(json.dumps(row), flush=True)
