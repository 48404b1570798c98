"""Prompt attention (csrc/qwen.hip prompt_attn_kernel) timing on a small decoder so that attention dominates the pass: 1024
prompts of 200 tokens, 3 layers of hidden 256 / 4 heads; prints ms per prefill with the MFMA tiles and with the row kernel.
Run once per library build (WJHIP_LIB=...) to compare register allocations."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from whisperjav_amd import hipbind, qwen  # noqa: E402

d = qwen.Qwen3Dims(hidden=256, n_layer=3, n_head=4, n_kv_head=2, head_dim=128, ffn=640, vocab=4096, rope_theta=10000.0,
                   audio_token_id=9, eos_token_ids=(1, 2))
B, T = 1024, 200
model = qwen.HipQwen3Decoder(d, qwen.synth_weights(d, seed=3), dtype="float16", max_seqs=B, max_ctx=256, max_rows=B * T)
packed = torch.randn(B * T, d.hidden, device="cuda") * 0.5
n = np.full(B, T, dtype=np.int32)
out = {"lib": os.path.basename(os.environ.get("WJHIP_LIB", "libwjhip.so"))}
for mode in (1, 0):
    hipbind.tune("qwen_prompt_mfma", mode)
    model.prefill_packed(packed, n)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(4):
        model.prefill_packed(packed, n)
    torch.cuda.synchronize()
    out[f"mfma_{mode}_ms"] = round((time.perf_counter() - t0) * 250.0, 2)
hipbind.tune("qwen_prompt_mfma", 1)
print(json.dumps(out), flush=True)
