"""Encoder attention kernel variants (bit0 XCD remap, bit1 base-2 softmax on v_exp_f32, bit2 lazy rescale),
HIP-event timed on the engine stream; B windows x 20 heads x 1500 positions, random bf16 inputs."""
import ctypes as C, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whisperjav_amd import hipbind
lib, ctx = hipbind.lib(), hipbind.context(0)
rows = []
for B in (64, 128):
    H, T = 20, 1500
    g = torch.Generator(device="cuda").manual_seed(B)
    qkv = torch.randn((B, T, 3 * H * 64), device="cuda", generator=g)
    out = torch.empty((B, T, H * 64), dtype=torch.bfloat16, device="cuda")
    torch.cuda.synchronize()
    flops = 4.0 * B * H * T * T * 64
    ref_out = None
    for var in (7, 9, 25):
        hipbind.tune("attn_enc_variant", var)
        best = 1e9
        for _ in range(3):
            ms = C.c_float()
            hipbind.check(lib.wj_k_attention_enc_timed(ctx.handle, 1, C.c_void_p(qkv.data_ptr()), C.c_void_p(out.data_ptr()),
                                                       B, T, H, 10, C.byref(ms)))
            best = min(best, ms.value)
        if var == 7:
            ref_out = out.float().clone()
        rows.append({"B": B, "variant": var, "ms": round(best, 4), "TFLOPs": round(flops / best / 1e9, 1),
                     "max_abs_diff_vs_7": float((out.float() - ref_out).abs().max())})
        print(rows[-1], flush=True)
json.dump(rows, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "attn_sweep.json"), "w"), indent=1)
