"""Oracle-only study (CPU): which float16 rounding points of the Qwen3-ASR audio tower carry the error of the audio embeddings
(end to end 1.3-1.55e-3 while the decoder fed exact embeddings is at 0.65-1.06e-3).  The tower of
oracle/qwen3_ref.py is restated with a rounding hook at every point where csrc/qwen_audio.hip stores a 16-bit tensor; each row
of the output = max |embedding error| / max |embedding| with fp16 rounding at that subset of points (published 1.7 B tower
geometry, bf16-representable seeded weights, a 4 s clip).  Writes profiles/r05_precision_qwen_tower_cpu.json."""
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import logmel, qwen3_ref  # noqa: E402
from whisperjav_amd import qwen, synth  # noqa: E402

POINTS = ("mel", "conv_act1", "conv_act2", "conv_act3", "ln_out", "q", "k", "v", "attn_out", "gelu_out", "lnpost_out", "proj1_out")


def f16(x):
    return x.half().float()


def tower(o, mel, on):
    r = lambda name, x: f16(x) if name in on else x           # noqa: E731
    d, w = o.dims, o.w
    p = "model.audio_tower."
    chunk = 2 * d.n_window
    n = mel.shape[1]
    n_chunks = (n + chunk - 1) // chunk
    x = F.pad(r("mel", mel), (0, n_chunks * chunk - n))
    x = x.view(d.n_mels, n_chunks, chunk).permute(1, 0, 2)[:, None]
    for i in (1, 2, 3):
        x = r(f"conv_act{i}", F.gelu(F.conv2d(x, w[f"{p}conv2d{i}.weight"], w[f"{p}conv2d{i}.bias"], stride=2, padding=1)))
    c, ch, fb, ts = x.shape
    x = x.permute(0, 3, 1, 2).reshape(c, ts, ch * fb) @ w[p + "conv_out.weight"].T
    x = x + qwen3_ref.sinusoid_table(d.a_max_pos, d.a_d)[:ts]
    lens = [qwen3_ref.post_cnn_length(min(chunk, n - i * chunk)) for i in range(n_chunks)]
    h = torch.cat([x[i, : lens[i]] for i in range(n_chunks)], dim=0)
    win = max(lens) * (d.n_window_infer // chunk)
    total = h.shape[0]
    bounds = list(range(0, total, win)) + [total]
    hd = d.a_d // d.a_heads
    for l in range(d.a_layers):
        q = f"{p}layers.{l}."
        y = r("ln_out", F.layer_norm(h, (d.a_d,), w[q + "self_attn_layer_norm.weight"], w[q + "self_attn_layer_norm.bias"], 1e-5))
        qs = r("q", y @ w[q + "self_attn.q_proj.weight"].T + w[q + "self_attn.q_proj.bias"]).view(total, d.a_heads, hd)
        ks = r("k", y @ w[q + "self_attn.k_proj.weight"].T + w[q + "self_attn.k_proj.bias"]).view(total, d.a_heads, hd)
        vs = r("v", y @ w[q + "self_attn.v_proj.weight"].T + w[q + "self_attn.v_proj.bias"]).view(total, d.a_heads, hd)
        outs = []
        for a, b in zip(bounds[:-1], bounds[1:]):
            s = torch.einsum("qhd,khd->hqk", qs[a:b], ks[a:b]) * hd ** -0.5
            outs.append(torch.einsum("hqk,khd->qhd", torch.softmax(s, dim=-1), vs[a:b]).reshape(b - a, d.a_d))
        h = h + r("attn_out", torch.cat(outs, 0)) @ w[q + "self_attn.out_proj.weight"].T + w[q + "self_attn.out_proj.bias"]
        y = r("ln_out", F.layer_norm(h, (d.a_d,), w[q + "final_layer_norm.weight"], w[q + "final_layer_norm.bias"], 1e-5))
        y = r("gelu_out", F.gelu(y @ w[q + "fc1.weight"].T + w[q + "fc1.bias"]))
        h = h + y @ w[q + "fc2.weight"].T + w[q + "fc2.bias"]
    h = r("lnpost_out", F.layer_norm(h, (d.a_d,), w[p + "ln_post.weight"], w[p + "ln_post.bias"], 1e-5))
    m = "model.multi_modal_projector."
    h = r("proj1_out", F.gelu(h @ w[m + "linear_1.weight"].T + w[m + "linear_1.bias"]))
    return h @ w[m + "linear_2.weight"].T + w[m + "linear_2.bias"]


def main():
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    ad = qwen.Qwen3AudioDims()
    d = qwen.Qwen3Dims()
    ramp = qwen.QwenEosRamp.for_dims(d)
    w = qwen.synth_audio_weights(ad, seed=2, ramp=ramp)
    w = {k: torch.from_numpy(v).bfloat16().float().numpy() if v.ndim >= 2 else v for k, v in w.items()}
    o = qwen3_ref.Qwen3AsrOracle(qwen3_ref.Qwen3AsrDims(), w)
    mel = torch.from_numpy(logmel.logmel_ow(synth.speech_like(4.0, seed=7), 128, padding=0))
    rows = {}
    with torch.no_grad():
        ref = tower(o, mel, ())
        scale = float(ref.abs().max())
        shipped = ("conv_act1", "conv_act2")          # what round 5's tower still rounds (qwen_tower_split: everything else travels as [hi | lo])
        for name, on in [("all (round 4)", POINTS)] + [(p, (p,)) for p in POINTS] + [("round 5 tower: " + ", ".join(shipped), shipped)] + [
                ("round 5 + " + p, shipped + (p,)) for p in ("q", "k", "v", "mel", "conv_act3")] + [
                ("round 5 + q, k, v (attention inputs plain)", shipped + ("q", "k", "v")), ("round 5 + k, v", shipped + ("k", "v")),
                ("round 5 + v", shipped + ("v",))]:
            rows[name] = float((tower(o, mel, on) - ref).abs().max()) / scale
            print(f"{name:60s} {rows[name]:.3e}", flush=True)
    with open(os.path.join(ROOT, "profiles", "r05_precision_qwen_tower_cpu.json"), "w") as f:
        json.dump({"what": __doc__, "embedding_max_abs": scale, "rel_err_by_rounding_points": rows}, f, indent=1)


if __name__ == "__main__":
    main()
