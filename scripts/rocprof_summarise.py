"""Turn rocprofv3 output directories into the small summaries committed under profiles/.

    python scripts/rocprof_summarise.py stats <dir> <out.csv>        # --kernel-trace --stats run: top kernels by time
    python scripts/rocprof_summarise.py pmc <dir> <out.json> <kernel substring> <windows per launch | auto:bench.json> <bench command>
    python scripts/rocprof_summarise.py mfma <dir> <out.json> <bench command>     # --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE

``pmc``: FETCH_SIZE of a `rocprofv3 --pmc FETCH_SIZE` pass (own pass, no tracing domains), averaged over the
dispatches of the named kernel.  FETCH_SIZE counts kilobytes of 64-byte fabric requests; on gfx950 a wide coalesced
streaming read is tallied at half its bytes (MI355X_MICROARCH.md "HBM"), hence the x2 correction.
"""
import csv
import glob
import json
import os
import sys


def find(d, pattern):
    hits = sorted(glob.glob(os.path.join(d, "**", pattern), recursive=True))
    if not hits:
        raise SystemExit(f"no {pattern} under {d}")
    return hits


def stats(d, out):
    rows = []
    for path in find(d, "*kernel_stats.csv"):
        with open(path) as f:
            rows.extend(csv.DictReader(f))
    rows.sort(key=lambda r: -float(r.get("TotalDurationNs", 0) or 0))
    keep = ["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"]
    with open(out, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=keep, extrasaction="ignore")
        w.writeheader()
        for r in rows[:60]:
            w.writerow(r)
    for r in rows[:8]:
        print(f"{float(r['Percentage']):6.2f}%  {int(r['Calls']):7d} x {float(r['AverageNs']) / 1e3:9.2f} us  {r['Name'][:90]}")


KERNEL_CLASSES = (        # kernel-name substring(s) -> class of the encoder / decode step (first match wins)
    ("attn_enc", "encoder attention"), ("attn_cross_mfma", "decode cross attention"), ("attn_dec", "decode self attention"),
    ("gemm_h_big", "encoder GEMMs (256-tile: qk, v, out, fc1, fc2, cross K/V)"), ("gemm_h_tile", "conv stem + decode tile GEMMs"),
    ("gemm_h_skinny", "decode skinny GEMMs"), ("gemm_h_rows", "decode row GEMMs"), ("layernorm", "LayerNorm"),
    ("beam_topk", "beam top-2K"), ("beam_merge", "beam merge"),
    # Qwen3-ASR (cfg5)
    ("gqa_attn", "Qwen decode attention (GQA, one wave per row and KV head)"), ("prompt_attn", "Qwen prompt attention (MFMA tiles)"),
    ("win_attn", "audio tower windowed attention"), ("swiglu", "SwiGLU (unfused passes)"), ("rmsnorm", "RMSNorm"),
    ("qk_norm_rope", "q/k norm + RoPE + cache append"), ("im2col", "tower im2col"), ("topk_logprob", "top-1 + log-prob"),
    ("gemm_mx8", "MX-fp8 GEMMs"))


def mfma(d, out, command):
    """MfmaUtil per kernel class = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 128): GUI_ACTIVE is summed over the 8 XCDs,
    each with 32 CUs x 4 SIMDs; busy cycles are per SIMD (MI355X_MICROARCH.md, profiling section)."""
    acc = {}
    per_kernel = {}
    # kernel durations of the same run (it carries --kernel-trace): dispatch id -> ns, for the clock the part sustained
    dur = {}
    for path in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                try:
                    dur[r.get("Dispatch_Id")] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
                except (KeyError, ValueError):
                    pass
    seen = set()
    for path in find(d, "*counter_collection.csv"):
        with open(path) as f:
            for r in csv.DictReader(f):
                name, cn = r.get("Kernel_Name", ""), r.get("Counter_Name")
                if cn not in ("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_BUSY_CYCLES"):
                    continue
                label = next((lab for sub, lab in KERNEL_CLASSES if sub in name), None)
                if label is None:
                    continue
                e = acc.setdefault(label, {"SQ_VALU_MFMA_BUSY_CYCLES": 0.0, "GRBM_GUI_ACTIVE": 0.0, "SQ_BUSY_CYCLES": 0.0, "rows": 0, "ns": 0.0})
                e[cn] += float(r["Counter_Value"])
                e["rows"] += cn == "GRBM_GUI_ACTIVE"
                did = r.get("Dispatch_Id")
                if cn == "GRBM_GUI_ACTIVE" and did in dur and (label, did) not in seen:
                    seen.add((label, did))
                    e["ns"] += dur[did]
                short = name.split("(")[0][-120:]
                k = per_kernel.setdefault(short, {"SQ_VALU_MFMA_BUSY_CYCLES": 0.0, "GRBM_GUI_ACTIVE": 0.0, "dispatches": 0})
                if cn in k:
                    k[cn] += float(r["Counter_Value"])
                k["dispatches"] += cn == "GRBM_GUI_ACTIVE"
    if not acc:
        raise SystemExit("no SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE rows found")
    rep = {"source": f"rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -- {command}",
           "formula": "MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 128)", "classes": {}, "kernels": {}}
    enc_busy = enc_gui = 0.0
    for label, e in sorted(acc.items(), key=lambda kv: -kv[1]["GRBM_GUI_ACTIVE"]):
        util = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (e["GRBM_GUI_ACTIVE"] * 128.0) if e["GRBM_GUI_ACTIVE"] else 0.0
        rep["classes"][label] = {"dispatches": e["rows"], "mfma_busy_cycles": e["SQ_VALU_MFMA_BUSY_CYCLES"],
                                 "gui_active_cycles": e["GRBM_GUI_ACTIVE"], "mfma_util": round(util, 4)}
        if e["ns"] > 0:      # GUI_ACTIVE is summed over the 8 XCDs: cycles per XCD / wall time = the clock under this kernel
            ghz = e["GRBM_GUI_ACTIVE"] / 8.0 / e["ns"]
            rep["classes"][label].update({"kernel_time_ms": round(e["ns"] * 1e-6, 3), "effective_clock_ghz": round(ghz, 3),
                                          "mfma_util_x_clock_over_2p4ghz": round(util * ghz / 2.4, 4),
                                          # a derived clock above the part's 2.4 GHz means GRBM_GUI_ACTIVE / kernel time is polluted for
                                          # launches this short (the counter keeps running between back-to-back dispatches): the class's
                                          # mfma_util is then not trustworthy in either direction
                                          "valid": bool(ghz <= 2.4)})
        if label.startswith("encoder"):
            enc_busy += e["SQ_VALU_MFMA_BUSY_CYCLES"]; enc_gui += e["GRBM_GUI_ACTIVE"]
            enc_ns = rep.setdefault("_enc_ns", 0.0) + e["ns"]
            rep["_enc_ns"] = enc_ns
    for name, k in sorted(per_kernel.items(), key=lambda kv: -kv[1]["GRBM_GUI_ACTIVE"])[:24]:
        util = k["SQ_VALU_MFMA_BUSY_CYCLES"] / (k["GRBM_GUI_ACTIVE"] * 128.0) if k["GRBM_GUI_ACTIVE"] else 0.0
        rep["kernels"][name] = {"dispatches": k["dispatches"], "gui_active_cycles": k["GRBM_GUI_ACTIVE"], "mfma_util": round(util, 4)}
    enc_ns = rep.pop("_enc_ns", 0.0)
    if enc_gui:
        rep["encoder_time_weighted_mfma_util"] = round(enc_busy / (enc_gui * 128.0), 4)
        if enc_ns > 0:
            ghz = enc_gui / 8.0 / enc_ns
            rep["encoder_effective_clock_ghz"] = round(ghz, 3)
            rep["encoder_mfma_util_at_nominal_2p4ghz"] = round(enc_busy / (enc_gui * 128.0) * ghz / 2.4, 4)
            rep["note"] = ("mfma_util counts matrix-pipe busy cycles per ACTIVE cycle; FLOP / wall time against the 2.5 PFLOP/s peak (quoted at "
                           "2.4 GHz) is lower by effective_clock / 2.4 GHz -- the clock the part sustains under this load")
    with open(out, "w") as f:
        json.dump(rep, f, indent=1)
    print(json.dumps({"encoder_time_weighted_mfma_util": rep.get("encoder_time_weighted_mfma_util"),
                      **{k: v["mfma_util"] for k, v in rep["classes"].items()}}))


def fetch(d, out, bench_json, command):
    """``rocprofv3 --pmc FETCH_SIZE --kernel-trace`` of a cfg5 step: HBM read bytes per kernel (x2 gfx950 wide-read correction,
    MI355X_MICROARCH.md "HBM") and per greedy DECODE ITERATION -- the dispatches between the first and the last launch of the
    decode-only attention kernel (gqa_attn_kernel; prompts and the aligner's classification pass use prompt_attn_kernel) --
    divided by the iterations the bench line reports."""
    rows = []
    for path in find(d, "*counter_collection.csv"):
        with open(path) as f:
            for r in csv.DictReader(f):
                if r.get("Counter_Name") == "FETCH_SIZE":
                    rows.append((int(r["Dispatch_Id"]), r.get("Kernel_Name", ""), float(r["Counter_Value"])))
    if not rows:
        raise SystemExit("no FETCH_SIZE rows")
    rows.sort()
    line = json.loads(open(bench_json).read().strip().splitlines()[-1])
    cfg = line["config"] if "decode_iterations" in line.get("config", {}) else line.get("cfg5", {}).get("config", {})
    iters = int(cfg["decode_iterations"])
    per = {}
    for _, name, kb in rows:
        short = name.split("(")[0][-110:]
        e = per.setdefault(short, [0, 0.0])
        e[0] += 1
        e[1] += kb * 1024.0 * 2.0
    gqa = [i for i, (_, name, _) in enumerate(rows) if "gqa_attn" in name]
    rep = {"source": f"rocprofv3 --pmc FETCH_SIZE --kernel-trace -- {command}", "gfx950_wide_read_correction": 2.0,
           "hbm_read_bytes_total": sum(v[1] for v in per.values()),
           "kernels": {k: {"dispatches": v[0], "hbm_read_bytes": v[1], "per_dispatch": v[1] / v[0]}
                       for k, v in sorted(per.items(), key=lambda kv: -kv[1][1])[:24]}}
    if gqa:
        # one generation per step: with several steps the ranges of the steps are contiguous blocks separated by prompt passes
        blocks, start, prev = [], gqa[0], gqa[0]
        big_gap = 4000                   # dispatches: a prefill / aligner pass between two generations is far longer than a decode layer
        for i in gqa[1:]:
            if i - prev > big_gap:
                blocks.append((start, prev)); start = i
            prev = i
        blocks.append((start, prev))
        tot = sum(sum(kb for _, _, kb in rows[a: b + 1]) for a, b in blocks) * 1024.0 * 2.0
        rep["decode"] = {"generations": len(blocks), "iterations_per_generation": iters,
                         "hbm_read_bytes_per_iteration": tot / (len(blocks) * iters),
                         "decoder_weight_bytes_per_iteration": cfg.get("decoder_weight_bytes_per_iteration"),
                         "note": "all dispatches from the first to the last gqa_attn_kernel launch of a generation (layers' GEMMs, norms, RoPE, attention, "
                                 "LM head, top-1); the KV cache reads of the live rows come on top of the weight stream"}
    with open(out, "w") as f:
        json.dump(rep, f, indent=1)
    print(json.dumps({k: rep[k] for k in ("hbm_read_bytes_total", "decode") if k in rep}))


def pmc(d, out, kernel, windows, command):
    if isinstance(windows, str) and windows.startswith("auto:"):      # windows of a full-batch launch from the bench line
        line = json.load(open(windows[5:]))
        total = line["workload_facts"]["decode_last_step"]["windows"]
        windows = min(int(line["config"]["windows_per_batch"]), int(total))
    vals = []
    for path in find(d, "*counter_collection.csv"):
        with open(path) as f:
            for r in csv.DictReader(f):
                if kernel in r.get("Kernel_Name", "") and r.get("Counter_Name") == "FETCH_SIZE":
                    vals.append(float(r["Counter_Value"]))
    if not vals:
        raise SystemExit(f"no FETCH_SIZE rows for a kernel containing {kernel!r}")
    # launches of the full batch only (the recording's last batch is smaller): the mode of the distribution
    vals.sort()
    full = [v for v in vals if v > 0.9 * vals[-1]]
    avg_kb = sum(full) / len(full)
    per_launch = avg_kb * 1024.0 * 2.0
    windows = float(windows)
    algo = windows * 2 * 20 * 1500 * 64 * 2.0
    rep = {"source": f"rocprofv3 --pmc FETCH_SIZE -- {command}", "kernel_contains": kernel, "dispatches": len(vals),
           "dispatches_full_batch": len(full), "windows_per_launch_full_batch": windows, "FETCH_SIZE_KB_avg": round(avg_kb, 1),
           "gfx950_wide_read_correction": 2.0, "hbm_read_bytes_per_launch": per_launch,
           "hbm_read_bytes_per_window": per_launch / windows, "algorithmic_bytes_per_launch": algo,
           "traffic_over_algorithmic": round(per_launch / algo, 4),
           "note": "algorithmic = K and V of 1500 keys x 20 heads x 64 x 2 B per window; transposed V rows are padded to 1504 keys"}
    with open(out, "w") as f:
        json.dump(rep, f, indent=1)
    print(json.dumps(rep))


if __name__ == "__main__":
    if len(sys.argv) >= 4 and sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    elif len(sys.argv) >= 5 and sys.argv[1] == "mfma":
        mfma(sys.argv[2], sys.argv[3], " ".join(sys.argv[4:]))
    elif len(sys.argv) >= 6 and sys.argv[1] == "fetch":
        fetch(sys.argv[2], sys.argv[3], sys.argv[4], " ".join(sys.argv[5:]))
    elif len(sys.argv) >= 7 and sys.argv[1] == "pmc":
        pmc(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5], " ".join(sys.argv[6:]))
    else:
        raise SystemExit(__doc__)
