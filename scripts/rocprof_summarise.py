"""Turn rocprofv3 output directories into the small summaries committed under profiles/.

    python scripts/rocprof_summarise.py stats <dir> <out.csv>        # --kernel-trace --stats run: top kernels by time
    python scripts/rocprof_summarise.py pmc <dir> <out.json> <kernel substring> <windows per launch> <bench command>

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


def pmc(d, out, kernel, windows, command):
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
    elif len(sys.argv) >= 7 and sys.argv[1] == "pmc":
        pmc(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5], " ".join(sys.argv[6:]))
    else:
        raise SystemExit(__doc__)
