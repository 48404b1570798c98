cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out && rm -rf gpurun_out/diag_*.jsonl gpurun_out/summary.log gpurun_out/prof_* gpurun_out/bench_*
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/summary.log
timeout 200 python scripts/gemm_sweep.py > gpurun_out/gemm_sweep.log 2>&1; echo "sweep rc=$?" >> gpurun_out/summary.log
timeout 600 python bench.py --steps 2 --warmup 1 --batch 64 --no-cpu-baseline > gpurun_out/bench_b64.json 2> gpurun_out/bench_b64.err; echo "bench64 rc=$?" >> gpurun_out/summary.log
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench default rc=$?" >> gpurun_out/summary.log
(cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_pmc_fetch128 -o b128 -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --batch 128 --decode-tokens 2 --no-cpu-baseline --no-profile > $GRAFT_REPO_ROOT/gpurun_out/rocprof_pmc_fetch128.log 2>&1); echo "rocprof fetch128 rc=$?" >> gpurun_out/summary.log
tail -5 gpurun_out/pytest_gpu.log; grep glds gpurun_out/gemm_sweep.log; cat gpurun_out/summary.log
python - <<'PY'
import json
for f in ("gpurun_out/bench_b64.json", "gpurun_out/bench_default.json"):
    try:
        d = json.load(open(f)); print(f, d["value"], d["ms_per_step"], d["roofline"], d.get("cpu_baseline")); print({k: (v.get("us_per_launch"), v.get("achieved")) for k, v in d["stages"].items()})
    except Exception as e:
        print(f, "ERR", e)
PY
