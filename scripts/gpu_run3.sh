cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out && rm -rf gpurun_out/diag_*.jsonl gpurun_out/summary.log gpurun_out/prof_*
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/summary.log
timeout 300 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/summary.log
timeout 200 python scripts/gemm_sweep.py > gpurun_out/gemm_sweep.log 2>&1; echo "sweep rc=$?" >> gpurun_out/summary.log
timeout 600 python bench.py --steps 2 --warmup 1 --batch 64 --no-cpu-baseline > gpurun_out/bench_b64.json 2> gpurun_out/bench_b64.err; echo "bench64 rc=$?" >> gpurun_out/summary.log
timeout 600 python bench.py --steps 2 --warmup 1 --batch 128 --no-cpu-baseline > gpurun_out/bench_b128.json 2> gpurun_out/bench_b128.err; echo "bench128 rc=$?" >> gpurun_out/summary.log
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_stats -o b64 -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --batch 64 --decode-tokens 32 --no-cpu-baseline --no-profile > $GRAFT_REPO_ROOT/gpurun_out/rocprof_stats.log 2>&1); echo "rocprof stats rc=$?" >> gpurun_out/summary.log
(cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_pmc_fetch -o b16 -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --batch 16 --decode-tokens 4 --no-cpu-baseline --no-profile > $GRAFT_REPO_ROOT/gpurun_out/rocprof_pmc_fetch.log 2>&1); echo "rocprof fetch rc=$?" >> gpurun_out/summary.log
(cd /tmp && timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_pmc_write -o b16 -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --batch 16 --decode-tokens 4 --no-cpu-baseline --no-profile > $GRAFT_REPO_ROOT/gpurun_out/rocprof_pmc_write.log 2>&1); echo "rocprof write rc=$?" >> gpurun_out/summary.log
find gpurun_out/prof_* -type f | head -20; du -sh gpurun_out
tail -6 gpurun_out/pytest_gpu.log; tail -12 gpurun_out/gemm_sweep.log; cat gpurun_out/summary.log
python - <<'PY'
import json
for f in ("gpurun_out/bench_b64.json", "gpurun_out/bench_b128.json"):
    try:
        d = json.load(open(f)); print(f, d["value"], d["ms_per_step"], d["roofline"]); print({k: (v.get("us_per_launch"), v.get("achieved")) for k, v in d["stages"].items()})
    except Exception as e:
        print(f, "ERR", e)
PY
