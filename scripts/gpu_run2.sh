cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out && rm -f gpurun_out/diag_*.jsonl gpurun_out/summary.log
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 300 > gpurun_out/pytest_kernels.log 2>&1; echo "kernels rc=$?" >> gpurun_out/summary.log
timeout 600 python -m pytest tests/test_gpu_pipeline.py -m gpu -q --timeout 600 -k "not large" > gpurun_out/pytest_pipeline.log 2>&1; echo "pipeline rc=$?" >> gpurun_out/summary.log
timeout 300 python scripts/gemm_sweep.py > gpurun_out/gemm_sweep.log 2>&1; echo "sweep rc=$?" >> gpurun_out/summary.log
timeout 600 python bench.py --steps 2 --warmup 1 --batch 64 --no-cpu-baseline > gpurun_out/bench_b64_reg.json 2> gpurun_out/bench_b64_reg.err; echo "bench reg rc=$?" >> gpurun_out/summary.log
WJ_GEMM_TILE=glds timeout 600 python bench.py --steps 2 --warmup 1 --batch 64 --no-cpu-baseline > gpurun_out/bench_b64_glds.json 2> gpurun_out/bench_b64_glds.err; echo "bench glds rc=$?" >> gpurun_out/summary.log
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r1 -o b16 -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --batch 16 --decode-tokens 64 --no-cpu-baseline --no-profile > $GRAFT_REPO_ROOT/gpurun_out/rocprof_b16.log 2>&1); echo "rocprof rc=$?" >> gpurun_out/summary.log
ls -la gpurun_out/prof_r1 gpurun_out/prof_r1/* 2>/dev/null | head -30
tail -4 gpurun_out/pytest_kernels.log; tail -4 gpurun_out/pytest_pipeline.log; cat gpurun_out/gemm_sweep.log | tail -20; cat gpurun_out/summary.log
python - <<'PY'
import json
for f in ("gpurun_out/bench_b64_reg.json", "gpurun_out/bench_b64_glds.json"):
    try:
        d = json.load(open(f)); print(f, d["value"], d["ms_per_step"], d["roofline"]); print({k: (v.get("us_per_launch"), v.get("achieved")) for k, v in d["stages"].items()})
    except Exception as e:
        print(f, "ERR", e)
PY
