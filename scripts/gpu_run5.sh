cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out && rm -rf gpurun_out/diag_*.jsonl gpurun_out/summary.log gpurun_out/prof_* gpurun_out/bench_*
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_pipeline.py -m gpu -q --timeout 600 -k "gemm or chains or greedy or golden_small or bisection" > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/summary.log
timeout 200 python scripts/gemm_sweep.py > gpurun_out/gemm_sweep.log 2>&1; echo "sweep rc=$?" >> gpurun_out/summary.log
for c in 1 2 4; do
WJ_DECODE_CHAINS=$c timeout 400 python bench.py --steps 1 --warmup 1 --batch 64 --no-cpu-baseline --no-profile > gpurun_out/bench_b64_c$c.json 2> gpurun_out/bench_b64_c$c.err; echo "bench64 c$c rc=$?" >> gpurun_out/summary.log
done
for c in 2 4; do
WJ_DECODE_CHAINS=$c timeout 400 python bench.py --steps 1 --warmup 1 --batch 128 --no-cpu-baseline --no-profile > gpurun_out/bench_b128_c$c.json 2> gpurun_out/bench_b128_c$c.err; echo "bench128 c$c rc=$?" >> gpurun_out/summary.log
done
timeout 600 python bench.py --workload cfg3 --steps 1 --warmup 1 --batch 32 > gpurun_out/bench_cfg3.json 2> gpurun_out/bench_cfg3.err; echo "cfg3 rc=$?" >> gpurun_out/summary.log
tail -5 gpurun_out/pytest_gpu.log; grep glds gpurun_out/gemm_sweep.log; cat gpurun_out/summary.log; tail -3 gpurun_out/bench_cfg3.err; cat gpurun_out/bench_cfg3.json
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/bench_b*_c*.json")):
    try:
        d = json.load(open(f)); print(f, d["value"], d["ms_per_step"])
    except Exception as e:
        print(f, "ERR", e)
PY
