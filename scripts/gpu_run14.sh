cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out && rm -rf gpurun_out/diag_*.jsonl gpurun_out/summary.log gpurun_out/prof_* gpurun_out/bench_*
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 300 -x -k "gemm or attention" > gpurun_out/pytest_k.log 2>&1; echo "pytest_k rc=$?" >> gpurun_out/summary.log
timeout 900 python scripts/decode_sweep.py > gpurun_out/decode_sweep.log 2>&1; echo "sweep rc=$?" >> gpurun_out/summary.log
timeout 900 python -m pytest tests/test_gpu_pipeline.py -m gpu -q --timeout 600 -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/summary.log
timeout 600 python bench.py --no-cpu-baseline --steps 2 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc=$?" >> gpurun_out/summary.log
tail -5 gpurun_out/pytest_k.log; tail -14 gpurun_out/decode_sweep.log; tail -8 gpurun_out/pytest_gpu.log; cut -c1-200 gpurun_out/bench_default.json; tail -3 gpurun_out/bench_default.err; cat gpurun_out/summary.log
python - <<'PY'
import json
d=json.load(open("gpurun_out/bench_default.json")); st=d["stages"]
for k,v in sorted(st.items(), key=lambda kv:-kv[1].get('ms_total',0)):
    print(k, v.get("us_per_launch"), v.get("achieved"), v.get("share"))
PY
