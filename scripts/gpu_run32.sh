cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out && rm -rf gpurun_out/diag_*.jsonl gpurun_out/summary.log gpurun_out/bench_*
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/summary.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/summary.log
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc=$?" >> gpurun_out/summary.log
tail -5 gpurun_out/pytest_gpu.log; tail -2 gpurun_out/smoke.log; cut -c1-400 gpurun_out/bench_default.json; echo; cat gpurun_out/summary.log
