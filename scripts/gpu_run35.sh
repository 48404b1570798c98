cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out && rm -rf gpurun_out/diag_*.jsonl gpurun_out/summary.log
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q --timeout 900 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/summary.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/summary.log
tail -5 gpurun_out/pytest_gpu.log; tail -2 gpurun_out/smoke.log; cat gpurun_out/summary.log
