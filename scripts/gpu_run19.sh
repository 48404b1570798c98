cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out && rm -rf gpurun_out/diag_*.jsonl gpurun_out/summary.log gpurun_out/bench_*
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_pipeline.py -m gpu -q --timeout 600 -x -k "beam" > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/summary.log
tail -30 gpurun_out/pytest_gpu.log; cat gpurun_out/summary.log


