cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out && rm -rf gpurun_out/diag_*.jsonl gpurun_out/summary.log gpurun_out/bench_*
export TMPDIR=/tmp
timeout 300 python scripts/attn_sweep.py > gpurun_out/attn_sweep.log 2>&1; echo "sweep rc=$?" >> gpurun_out/summary.log
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 300 -x -k "attention" > gpurun_out/pytest_k.log 2>&1; echo "pytest_k rc=$?" >> gpurun_out/summary.log
timeout 900 python -m pytest tests/test_gpu_pipeline.py -m gpu -q --timeout 600 -x -k "encoder or golden or greedy_decode or tiny" > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/summary.log
tail -10 gpurun_out/attn_sweep.log; tail -4 gpurun_out/pytest_k.log; tail -6 gpurun_out/pytest_gpu.log; cat gpurun_out/summary.log
