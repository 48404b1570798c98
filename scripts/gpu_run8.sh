cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out && rm -rf gpurun_out/diag_*.jsonl gpurun_out/summary.log gpurun_out/prof_* gpurun_out/bench_*
export TMPDIR=/tmp
timeout 300 python scripts/attn_sweep.py > gpurun_out/attn_sweep.log 2>&1; echo "attn rc=$?" >> gpurun_out/summary.log
timeout 600 python scripts/decode_sweep.py > gpurun_out/decode_sweep.log 2>&1; echo "sweep rc=$?" >> gpurun_out/summary.log
timeout 900 python -m pytest tests/test_gpu_pipeline.py -m gpu -q --timeout 600 -k "full_size or minimum or degenerate or deterministic or attention" > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/summary.log
cat gpurun_out/attn_sweep.log | tail -17; tail -9 gpurun_out/decode_sweep.log; tail -5 gpurun_out/pytest_gpu.log; cat gpurun_out/summary.log
