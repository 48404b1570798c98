cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out && rm -rf gpurun_out/diag_*.jsonl gpurun_out/summary.log gpurun_out/bench_*
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_pipeline.py -m gpu -q --timeout 600 -x -k "large_v3_geometry" > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/summary.log
tail -15 gpurun_out/pytest_gpu.log
timeout 600 python scripts/decode_sweep.py > gpurun_out/decode_sweep.log 2>&1; echo "sweep rc=$?" >> gpurun_out/summary.log
tail -11 gpurun_out/decode_sweep.log; cat gpurun_out/summary.log; grep large_v3_cons gpurun_out/diag_pipeline.jsonl
