cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out && rm -rf gpurun_out/summary.log
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_gpu_pipeline.py -m gpu -q --timeout 300 -x -k "sharded_transcribe" > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/summary.log
tail -25 gpurun_out/pytest_gpu.log; cat gpurun_out/summary.log
