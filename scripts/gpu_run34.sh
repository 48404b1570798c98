cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out && rm -rf gpurun_out/summary.log
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_pipeline.py -m gpu -q --timeout 400 -x -k "shim or (device_beam and float32 and 5-1.2) or adapter or fidelity" > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/summary.log
tail -25 gpurun_out/pytest_gpu.log; cat gpurun_out/summary.log
