cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out && rm -rf gpurun_out/summary.log gpurun_out/prof_lds2
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 200 -x -k "attention_enc" > gpurun_out/pytest_k.log 2>&1; echo "pytest_k rc=$?" >> gpurun_out/summary.log
timeout 200 python scripts/attn_sweep.py > gpurun_out/attn_sweep.log 2>&1; echo "sweep rc=$?" >> gpurun_out/summary.log
tail -3 gpurun_out/pytest_k.log; tail -9 gpurun_out/attn_sweep.log; cat gpurun_out/summary.log
