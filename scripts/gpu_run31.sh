cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out && rm -rf gpurun_out/summary.log
export TMPDIR=/tmp
timeout 300 python scripts/gemm_sweep.py > gpurun_out/gemm_sweep.log 2>&1; echo "sweep rc=$?" >> gpurun_out/summary.log
tail -13 gpurun_out/gemm_sweep.log; cat gpurun_out/summary.log
