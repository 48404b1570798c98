cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out && rm -rf gpurun_out/summary.log
export TMPDIR=/tmp
timeout 600 python scripts/decode_sweep.py > gpurun_out/decode_sweep.log 2>&1; echo "sweep rc=$?" >> gpurun_out/summary.log
tail -7 gpurun_out/decode_sweep.log; cat gpurun_out/summary.log
