cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out && rm -rf gpurun_out/diag_*.jsonl gpurun_out/summary.log gpurun_out/prof_* gpurun_out/bench_*
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_trace384 -o t384 -- python $R/bench.py --steps 1 --warmup 0 --batch 384 --decode-tokens 32 --no-cpu-baseline --no-profile > $R/gpurun_out/rocprof_trace384.log 2>&1); echo "trace rc=$?" >> gpurun_out/summary.log
(cd /tmp && timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_pmc384 -o p384 -- python $R/bench.py --steps 1 --warmup 0 --batch 384 --decode-tokens 2 --no-cpu-baseline --no-profile > $R/gpurun_out/rocprof_pmc384.log 2>&1); echo "pmc rc=$?" >> gpurun_out/summary.log
ls -la gpurun_out/prof_trace384 gpurun_out/prof_pmc384 | head -30
# keep the merge small: drop everything but stats + kernel trace + counters
find gpurun_out/prof_trace384 gpurun_out/prof_pmc384 -type f -size +40M -delete
tail -3 gpurun_out/rocprof_trace384.log; tail -3 gpurun_out/rocprof_pmc384.log; cat gpurun_out/summary.log
