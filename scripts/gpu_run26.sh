cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out && rm -rf gpurun_out/diag_*.jsonl gpurun_out/summary.log gpurun_out/bench_* gpurun_out/prof_*
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/summary.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/summary.log
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc=$?" >> gpurun_out/summary.log
timeout 600 python bench.py --workload cfg3 --steps 2 --warmup 1 > gpurun_out/bench_cfg3.json 2> gpurun_out/bench_cfg3.err; echo "cfg3 rc=$?" >> gpurun_out/summary.log
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_trace384 -o t384 -- python $R/bench.py --steps 1 --warmup 0 --batch 384 --decode-tokens 32 --no-cpu-baseline --no-profile > $R/gpurun_out/rocprof_trace384.log 2>&1); echo "trace rc=$?" >> gpurun_out/summary.log
(cd /tmp && timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_pmc384 -o p384 -- python $R/bench.py --steps 1 --warmup 0 --batch 384 --decode-tokens 2 --no-cpu-baseline --no-profile > $R/gpurun_out/rocprof_pmc384.log 2>&1); echo "pmc rc=$?" >> gpurun_out/summary.log
find gpurun_out/prof_trace384 gpurun_out/prof_pmc384 -type f -size +40M -delete
rm -f gpurun_out/prof_trace384/t384_kernel_trace.csv gpurun_out/prof_pmc384/p384_kernel_trace.csv
tail -6 gpurun_out/pytest_gpu.log; tail -2 gpurun_out/smoke.log; cut -c1-1200 gpurun_out/bench_default.json; echo; cut -c1-900 gpurun_out/bench_cfg3.json; echo; cat gpurun_out/summary.log
