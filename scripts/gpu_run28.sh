cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out && rm -rf gpurun_out/summary.log
export TMPDIR=/tmp
timeout 900 python bench.py --workload cfg3 --minutes 120 --steps 1 --warmup 1 --no-profile > gpurun_out/bench_cfg3_120min.json 2> gpurun_out/bench_cfg3_120min.err; echo "cfg3-120 rc=$?" >> gpurun_out/summary.log
cat gpurun_out/bench_cfg3_120min.json; tail -3 gpurun_out/bench_cfg3_120min.err; cat gpurun_out/summary.log
