cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out && rm -rf gpurun_out/summary.log gpurun_out/bench_*
export TMPDIR=/tmp
timeout 600 python bench.py --workload cfg3 --batch 128 --steps 2 --warmup 1 > gpurun_out/bench_cfg3.json 2> gpurun_out/bench_cfg3.err; echo "cfg3 rc=$?" >> gpurun_out/summary.log
timeout 600 python bench.py --workload cfg3 --batch 128 --steps 2 --warmup 1 --word-timestamps 1 > gpurun_out/bench_cfg3_words.json 2> gpurun_out/bench_cfg3_words.err; echo "cfg3w rc=$?" >> gpurun_out/summary.log
cat gpurun_out/bench_cfg3.json; tail -2 gpurun_out/bench_cfg3.err; cat gpurun_out/bench_cfg3_words.json; tail -2 gpurun_out/bench_cfg3_words.err; cat gpurun_out/summary.log
