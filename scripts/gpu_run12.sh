cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out && rm -rf gpurun_out/diag_*.jsonl gpurun_out/summary.log gpurun_out/prof_* gpurun_out/bench_*
export TMPDIR=/tmp
for B in 256 384 512; do
timeout 900 python bench.py --no-cpu-baseline --batch $B --steps 2 --warmup 1 > gpurun_out/bench_b$B.json 2> gpurun_out/bench_b$B.err; echo "bench$B rc=$?" >> gpurun_out/summary.log
cut -c1-900 gpurun_out/bench_b$B.json; tail -2 gpurun_out/bench_b$B.err
done
cat gpurun_out/summary.log
