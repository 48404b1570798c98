cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out && rm -rf gpurun_out/diag_*.jsonl gpurun_out/summary.log gpurun_out/bench_*
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_pipeline.py -m gpu -q --timeout 300 -x -k "scene_detection" > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/summary.log
tail -20 gpurun_out/pytest_gpu.log
timeout 600 python bench.py --workload cfg3 --batch 128 --steps 2 --warmup 1 > gpurun_out/bench_cfg3.json 2> gpurun_out/bench_cfg3.err; echo "cfg3 rc=$?" >> gpurun_out/summary.log
cat gpurun_out/bench_cfg3.json; tail -3 gpurun_out/bench_cfg3.err; cat gpurun_out/summary.log
