mkdir -p gpurun_out && rm -f gpurun_out/diag_*.jsonl
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
(rocminfo | grep -E "gfx|Compute Unit" | head -4; python -c "import torch;print(torch.cuda.get_device_name(0), torch.cuda.mem_get_info())") > gpurun_out/env.log 2>&1
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --timeout 300 > gpurun_out/pytest_kernels.log 2>&1; echo "kernels rc=$?" >> gpurun_out/summary.log
timeout 1200 python -m pytest tests/test_gpu_pipeline.py -m gpu -q --timeout 600 > gpurun_out/pytest_pipeline.log 2>&1; echo "pipeline rc=$?" >> gpurun_out/summary.log
timeout 300 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/summary.log
timeout 900 python bench.py --steps 2 --warmup 1 --batch 16 > gpurun_out/bench_b16.json 2> gpurun_out/bench_b16.err; echo "bench rc=$?" >> gpurun_out/summary.log
tail -5 gpurun_out/pytest_kernels.log; tail -30 gpurun_out/pytest_pipeline.log; cat gpurun_out/summary.log; tail -3 gpurun_out/smoke.log; cat gpurun_out/bench_b16.json | head -c 3000; tail -5 gpurun_out/bench_b16.err
