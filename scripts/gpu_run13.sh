cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out && rm -rf gpurun_out/diag_*.jsonl gpurun_out/summary.log gpurun_out/prof_* gpurun_out/bench_*
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 300 -x -k "gemm" > gpurun_out/pytest_k.log 2>&1; echo "pytest_k rc=$?" >> gpurun_out/summary.log
timeout 300 python scripts/gemm_sweep.py > gpurun_out/gemm_sweep.log 2>&1; echo "sweep rc=$?" >> gpurun_out/summary.log
timeout 900 python -m pytest tests/test_gpu_pipeline.py -m gpu -q --timeout 600 -x -k "encoder or golden or greedy_decode" > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/summary.log
timeout 600 python bench.py --no-cpu-baseline --batch 128 --steps 2 > gpurun_out/bench_b128.json 2> gpurun_out/bench_b128.err; echo "bench rc=$?" >> gpurun_out/summary.log
tail -5 gpurun_out/pytest_k.log; tail -14 gpurun_out/gemm_sweep.log; tail -8 gpurun_out/pytest_gpu.log; cut -c1-200 gpurun_out/bench_b128.json; tail -3 gpurun_out/bench_b128.err; cat gpurun_out/summary.log
python - <<'PY'
import json
d=json.load(open("gpurun_out/bench_b128.json")); st=d["stages"]
for k in ("enc_fc1_gemm","enc_fc2_gemm","enc_qk_gemm","enc_v_gemm","enc_out_gemm","cross_kv_gemm","enc_attention","_encoder_mfma_aggregate"):
    print(k, st[k].get("us_per_launch"), st[k].get("achieved"))
PY
