cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out && rm -rf gpurun_out/summary.log gpurun_out/bench_*
export TMPDIR=/tmp
timeout 600 python bench.py --workload cfg3 --batch 128 --steps 2 --warmup 1 > gpurun_out/bench_cfg3.json 2> gpurun_out/bench_cfg3.err; echo "cfg3 rc=$?" >> gpurun_out/summary.log
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 1 --warmup 0 --batch 64 --no-cpu-baseline --no-profile > gpurun_out/bench_torchrun.json 2> gpurun_out/bench_torchrun.err; echo "torchrun rc=$?" >> gpurun_out/summary.log
python - <<'PY'
import json
d=json.load(open("gpurun_out/bench_cfg3.json")); print(d["value"], {k:v for k,v in d["config"].items() if k.startswith("t_") or k in ("scenes","groups")})
for k,v in d["stages"].items(): print("  ",k,v)
PY
cut -c1-300 gpurun_out/bench_torchrun.json; tail -3 gpurun_out/bench_torchrun.err; cat gpurun_out/summary.log
