cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out && rm -rf gpurun_out/summary.log gpurun_out/prof_lds
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $R/gpurun_out/prof_lds -o l64 -- python $R/bench.py --steps 1 --warmup 0 --batch 64 --decode-tokens 2 --no-cpu-baseline --no-profile > $R/gpurun_out/rocprof_lds.log 2>&1); echo "lds rc=$?" >> gpurun_out/summary.log
tail -2 gpurun_out/rocprof_lds.log; cat gpurun_out/summary.log
python - <<'PY'
import csv, collections
rows=list(csv.DictReader(open("gpurun_out/prof_lds/l64_counter_collection.csv")))
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in rows:
    k=r["Kernel_Name"][:64]; agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
for k,v in agg.items():
    if "big_kernel" in k or "attn_enc" in k or "tile_kernel<1" in k:
        print(k); print("   ", {n: f"{x:.3g}" for n,x in v.items()})
        if v.get("SQ_LDS_IDX_ACTIVE"): print("    bank_conflict/idx_active = %.3f" % (v["SQ_LDS_BANK_CONFLICT"]/v["SQ_LDS_IDX_ACTIVE"]))
        if v.get("SQ_WAVE_CYCLES"): print("    wait_any %.2f wait_inst_any %.2f active_inst %.2f lds_inst %.2f of wave cycles" % (v["SQ_WAIT_ANY"]/v["SQ_WAVE_CYCLES"], v["SQ_WAIT_INST_ANY"]/v["SQ_WAVE_CYCLES"], v["SQ_ACTIVE_INST_ANY"]/v["SQ_WAVE_CYCLES"], v["SQ_ACTIVE_INST_LDS"]/v["SQ_WAVE_CYCLES"]))
PY
