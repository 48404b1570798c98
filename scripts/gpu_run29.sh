cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out && rm -rf gpurun_out/summary.log gpurun_out/prof_mfma
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
(cd /tmp && timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/prof_mfma -o m128 -- python $R/bench.py --steps 1 --warmup 0 --batch 128 --decode-tokens 2 --no-cpu-baseline --no-profile > $R/gpurun_out/rocprof_mfma.log 2>&1); echo "mfma rc=$?" >> gpurun_out/summary.log
ls -la gpurun_out/prof_mfma | head; tail -3 gpurun_out/rocprof_mfma.log; cat gpurun_out/summary.log
python - <<'PY'
import csv, collections
rows=list(csv.DictReader(open("gpurun_out/prof_mfma/m128_counter_collection.csv")))
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items():
    if "gemm_bf16_big" in k or "attn_enc" in k or "cross_mfma" in k:
        m=sum(v["SQ_VALU_MFMA_BUSY_CYCLES"])/len(v["SQ_VALU_MFMA_BUSY_CYCLES"]); g=sum(v["GRBM_GUI_ACTIVE"])/len(v["GRBM_GUI_ACTIVE"]); b=sum(v.get("SQ_BUSY_CYCLES",[0]))/max(1,len(v.get("SQ_BUSY_CYCLES",[1])))
        print(k, "n=",len(v["GRBM_GUI_ACTIVE"]), "mfma_busy",m, "gui_active",g, "sq_busy", b, "ratio m/g", m/g if g else None)
PY
