# DEV: derived hardware metrics of the L&R kernel (300-structure batch)
export PYTHONUNBUFFERED=1
REPO=$(pwd)
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
for set in "VALUBusy VALUUtilization SALUBusy" "LDSBankConflict SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "MemUnitStalled MemUnitBusy LdsUtil" "SQ_INSTS_VALU_TRANS SQ_INSTS_VALU SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $REPO/gpurun_out/der_$tag -o d -- python $REPO/tools/gpu_shapes.py ${STRUCTS:-300} "$1" > $REPO/gpurun_out/der_$tag.log 2>&1
  python - <<PY
import csv, collections, glob
for f in glob.glob("$REPO/gpurun_out/der_$tag/*counter_collection.csv"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if "lr2_tile<" in r["Kernel_Name"] or "k_lr_tile<64, false, 0" in r["Kernel_Name"]:
            agg[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print(k, " ".join(f"{c}={sum(x)/len(x):.4g}" for c, x in sorted(v.items())))
PY
  tail -2 $REPO/gpurun_out/der_$tag.log | grep -i "error\|invalid" | head -2
done
