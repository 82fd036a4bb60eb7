export PYTHONUNBUFFERED=1 TMPDIR=/tmp
REPO=$(pwd); cd /tmp
PM="python $REPO/bench.py --steps 1 --warmup 1 --structs 300 --no-cpu-baseline"
rocprofv3 --kernel-trace --pmc VALUBusy SALUBusy VALUUtilization --output-format csv -d $REPO/gpurun_out/prof_d -o d1 -- $PM > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc LDSBankConflict MemUnitBusy MemUnitStalled --output-format csv -d $REPO/gpurun_out/prof_d -o d2 -- $PM > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc OccupancyPercent MeanOccupancyPerCU GRBM_GUI_ACTIVE --output-format csv -d $REPO/gpurun_out/prof_d -o d3 -- $PM > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_ACCUM_PREV_HIRES SQ_BUSY_CU_CYCLES SQ_CYCLES --output-format csv -d $REPO/gpurun_out/prof_d -o d4 -- $PM > /dev/null 2>&1
cd $REPO
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/prof_d/*_counter_collection.csv')):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name'][:40]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items():
        if 'k_lr_tile<128, false' in k:
            print(f.split('/')[-1][:2], {c: round(sum(x)/len(x),3) for c,x in v.items()})
PY
