# DEV: instruction counters of the S&R tile kernel for several library variants on one workload:  bash tools/dev/sr_pmc_ab.sh coil_sr lib1.so lib2.so ...
export PYTHONUNBUFFERED=1
REPO=$(pwd)
WL=$1; shift
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
i=0
for lib in "$@"; do
  i=$((i+1))
  rm -rf $REPO/gpurun_out/srpmc_$i
  for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU SQ_ACTIVE_INST_SCA"; do
    FREESASA_AMD_LIB=$REPO/$lib timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $REPO/gpurun_out/srpmc_$i -o pmc$RANDOM -- python $REPO/bench.py --workload $WL --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end --no-secondary --no-neighbors --no-live-counters --no-drivers > $REPO/gpurun_out/srpmc_$i.log 2>&1
  done
  echo "== $lib"
  python - $REPO/gpurun_out/srpmc_$i <<'PY'
import csv, collections, glob, sys
agg = collections.defaultdict(float); n = collections.defaultdict(set)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "k_sr_tile<128, false, 0" in k or "k_sr_tile<256, false, 0" in k:
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]].add(r["Dispatch_Id"])
print("  ".join(f"{c}={agg[c] / max(1, len(n[c])):.4g}" for c in sorted(agg)))
PY
done
