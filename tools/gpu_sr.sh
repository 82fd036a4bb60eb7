# Shrake-Rupley profile session: kernel trace + counters of k_sr_tile on the three S&R workloads of the bench
# (coil_sr: 1000 x 10 000 coils; pdb_sr: the reference's PDB entries x 251; globule_sr: one 200 000-atom globule).
#   bash tools/gpu_sr.sh r05 [lib.so]
TAG=${1:-r05}
export PYTHONUNBUFFERED=1
REPO=$(pwd)
O=$REPO/gpurun_out
mkdir -p $O
[ -n "$2" ] && export FREESASA_AMD_LIB=$REPO/$2
export TMPDIR=/tmp
cd /tmp
for wl in coil_sr pdb_sr globule_sr; do
  B="python $REPO/bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline --no-end-to-end --no-secondary --no-neighbors --no-live-counters --no-drivers"
  (timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/sr_${TAG}_$wl -o trace -- $B) > $O/sr_${TAG}_${wl}_trace.log 2>&1
  cp $O/sr_${TAG}_$wl/trace_kernel_stats.csv $O/${TAG}_sr_${wl}_kernel_stats.csv 2>/dev/null
  grep '^{' $O/sr_${TAG}_${wl}_trace.log | tail -1 | cut -c1-300
  P="python $REPO/bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end --no-secondary --no-neighbors --no-live-counters --no-drivers"
  i=0
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VALU" "VALUBusy VALUUtilization SALUBusy" "LDSBankConflict LdsUtil MemUnitStalled"; do
    i=$((i+1))
    (timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/sr_${TAG}_$wl -o pmc$i -- $P) > $O/sr_${TAG}_${wl}_pmc$i.log 2>&1
  done
  python - $O/sr_${TAG}_$wl <<'PY' > $O/${TAG}_sr_$(echo $wl)_pmc_summary.txt
import csv, collections, glob, sys
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(set))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "k_sr_tile" in k or "k_sort" in k or "k_scatter" in k or "k_count" in k:
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]].add(r["Dispatch_Id"])
for k in sorted(agg):
    print(k)
    for c in sorted(agg[k]):
        per = agg[k][c] / max(1, len(n[k][c]))
        print(f"    {c:24s} per launch {per:.6g}   (launches {len(n[k][c])})")
PY
  cat $O/${TAG}_sr_${wl}_pmc_summary.txt | head -40
  head -6 $O/${TAG}_sr_${wl}_kernel_stats.csv | cut -d, -f1-8
done
