# DEV: time and wave-level VALU instructions of k_sr_tile, cumulative by phase (variants built with -DSR_STOP_AFTER=k)
#   for k in 0 1 12 2 3; do bash tools/build_variant.sh srstop$k -DSR_STOP_AFTER=$k; done; bash tools/gpu_sr_ablate.sh
# third arrangement (sr_caps.h): after 0 load, 1 neighbors, 12 lists + lookup, 2 to-do, 3 exact; then the whole kernel
export PYTHONUNBUFFERED=1
REPO=$(pwd)
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
for wl in coil_sr pdb_sr; do
 echo "--- $wl"
 for lib in libvar_srstop0 libvar_srstop1 libvar_srstop12 libvar_srstop2 libvar_srstop3 libfreesasa_amd; do
  [ -f $REPO/freesasa_amd/lib/$lib.so ] || continue
  P="python $REPO/bench.py --workload $wl --steps 3 --warmup 2 --no-cpu-baseline --no-end-to-end --no-secondary --no-neighbors --no-live-counters --no-drivers"
  rm -rf $REPO/gpurun_out/srabl_$lib
  FREESASA_AMD_LIB=$REPO/freesasa_amd/lib/$lib.so timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $REPO/gpurun_out/srabl_$lib -o p -- $P > $REPO/gpurun_out/srabl_$lib.log 2>&1
  python - $REPO/gpurun_out/srabl_$lib $lib <<'PY'
import csv, collections, glob, sys, json
agg = collections.defaultdict(float); n = collections.defaultdict(set)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_sr_tile<128, false, 0, true>" in r["Kernel_Name"] or "k_sr_tile<256, false, 0, false>" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]].add(r["Dispatch_Id"])
line = open(sys.argv[1] + ".log").read()
js = [l for l in line.splitlines() if l.startswith("{")]
km = json.loads(js[-1])["roofline"]["kernel_ms"] if js else float("nan")
print(f"{sys.argv[2]:20s} kernel_ms {km:8.3f}  " + "  ".join(f"{c} {agg[c]/max(1,len(n[c])):.4g}" for c in sorted(agg)))
PY
 done
done
