# DEV: A/B of library variants on the S&R workloads: bash tools/gpu_sr_ab.sh lib1.so lib2.so ...   (kernel ms, best of REPS)
export PYTHONUNBUFFERED=1
REPO=$(pwd)
for wl in coil_sr pdb_sr globule_sr; do
 for lib in "$@"; do
  best=999
  for rep in $(seq 1 ${REPS:-2}); do
   km=$(FREESASA_AMD_LIB=$REPO/$lib python bench.py --workload $wl --steps 8 --warmup 3 --no-cpu-baseline --no-end-to-end --no-secondary --no-neighbors --no-live-counters --no-drivers 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%.4f %.4f' % (d['roofline']['kernel_ms'], d['ms_per_step']))")
   echo "$wl $lib rep$rep kernel_ms,ms_per_step $km"
  done
 done
done
