# DEV: A/B of library variants on the L&R workloads (headline coils, globules, PDB entries x 251): kernel ms, best of REPS
#   bash tools/gpu_lr_ab.sh lib1.so lib2.so ...
export PYTHONUNBUFFERED=1
REPO=$(pwd)
for wl in coil_lr pdb_lr; do
 for lib in "$@"; do
  for rep in $(seq 1 ${REPS:-2}); do
   km=$(FREESASA_AMD_LIB=$REPO/$lib python bench.py --workload $wl --steps 10 --warmup 3 --sustain-seconds 0 --no-cpu-baseline --no-end-to-end --no-secondary --no-neighbors --no-live-counters --no-drivers 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%.4f %.4f' % (d['roofline']['kernel_ms'], d['ms_per_step']))")
   echo "$wl $lib rep$rep kernel_ms,ms_per_step $km"
  done
 done
done
