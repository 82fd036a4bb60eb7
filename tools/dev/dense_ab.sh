# DEV: A/B of library variants on the DENSE Lee-Richards workloads (kernel ms, best of REPS): bash tools/dev/dense_ab.sh lib1.so lib2.so ...
export PYTHONUNBUFFERED=1
REPO=$(pwd)
for rep in $(seq 1 ${REPS:-3}); do
 for lib in "$@"; do
  for wl in pdb_lr; do
   km=$(FREESASA_AMD_LIB=$REPO/$lib python bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline --no-end-to-end --no-secondary --no-neighbors --no-live-counters --no-drivers --sustain-seconds 0 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%.4f' % d['roofline']['kernel_ms'])")
   echo "$wl $lib kernel_ms $km"
  done
  FREESASA_AMD_LIB=$REPO/$lib python tools/gpu_shapes.py g100 "0,0,-1,0" 2>&1 | grep kernel_ms | sed "s#^#glob $lib #"
 done
done 2>&1 | tee gpurun_out/dense_ab.txt | awk '{k=$1" "$2; v=$NF; for(i=1;i<=NF;i++) if($i=="kernel_ms") v=$(i+1); if(!(k in b) || v<b[k]) b[k]=v} END{for(k in b) print k, b[k]}' | sort
