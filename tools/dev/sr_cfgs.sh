export PYTHONUNBUFFERED=1
run() { # lib cfg
  for wl in coil_sr pdb_sr; do
   km=$(FREESASA_AMD_CFG="$2" FREESASA_AMD_LIB=$PWD/$1 python bench.py --workload $wl --steps 8 --warmup 3 --no-cpu-baseline --no-end-to-end --no-secondary --no-neighbors --no-live-counters --no-drivers 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%.4f %.4f ft=%d' % (d['roofline']['kernel_ms'], d['ms_per_step'], d['config']['fallback_tiles']))")
   echo "$wl $1 cfg=$2 kernel_ms,ms_per_step $km"
  done
}
run freesasa_amd/lib/libvar_srwpe7.so ""
run freesasa_amd/lib/libvar_srwpe8.so ""
run freesasa_amd/lib/libfreesasa_amd.so "256,8,48,0"
run freesasa_amd/lib/libfreesasa_amd.so "256,8,56,0"
run freesasa_amd/lib/libfreesasa_amd.so "256,8,64,0"
run freesasa_amd/lib/libfreesasa_amd.so "256,8,80,0"
run freesasa_amd/lib/libfreesasa_amd.so "128,4,64,0"
run freesasa_amd/lib/libfreesasa_amd.so "128,8,64,0"
run freesasa_amd/lib/libfreesasa_amd.so "256,16,64,0"
run freesasa_amd/lib/libfreesasa_amd.so "64,2,64,0"
run freesasa_amd/lib/libvar_srwpe7.so "256,8,56,0"
