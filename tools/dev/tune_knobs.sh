export PYTHONUNBUFFERED=1
REPO=$(pwd)
REPS=2 bash tools/gpu_sr_ab.sh freesasa_amd/lib/libfreesasa_amd.so freesasa_amd/lib/libvar_srf6.so freesasa_amd/lib/libvar_srf10.so 2>&1 | grep -v globule
for lib in libfreesasa_amd libvar_cw10 libvar_cw14 libvar_cm16 libvar_cm24; do
  for rep in 1 2; do
    km=$(FREESASA_AMD_LIB=$REPO/freesasa_amd/lib/$lib.so python bench.py --workload pdb_lr --steps 10 --warmup 3 --sustain-seconds 0 --no-cpu-baseline --no-end-to-end --no-secondary --no-neighbors --no-live-counters --no-drivers 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%.4f' % d['roofline']['kernel_ms'])")
    gm=$(FREESASA_AMD_LIB=$REPO/freesasa_amd/lib/$lib.so python tools/gpu_shapes.py g100 "0,0,-1,0" 2>&1 | grep kernel_ms | tail -1)
    echo "$lib rep$rep pdb_lr kernel_ms $km | glob $gm"
  done
done
