# DEV: atoms per tile at 100 slices (BASELINE configs[2] as written): FREESASA_AMD_LR2="TA,pool,ds,refill"
export PYTHONUNBUFFERED=1
for ta in 0 2 3 4 5; do
  export FREESASA_AMD_LR2="$ta,0,-1,0"
  python bench.py --workload coil_lr --slices 100 --structs 300 --steps 4 --warmup 2 --sustain-seconds 0 --no-cpu-baseline --no-end-to-end --no-secondary --no-neighbors --no-live-counters --no-drivers 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('lr100 TA=$ta kernel_ms %.4f step %.4f value %.4e' % (d['roofline']['kernel_ms'], d['ms_per_step'], d['value']))"
done
