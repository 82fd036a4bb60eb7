# DEV: the S&R arrangements side by side (FREESASA_AMD_SR_CAPS=0: the second; "N,L": the third with that table), kernel ms
export PYTHONUNBUFFERED=1
REPO=$(pwd)
run() { # lib caps workload
  FREESASA_AMD_SR_CAPS=$2 FREESASA_AMD_LIB=$REPO/$1 python bench.py --workload $3 --steps 8 --warmup 3 --no-cpu-baseline --no-end-to-end --no-secondary --no-neighbors --no-live-counters --no-drivers 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%.4f %.4f' % (d['roofline']['kernel_ms'], d['ms_per_step']), d.get('atoms_with_a_different_count', d['config'].get('atoms_with_a_different_count')))"
}
for wl in pdb_sr coil_sr globule_sr; do
  for spec in "$@"; do
    lib=${spec%%:*}; caps=${spec##*:}
    echo "$wl $lib caps=$caps kernel_ms,ms_per_step,diff: $(run $lib $caps $wl) | $(run $lib $caps $wl)"
  done
done
