# A/B timing of alternative builds of the library (FREESASA_AMD_LIB) on a 300-structure batch,
# with the two LDS counters that matter
export PYTHONUNBUFFERED=1
for lib in "$@"; do
  echo "== lib=$lib"
  FREESASA_AMD_LIB=$lib python bench.py --steps 3 --warmup 1 --structs 300 --no-cpu-baseline 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('kernel_ms %.3f prep_ms %.3f value %.4g' % (d['roofline']['kernel_ms'], d['roofline']['prep_ms'], d['value']))
"
done
