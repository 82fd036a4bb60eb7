export PYTHONUNBUFFERED=1
for g in "$@"; do
  echo -n "grid=$g : "
  FREESASA_AMD_GRID=$g python bench.py --steps 3 --warmup 1 --structs 300 --no-cpu-baseline 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('value %.4g kernel_ms %.3f fallback %d lds %d' % (d['value'], d['roofline']['kernel_ms'], d['config']['fallback_tiles'], d['config']['lds_bytes_per_block']))
"
done
