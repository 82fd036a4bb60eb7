export PYTHONUNBUFFERED=1
for cfg in "$@"; do
  echo "== CFG $cfg"
  FREESASA_AMD_CFG=$cfg python bench.py --slices 100 --steps 3 --warmup 1 --structs 100 --no-cpu-baseline 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('value %.4g kernel_ms %.3f prep_ms %.3f fallback %d lds %d B %d TA %d' % (d['value'], d['roofline']['kernel_ms'], d['roofline']['prep_ms'], d['config']['fallback_tiles'], d['config']['lds_bytes_per_block'], d['config']['block_threads'], d['config']['tile_atoms']))
"
done
