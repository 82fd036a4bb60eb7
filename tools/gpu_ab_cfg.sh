# A/B: library variants x launch configs.  usage: gpu_ab_cfg.sh "lib1 lib2" "cfg1 cfg2 ..."
export PYTHONUNBUFFERED=1
for lib in $1; do for cfg in $2; do
  echo -n "lib=$(basename $lib) cfg=$cfg : "
  FREESASA_AMD_LIB=$lib FREESASA_AMD_CFG=$cfg python bench.py --steps 3 --warmup 1 --structs 300 --no-cpu-baseline 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('value %.4g kernel_ms %.3f fallback %d lds %d' % (d['value'], d['roofline']['kernel_ms'], d['config']['fallback_tiles'], d['config']['lds_bytes_per_block']))
"
done; done
