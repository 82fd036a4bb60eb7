# sweep launch configurations of the L&R tile kernel on the bench batch (200 structs for speed)
set -x
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
for cfg in "$@"; do
  echo "== CFG $cfg" >> gpurun_out/sweep.log
  FREESASA_AMD_CFG=$cfg timeout 300 python bench.py --steps 3 --warmup 1 --structs 300 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
l=sys.stdin.readline()
try:
    d=json.loads(l); print('value %.4g kernel_ms %.3f prep_ms %.3f fallback %d lds %d B %d TA %d' % (d['value'], d['roofline']['kernel_ms'], d['roofline']['prep_ms'], d['config']['fallback_tiles'], d['config']['lds_bytes_per_block'], d['config']['block_threads'], d['config']['tile_atoms']))
except Exception as e: print('ERR', l[:300])
" >> gpurun_out/sweep.log
done
cat gpurun_out/sweep.log
