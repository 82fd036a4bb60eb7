# DEV: launch shapes of the L&R kernel at other slice counts: bash tools/gpu_r2_shapes.sh "<ns> <structs> <spec> ..." ...
for job in "$@"; do
  set -- $job; ns=$1; st=$2; shift 2
  for spec in "$@"; do
    [ "$spec" = "-" ] && spec=""
    FREESASA_AMD_LR2="$spec" timeout 200 python bench.py --slices $ns --structs $st --no-cpu-baseline --no-end-to-end 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print('ns=$ns spec=%-14s %.4g atoms/s  ms %.3f kernel %.3f fallback %s lds %s TA %s' % ('$spec', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], c['fallback_tiles'], c['lds_bytes_per_block'], c['tile_atoms']))"
  done
done
