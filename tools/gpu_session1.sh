# DEV: scratch GPU session (edited per call)
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
(timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -3)
run() { # envs
  env $1 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-end-to-end --no-secondary --no-neighbors --no-live-counters 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', 'value %.4g ms_per_step %.3f kernel_ms %.3f prep_ms %.3f fallback %d' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['prep_ms'], d['config']['fallback_tiles']))"
}
for rep in 1 2 3; do run "FREESASA_AMD_DENSE_CELLS=1"; run "X=1"; done 2>&1 | tee gpurun_out/s26.log
