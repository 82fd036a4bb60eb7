# DEV: scratch GPU session (edited per call; not part of the product)
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
(timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -3)
for rep in 1 2 3; do python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-end-to-end --no-secondary --no-neighbors --no-live-counters 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('value %.4g ms_per_step %.3f kernel_ms %.3f prep_ms %.3f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['prep_ms']))"; done
