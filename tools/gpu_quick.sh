# quick GPU session: the -m gpu suite (optionally a -k filter) and, with BENCH=1, the default bench line
#   gpurun -- 'bash tools/gpu_quick.sh r06a'      K="expr" limits pytest; BENCH=0 skips the bench
TAG=${1:-r06}
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
O=gpurun_out
if [ -n "$K" ]; then
  (timeout 1200 python -m pytest tests -m gpu -q -x -k "$K" --durations=8) > $O/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/${TAG}_pytest_gpu.log
else
  (timeout 1500 python -m pytest tests -m gpu -q --durations=8) > $O/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/${TAG}_pytest_gpu.log
fi
tail -25 $O/${TAG}_pytest_gpu.log
if [ "${BENCH:-1}" = "1" ]; then
  (timeout 900 python bench.py ${BENCH_ARGS}) > $O/${TAG}_bench.out 2> $O/${TAG}_bench.err; echo "bench rc=$?" >> $O/${TAG}_bench.err
  grep '^{' $O/${TAG}_bench.out | tail -1 > $O/${TAG}_bench.json
  tail -3 $O/${TAG}_bench.err
  python - $O/${TAG}_bench.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
def show(k, v, ind=0):
    if isinstance(v, dict):
        flat = {a: b for a, b in v.items() if not isinstance(b, (dict, list))}
        print(" " * ind + k + ": " + ", ".join(f"{a}={b:.4g}" if isinstance(b, float) else f"{a}={b}" for a, b in flat.items() if len(str(b)) < 60))
        for a, b in v.items():
            if isinstance(b, dict): show(a, b, ind + 2)
    else:
        print(" " * ind + f"{k}: {v}" if len(str(v)) < 100 else " " * ind + f"{k}: ...")
for k, v in d.items(): show(k, v)
PY
fi
