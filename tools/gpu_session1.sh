# DEV: scratch GPU session (edited per call)
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
(timeout 900 python -m pytest tests -m gpu -q -x) > gpurun_out/s18_pytest.log 2>&1; tail -4 gpurun_out/s18_pytest.log
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-end-to-end 2>gpurun_out/s18_bench.err | tail -1 > gpurun_out/s18_bench.json
python - <<'PY'
import json,sys
d=json.load(open('gpurun_out/s18_bench.json'))
print("value %.4g ms_per_step %.3f kernel_ms %.3f prep_ms %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["prep_ms"]), d.get("synchronous_entry"))
for k in ("lr100_config2","globule_lr20","globule_sr100_200k"):
    if k in d: print(k, "%.4g"%d[k]["value"], "kernel_ms %.3f"%d[k]["kernel_ms"])
PY
