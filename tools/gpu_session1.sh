# DEV: scratch GPU session (edited per call)
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
(timeout 900 python -m pytest tests -m gpu -q -x -s 2>&1 | grep -v "^$") > gpurun_out/s23_pytest.log 2>&1; grep "adversarial\|passed\|failed\|Error" gpurun_out/s23_pytest.log | tail -12
(timeout 900 python bench.py) > gpurun_out/s23_bench.out 2> gpurun_out/s23_bench.err; echo "bench rc=$?"; tail -3 gpurun_out/s23_bench.err
grep '^{' gpurun_out/s23_bench.out | tail -1 > gpurun_out/s23_bench.json
python - <<'PY'
import json,sys
d=json.load(open('gpurun_out/s23_bench.json'))
print("value %.4g ms_per_step %.3f kernel_ms %.3f prep_ms %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["prep_ms"]), d.get("synchronous_entry"))
print("roofline", {k:d["roofline"][k] for k in ("achieved","frac","traffic","traffic_source")}, d["roofline"]["valu_issue"])
for k in ("lr100_config2","globule_lr20","real_pdb_lr20","globule_sr100_200k"):
    if k in d: print(k, {kk:vv for kk,vv in d[k].items() if kk!="workload"})
print("cpu", d.get("cpu_baseline",{}).get("value"), d.get("max_abs_dsasa_vs_cpu"), "e2e", d.get("end_to_end",{}).get("value"))
PY
