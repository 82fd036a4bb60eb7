export PYTHONUNBUFFERED=1
python tools/dev/sr200k_timeline.py
export TMPDIR=/tmp; R=$PWD; cd /tmp
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl -o tl -- python $R/tools/dev/sr200k_timeline.py > /dev/null 2>&1
cd $R
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/tl/**/tl_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# the last complete step: from the last k_prep_general / k_bounds backwards
idx = [i for i, r in enumerate(rows) if "k_prep_general" in r["Kernel_Name"] or "k_bounds" in r["Kernel_Name"]]
a, b = idx[-3], idx[-2]
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:b + 1]:
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} us  +{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.1f} us  {r['Kernel_Name'][:60]}")
PY
