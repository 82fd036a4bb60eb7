# DEV: kernel + memory-copy trace of the trajectory-file driver (where do a shard's 4 ms on the device go?)
export PYTHONUNBUFFERED=1
R=$PWD; export TMPDIR=/tmp; cd /tmp
cat > /tmp/tt.py <<PY
import os, sys
sys.path.insert(0, "$R")
import numpy as np, tools, freesasa_amd as fa
scratch = "/tmp/drv_tuning"; os.makedirs(scratch, exist_ok=True)
n_atoms, n_frames = 100_000, 240
base, r = tools.globule(n_atoms, 5)
f64 = os.path.join(scratch, "frames240.f64")
with open(f64, "wb") as a:
    for f in range(n_frames): tools.jitter(base, 100 + f, 0.5).tofile(a)
os.environ["FREESASA_AMD_TRAJ_LANES"] = "3"
fa.trajectory_file(f64, r, os.path.join(scratch, "t.bin"), os.path.join(scratch, "s.bin"), n_frames=72)
fa.trajectory_file(f64, r, os.path.join(scratch, "t.bin"), os.path.join(scratch, "s.bin"))
PY
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/tt -o tt -- python /tmp/tt.py > /dev/null 2>&1
cd $R
python - <<'PY'
import csv, glob
ev = []
for f in glob.glob("gpurun_out/tt/**/tt_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)): ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"][:40], r.get("Queue_Id", "")))
for f in glob.glob("gpurun_out/tt/**/tt_memory_copy_trace.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    print("copy columns:", list(rows[0].keys()) if rows else None)
    for r in rows: ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C %s %s B" % (r.get("Direction", "?"), r.get("Bytes", r.get("Size", "?"))), ""))
ev.sort()
t0 = ev[-1][0] - 15_000_000   # the last 15 ms
for s, e, n, q in ev:
    if s >= t0 and (e - s) > 20_000: print(f"{(s - t0) / 1e6:8.3f} ms  +{(e - s) / 1e6:7.3f} ms  {n} {q}")
PY
