# DEV: kernel + memory-copy trace of the file sweep with the parser on the device (what does the GPU do in a sweep's 70 ms?)
export PYTHONUNBUFFERED=1
R=$PWD; export TMPDIR=/tmp; cd /tmp
cat > /tmp/st.py <<PY
import os, sys, shutil
sys.path.insert(0, "$R")
import numpy as np, freesasa_amd as fa, bench
from freesasa_amd import ingest
d = "/tmp/fsbench/sweep_files"
if not os.path.isdir(d):
    os.makedirs(d, exist_ok=True)
    pdb_dir, cif_dir = os.path.join(bench.ROOT, "tests", "golden", "pdb"), os.path.join(bench.ROOT, "tests", "golden", "cif")
    srcs = [os.path.join(pdb_dir, nm + ".pdb") for nm in bench.PDB_NAMES] + sorted(os.path.join(cif_dir, f) for f in os.listdir(cif_dir) if f.endswith(".cif"))[:4]
    for k in range(163):
        for sp in srcs: shutil.copyfile(sp, os.path.join(d, f"{k:04d}_{os.path.basename(sp)}"))
paths = sorted(os.path.join(d, f) for f in os.listdir(d)) * 4
fa.sweep_files(paths[:400], device=0, ingest_options=ingest.PARSE_ON_DEVICE)
fa.sweep_files(paths, device=0, ingest_options=ingest.PARSE_ON_DEVICE)
fa.sweep_files(paths, device=0, ingest_options=ingest.PARSE_ON_DEVICE)
PY
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/st -o st -- python /tmp/st.py > /dev/null 2>&1
cd $R
python - <<'PY'
import csv, glob
ev = []
for f in glob.glob("gpurun_out/st/**/st_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)): ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K", r["Kernel_Name"][:34], r.get("Queue_Id", "")))
for f in glob.glob("gpurun_out/st/**/st_memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)): ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C", r.get("Direction", "?")[12:], ""))
ev.sort()
end = ev[-1][1]
t0 = end - 70_000_000   # the last sweep
sel = [e for e in ev if e[0] >= t0]
# busy time of kernels (union), of copies (union), by kernel name
def union(iv):
    iv = sorted(iv); tot = 0; cs, ce = None, None
    for s, e in iv:
        if cs is None: cs, ce = s, e
        elif s <= ce: ce = max(ce, e)
        else: tot += ce - cs; cs, ce = s, e
    if cs is not None: tot += ce - cs
    return tot
ks = [(s, e) for s, e, k, n, q in sel if k == "K"]; cs = [(s, e) for s, e, k, n, q in sel if k == "C"]
print(f"last 70 ms: kernels busy {union(ks) / 1e6:.1f} ms (sum {sum(e - s for s, e in ks) / 1e6:.1f}), copies busy {union(cs) / 1e6:.1f} ms, either {union(ks + cs) / 1e6:.1f} ms")
by = {}
for s, e, k, n, q in sel:
    by.setdefault(k + " " + n, [0, 0]); by[k + " " + n][0] += e - s; by[k + " " + n][1] += 1
for n, (t, c) in sorted(by.items(), key=lambda x: -x[1][0])[:14]: print(f"  {t / 1e6:7.2f} ms  {c:5d} x  {n}")
w0 = end - 12_000_000
for s, e, k, n, q in sel:
    if s >= w0 and (e - s) > 30_000: print(f"{(s - w0) / 1e6:8.3f} ms  +{(e - s) / 1e6:7.3f} ms  {k} {n} {q}")
PY
