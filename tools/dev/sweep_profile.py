"""DEV: where the file sweep's wall clock goes (FREESASA_AMD_SWEEP_PROFILE; parser on the device)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["FREESASA_AMD_SWEEP_PROFILE"] = "1"
import numpy as np, freesasa_amd as fa, bench
from freesasa_amd import ingest
d = "/tmp/fsbench/sweep_files"
if not os.path.isdir(d):
    import shutil
    os.makedirs(d, exist_ok=True)
    pdb_dir, cif_dir = os.path.join(bench.ROOT, "tests", "golden", "pdb"), os.path.join(bench.ROOT, "tests", "golden", "cif")
    srcs = [os.path.join(pdb_dir, nm + ".pdb") for nm in bench.PDB_NAMES] + sorted(os.path.join(cif_dir, f) for f in os.listdir(cif_dir) if f.endswith(".cif"))[:4]
    for k in range(163):
        for sp in srcs: shutil.copyfile(sp, os.path.join(d, f"{k:04d}_{os.path.basename(sp)}"))
paths = sorted(os.path.join(d, f) for f in os.listdir(d)) * 4
nbytes = sum(os.path.getsize(p) for p in paths)
for devs in ([0, 0], [0, 0, 0]):
    for ba in (500_000, 1_000_000, 1_500_000, 2_000_000):
        fa.sweep_files(paths[:400], devices=devs, ingest_options=ingest.PARSE_ON_DEVICE, batch_atoms=ba)
        for _ in range(2):
            t0 = time.perf_counter(); r = fa.sweep_files(paths, devices=devs, ingest_options=ingest.PARSE_ON_DEVICE, batch_atoms=ba); dt = time.perf_counter() - t0
            n = int(r[2].sum())
            print(f"workers {len(devs)} batch {ba:8d}: {n / dt:.3e} atoms/s ({dt * 1e3:.1f} ms, {nbytes / dt / 1e9:.1f} GB/s of text)", flush=True)
