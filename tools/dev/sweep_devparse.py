"""DEV: the file sweep with the device-side parser: workers per device, batch size."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, freesasa_amd as fa, bench
from freesasa_amd import ingest
d = "/tmp/fsbench/sweep_files"
if not os.path.isdir(d):            # bench.py's own copies of the reference's structure files (tests/golden/structures)
    import shutil
    os.makedirs(d, exist_ok=True)
    pdb_dir, cif_dir = os.path.join(bench.ROOT, "tests", "golden", "pdb"), os.path.join(bench.ROOT, "tests", "golden", "cif")
    srcs = [os.path.join(pdb_dir, nm + ".pdb") for nm in bench.PDB_NAMES] + sorted(os.path.join(cif_dir, f) for f in os.listdir(cif_dir) if f.endswith(".cif"))[:4]
    for k in range(163):
        for sp in srcs: shutil.copyfile(sp, os.path.join(d, f"{k:04d}_{os.path.basename(sp)}"))
paths = sorted(os.path.join(d, f) for f in os.listdir(d)) * 4
n = None
for opt, tag in ((ingest.PARSE_ON_DEVICE, "device"), (0, "host")):
    for devs in ([0], [0, 0], [0, 0, 0], [0, 0, 0, 0]):
        for ba in (500_000, 1_000_000, 2_000_000):
            fa.sweep_files(paths[:400], devices=devs, ingest_options=opt, batch_atoms=ba)
            best = 1e9
            for _ in range(2):
                t0 = time.perf_counter(); r = fa.sweep_files(paths, devices=devs, ingest_options=opt, batch_atoms=ba); best = min(best, time.perf_counter() - t0)
            n = int(r[2].sum())
            print(f"{tag:6s} workers {len(devs)} batch {ba:8d}: {n / best:.3e} atoms/s ({best * 1e3:.1f} ms)", flush=True)
