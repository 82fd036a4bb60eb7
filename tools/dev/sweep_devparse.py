"""DEV: the file sweep with the device-side parser: workers per device, batch size."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, freesasa_amd as fa, bench
from freesasa_amd import ingest
d = "/tmp/fsbench/sweep_files"
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
