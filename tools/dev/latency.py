"""DEV: latency of freesasa_calc_coord on small structures under FREESASA_AMD_SMALL_FUSED limits."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
lim = sys.argv[1]
os.environ["FREESASA_AMD_SMALL_FUSED"] = lim
import freesasa_amd as fa, tools
from freesasa_amd import ingest
import oracle
o = oracle.Oracle()
for nm, (x, r) in (("1ubq", (None, None)), ("1a0q", (None, None)), ("coil2000", tools.coil(2000, 5)), ("coil4000", tools.coil(4000, 5)), ("coil8000", tools.coil(8000, 6)), ("coil16000", tools.coil(16000, 7))):
    if x is None:
        b = ingest.load_pdb_files([f"tests/golden/pdb/{nm}.pdb"]); x, r = np.ascontiguousarray(b.xyz.reshape(-1, 3)), np.ascontiguousarray(b.radii)
    for alg, kw in ((fa.LEE_RICHARDS, {"n_slices": 20}), (fa.SHRAKE_RUPLEY, {"n_points": 100})):
        for _ in range(5): got = fa.calc_coord(x, r, alg, **kw)
        ts = []
        for _ in range(40):
            t0 = time.perf_counter(); fa.calc_coord(x, r, alg, **kw); ts.append(time.perf_counter() - t0)
        want = o.lee_richards(x, r, 1.4, 20) if alg == fa.LEE_RICHARDS else o.shrake_rupley(x, r, 1.4, 100)[0]
        print(f"limit {lim} {nm} n={len(r)} alg={alg}: median {1e6*np.median(ts):.1f} us best {1e6*min(ts):.1f} us  maxdiff {np.max(np.abs(got[0]-want)):.2g}", flush=True)
