"""One-off parity sweep against the REAL reference library (oracle/_ref, the checker; not part of the product):
per-atom L&R-20 / L&R-100 areas and S&R-100 counts of many synthetic structures, GPU engine vs reference.
usage: python tools/deep_parity.py [n_coils] [n_globules]   -> one JSON line"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import freesasa_amd as fa, tools, oracle

n_coils = int(sys.argv[1]) if len(sys.argv) > 1 else 120
n_glob = int(sys.argv[2]) if len(sys.argv) > 2 else 24
ref = oracle.Reference()
ref.lib.freesasa_set_verbosity(2)
parts = [tools.coil(10000, 4000 + k) for k in range(n_coils)] + [tools.globule(10000, 700 + k) for k in range(n_glob)]
xyz = np.concatenate([p[0] for p in parts]); r = np.concatenate([p[1] for p in parts])
offs = np.arange(len(parts) + 1, dtype=np.int64) * 10000
out = {"structures": len(parts), "atoms": int(len(r))}
t0 = time.perf_counter()
for ns in (20, 100):
    got, _, _ = fa.calc_batch(xyz, r, offs, fa.LEE_RICHARDS, 1.4, ns)
    worst, worst_rel = 0.0, 0.0
    for k, (px, pr) in enumerate(parts if ns == 20 else parts[::6]):
        kk = k if ns == 20 else 6 * k
        want = ref.calc_coord(px, pr, oracle.LEE_RICHARDS, 1.4, n_slices=ns, n_threads=1)[0]
        d = np.abs(got[offs[kk]:offs[kk + 1]] - want)
        worst = max(worst, float(d.max()))
        worst_rel = max(worst_rel, float((d / np.maximum(want, 1.0)).max()))
    out[f"lr{ns}_max_abs_dsasa"] = worst
    out[f"lr{ns}_max_rel"] = worst_rel
    out[f"lr{ns}_structures_checked"] = len(parts) if ns == 20 else len(parts[::6])
sr, cnt, _ = fa.calc_batch(xyz, r, offs, fa.SHRAKE_RUPLEY, 1.4, 100)
bad_counts = bad_areas = 0
for k, (px, pr) in enumerate(parts):
    want = ref.calc_coord(px, pr, oracle.SHRAKE_RUPLEY, 1.4, n_points=100, n_threads=1)[0]
    sl = slice(offs[k], offs[k + 1])
    bad_areas += int(np.count_nonzero(sr[sl] != want))
out["sr100_atoms_with_different_area"] = bad_areas
out["seconds"] = time.perf_counter() - t0
print(json.dumps(out))
