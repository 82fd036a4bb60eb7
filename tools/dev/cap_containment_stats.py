"""DEV (CPU, numpy): how many neighbor caps lie inside another neighbor's cap (what lr2_prune_contained drops), by the number
of largest caps a neighbor is tested against; records and arc-weighted (a cap's share of the slices).  Random coil and the
reference's 1a0q.  profiles/r06_cap_containment_stats.txt."""
import sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import tools
from scipy.spatial import cKDTree
def wfrac(xyz, r, probe=1.4, margin=1e-5, tops=(1,2,3,4,6,8,12,99)):
    R = r + probe
    tree = cKDTree(xyz)
    W = 0.0; N = 0; Wt = {t: 0.0 for t in tops}; Nt = {t: 0 for t in tops}
    for i in range(0, len(R), 5):
        js = [j for j in tree.query_ball_point(xyz[i], R[i] + R.max()) if j != i]
        v = xyz[js] - xyz[i]; d = np.linalg.norm(v, axis=1)
        ok = d < R[i] + R[js]
        v, d, Rj = v[ok], d[ok], R[js][ok]
        if len(d) == 0: continue
        n = v / d[:, None]
        c = np.clip((d * d + R[i] ** 2 - Rj ** 2) / (2 * d * R[i]), -1, 1)
        th = np.arccos(c)
        ph = np.arccos(np.clip(n[:, 2], -1, 1))
        zhi = np.cos(np.maximum(ph - th, 0)); zlo = np.cos(np.minimum(ph + th, np.pi))
        w = (zhi - zlo) / 2
        ang = np.arccos(np.clip(n @ n.T, -1, 1))
        cont = ang + th[:, None] <= th[None, :] - margin
        np.fill_diagonal(cont, False)
        W += w.sum(); N += len(d)
        order = np.argsort(-th)
        for t in tops:
            p = cont[:, order[:t]].any(axis=1)
            Wt[t] += w[p].sum(); Nt[t] += p.sum()
    return {t: (round(Nt[t] / N, 3), round(Wt[t] / W, 3)) for t in tops}
xyz, r = tools.coil(6000, 1234)[:2]
print("coil (records, arc-weighted) by top-K:", wfrac(np.asarray(xyz).reshape(-1, 3), np.asarray(r)))
from conftest import load_golden
g = load_golden("1a0q")
print("1a0q:", wfrac(np.asarray(g["xyz"]).reshape(-1, 3), np.asarray(g["radii"])))
