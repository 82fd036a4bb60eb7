import sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import tools
from scipy.spatial import cKDTree
rng = np.random.default_rng(1)
def study(xyz, r, probe=1.4, margin=1e-5):
    R = r + probe
    tree = cKDTree(xyz)
    W = 0.0; N = 0
    res = {}
    def acc(key, w, p): 
        a = res.setdefault(key, [0, 0.0, 0, 0]); a[0] += p.sum(); a[1] += w[p].sum()
    trips = {}
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
        w = (np.cos(np.maximum(ph - th, 0)) - np.cos(np.minimum(ph + th, np.pi))) / 2
        ang = np.arccos(np.clip(n @ n.T, -1, 1))
        cont = ang + th[:, None] <= th[None, :] - margin
        np.fill_diagonal(cont, False)
        up = n[:, 1] > 0
        same = up[:, None] == up[None, :]
        right = n[:, 0] > 0
        sameR = same | (right[:, None] & right[None, :])
        W += w.sum(); N += len(d)
        order = np.argsort(-th)
        acc("all pairs, circular", w, cont.any(axis=1))
        acc("all pairs, same side|right", w, (cont & sameR).any(axis=1))
        acc("all pairs, same side only", w, (cont & same).any(axis=1))
        for K in (3, 4, 6):
            # per side top-K
            m = np.zeros_like(cont)
            for side in (True, False):
                idx = [k for k in order if up[k] == side][:K]
                m[:, idx] = True
            acc(f"per-side top-{K}, same side only", w, (cont & same & m).any(axis=1))
            for T in (0.4, 0.5, 0.6):
                m = np.zeros_like(cont); tr = 0
                for side in (True, False):
                    idx = [k for k in rng.permutation(len(c)) if up[k] == side and c[k] <= T][:K]
                    m[:, idx] = True; tr = max(tr, len(idx))
                acc(f"per-side c<={T} first {K} by arrival", w, (cont & same & m).any(axis=1))
                trips.setdefault((T, K), []).append(tr)
    for k, a in res.items(): print(f"  {k:45s} records {a[0]/N:.3f} arcs {a[1]/W:.3f}")
xyz, r = tools.coil(6000, 1234)[:2]
print("coil"); study(np.asarray(xyz).reshape(-1, 3), np.asarray(r))
from conftest import load_golden
g = load_golden("1a0q")
print("1a0q"); study(np.asarray(g["xyz"]).reshape(-1, 3), np.asarray(g["radii"]))
