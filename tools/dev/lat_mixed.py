"""DEV: latency of one small structure through freesasa_calc_coord after the shared context has seen large batches."""
import sys, time, numpy as np
sys.path.insert(0, '.')
import freesasa_amd as fa, tools
g = np.load('tests/golden/1ubq.npz')
xyz, r = g['xyz'], g['radii']
def t(f, n=30):
    f(); t0 = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t0) / n * 1e6
def show(tag):
    print(tag, "calc_coord LR %.0f us  SR %.0f us" % (t(lambda: fa.calc_coord(xyz, r, fa.LEE_RICHARDS)), t(lambda: fa.calc_coord(xyz, r, fa.SHRAKE_RUPLEY))), flush=True)
show("fresh:")
parts = [tools.globule(10000, 500 + k) for k in range(50)]
X = np.concatenate([p[0] for p in parts]); R = np.concatenate([p[1] for p in parts])
offs = np.arange(51, dtype=np.int64) * 10000
fa.calc_batch(X, R, offs, fa.LEE_RICHARDS, resolution=20); fa.calc_batch(X, R, offs, fa.SHRAKE_RUPLEY, resolution=100)
show("after 50 globules:")
X, R, offs = tools.coil_batch(100, 10000, seed0=1000)
fa.calc_batch(X, R, offs, fa.LEE_RICHARDS, resolution=20); fa.calc_batch(X, R, offs, fa.SHRAKE_RUPLEY, resolution=100)
show("after 100 coils:")
