import sys, time, numpy as np
sys.path.insert(0, '.')
import freesasa_amd as fa
g = np.load('tests/golden/1ubq.npz')
xyz, r = g['xyz'], g['radii']
offs = np.array([0, len(r)], dtype=np.int64)
def t(f, n=30):
    f(); t0 = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t0) / n * 1e6
print("calc_coord LR %.0f us" % t(lambda: fa.calc_coord(xyz, r, fa.LEE_RICHARDS)))
print("calc_coord SR %.0f us" % t(lambda: fa.calc_coord(xyz, r, fa.SHRAKE_RUPLEY)))
print("calc_batch LR %.0f us" % t(lambda: fa.calc_batch(xyz, r, offs, fa.LEE_RICHARDS, resolution=20)))
print("calc_batch SR %.0f us" % t(lambda: fa.calc_batch(xyz, r, offs, fa.SHRAKE_RUPLEY, resolution=100)))
print("test_points(100) %.0f us" % t(lambda: fa.test_points(100)))
print("calc_batch SR 20 pts %.0f us" % t(lambda: fa.calc_batch(xyz, r, offs, fa.SHRAKE_RUPLEY, resolution=20)))
