"""DEV: does freesasa_calc_coord's latency change when torch is alive in the process?"""
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
show("no torch:")
import torch
show("torch imported:")
x = torch.zeros(1000, device="cuda:0"); torch.cuda.synchronize()
show("torch has a tensor on the device:")
ctx = fa.GpuContext(0, timing=True)
show("a second context exists (timing on):")
parts = [tools.globule(10000, 500 + k) for k in range(20)]
X = np.concatenate([p[0] for p in parts]); R = np.concatenate([p[1] for p in parts]); offs = np.arange(21, dtype=np.int64) * 10000
dx, dr = torch.from_numpy(X).to("cuda:0"), torch.from_numpy(R).to("cuda:0"); out = torch.empty(len(R), dtype=torch.float64, device="cuda:0")
ctx.shrake_rupley(dx.data_ptr(), dr.data_ptr(), offs, out.data_ptr())
show("after a device-resident S&R batch on it:")
ctx.lee_richards(dx.data_ptr(), dr.data_ptr(), offs, out.data_ptr())
show("after an L&R batch on it:")
