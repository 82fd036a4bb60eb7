import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import freesasa_amd as fa
g = np.load('tests/golden/1ubq.npz')
xyz, r = g['xyz'], g['radii']
dev = torch.device('cuda:0')
dx, dr = torch.from_numpy(xyz).to(dev), torch.from_numpy(r).to(dev)
out = torch.empty(len(r), dtype=torch.float64, device=dev)
ctx = fa.GpuContext(0, timing=True)
offs = np.array([0, len(r)], dtype=np.int64)
for alg in ('lr', 'sr'):
    f = ctx.lee_richards if alg == 'lr' else ctx.shrake_rupley
    for _ in range(20): f(dx.data_ptr(), dr.data_ptr(), offs, out.data_ptr())
    t0 = time.perf_counter()
    for _ in range(200): f(dx.data_ptr(), dr.data_ptr(), offs, out.data_ptr())
    dt = (time.perf_counter() - t0) / 200
    st = ctx.stats()
    print('%s device-resident call: %.0f us wall; events: prep %.0f us, tile kernels %.0f us' % (alg, 1e6 * dt, 1e3 * st['ms_prep'], 1e3 * st['ms_kernel']))
ctx2 = fa.GpuContext(0, timing=False)
for alg in ('lr', 'sr'):
    f = ctx2.lee_richards if alg == 'lr' else ctx2.shrake_rupley
    for _ in range(20): f(dx.data_ptr(), dr.data_ptr(), offs, out.data_ptr())
    t0 = time.perf_counter()
    for _ in range(200): f(dx.data_ptr(), dr.data_ptr(), offs, out.data_ptr())
    print('%s device-resident call without event timing: %.0f us' % (alg, 1e6 * (time.perf_counter() - t0) / 200))
