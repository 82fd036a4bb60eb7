import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import freesasa_amd as fa, tools
g = np.load('tests/golden/1ubq.npz')
dev = torch.device('cuda:0')
dx, dr = torch.from_numpy(g['xyz'].reshape(-1)).to(dev), torch.from_numpy(g['radii']).to(dev)
out = torch.empty(len(g['radii']), dtype=torch.float64, device=dev)
offs = np.array([0, len(g['radii'])], dtype=np.int64)
ctx = fa.GpuContext(0, timing=True)
for alg in ('lr', 'sr'):
    for i in range(4):
        t0 = time.perf_counter()
        if alg == 'lr': ctx.lee_richards(dx.data_ptr(), dr.data_ptr(), offs, out.data_ptr())
        else: ctx.shrake_rupley(dx.data_ptr(), dr.data_ptr(), offs, out.data_ptr())
        dt = time.perf_counter() - t0
        st = ctx.stats()
        print(alg, i, "wall %.0f us kernel %.0f us prep %.0f us total %.0f us" % (dt*1e6, st['ms_kernel']*1e3, st['ms_prep']*1e3, st['ms_total']*1e3), st['tile_atoms'], st['lds_bytes'], st['fallback_tiles'])
import os
# globule configs
parts = [tools.globule(10000, 500 + k) for k in range(100)]
xyz = np.concatenate([p[0] for p in parts]); r = np.concatenate([p[1] for p in parts])
offs = np.arange(101, dtype=np.int64) * 10000
dx, dr = torch.from_numpy(xyz).to(dev), torch.from_numpy(r).to(dev)
out = torch.empty(len(r), dtype=torch.float64, device=dev)
for spec in ("0,0,-1,0", "3,0,-1,16", "3,224,-1,16", "3,256,-1,16", "4,256,-1,16", "2,0,-1,16", "5,256,-1,16"):
    os.environ["FREESASA_AMD_LR2"] = spec
    ctx = fa.GpuContext(0, timing=True)
    best = 1e9
    for i in range(4):
        t0 = time.perf_counter()
        ctx.lee_richards(dx.data_ptr(), dr.data_ptr(), offs, out.data_ptr())
        best = min(best, time.perf_counter() - t0)
    st = ctx.stats()
    print('globule L&R %-14s %.4g atoms/s kernel_ms %.3f total %.3f fallback %d lds %d TA %d' % (spec, len(r)/best, st['ms_kernel'], st['ms_total'], st['fallback_tiles'], st['lds_bytes'], st['tile_atoms']))
    ctx.close()
