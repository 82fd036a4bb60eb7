"""First call on a fresh context vs steady state (the density estimate should make them about equal)."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
import freesasa_amd as fa, tools
dev = torch.device('cuda:0')
def batch(gen, n, k):
    parts = [gen(n, 500 + i) for i in range(k)]
    xyz = np.concatenate([p[0] for p in parts]); r = np.concatenate([p[1] for p in parts])
    return xyz, r, np.arange(k + 1, dtype=np.int64) * n
for name, gen in (('coil', tools.coil), ('globule', tools.globule)):
    xyz, r, offs = batch(gen, 10000, 60)
    dx, dr = torch.from_numpy(xyz).to(dev), torch.from_numpy(r).to(dev)
    out = torch.empty(len(r), dtype=torch.float64, device=dev)
    for alg in ('lr', 'sr'):
        ctx = fa.GpuContext(0, timing=True)
        ks = []
        for i in range(4):
            (ctx.lee_richards if alg == 'lr' else ctx.shrake_rupley)(dx.data_ptr(), dr.data_ptr(), offs, out.data_ptr())
            st = ctx.stats(); ks.append((st['ms_kernel'], st['fallback_tiles'], st['lds_bytes']))
        print(name, alg, 'kernel_ms/fallback/lds per call:', ' | '.join('%.3f/%d/%d' % k for k in ks), flush=True)
        ctx.close()
