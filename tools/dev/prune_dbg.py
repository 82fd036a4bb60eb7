import sys, numpy as np, torch
sys.path.insert(0, '.')
import freesasa_amd as fa, tools
parts = [tools.globule(10000, 500 + k) for k in range(30)]
xyz = np.concatenate([p[0] for p in parts]); r = np.concatenate([p[1] for p in parts])
offs = np.arange(31, dtype=np.int64) * 10000
dev = torch.device('cuda:0')
dx, dr = torch.from_numpy(xyz).to(dev), torch.from_numpy(r).to(dev)
out = torch.empty(len(r), dtype=torch.float64, device=dev)
ctx = fa.GpuContext(0, timing=True)
for i in range(4):
    ctx.lee_richards(dx.data_ptr(), dr.data_ptr(), offs, out.data_ptr())
    st = ctx.stats(); print('call', i, 'kernel_ms %.3f lds %d fallback %d' % (st['ms_kernel'], st['lds_bytes'], st['fallback_tiles']), flush=True)
