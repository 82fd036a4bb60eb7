"""DEV: S&R kernel time on the 200k-atom globule and on 100 x 10k globules (kernel ms, min of 6)."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import freesasa_amd as fa, tools
dev = torch.device('cuda:0')
def run(xyz, r, offs, tag):
    dx, dr = torch.from_numpy(xyz).to(dev), torch.from_numpy(r).to(dev)
    out = torch.empty(len(r), dtype=torch.float64, device=dev); cnt = torch.empty(len(r), dtype=torch.int32, device=dev)
    ctx = fa.GpuContext(0, timing=True)
    ks = []
    for i in range(7):
        ctx.shrake_rupley(dx.data_ptr(), dr.data_ptr(), offs, out.data_ptr(), cnt.data_ptr())
        ks.append(ctx.stats()['ms_kernel'])
    print(tag, "kernel_ms %.4f (first %.4f)  atoms/s(kernel) %.4g  checksum %d" % (min(ks[1:]), ks[0], len(r) / min(ks[1:]) * 1e3, int(cnt.sum().item())), flush=True)
    ctx.close()
x, r = tools.globule(200000, 77)
run(x, r, np.array([0, 200000], dtype=np.int64), "globule 200k:")
parts = [tools.globule(10000, 500 + k) for k in range(100)]
run(np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts]), np.arange(101, dtype=np.int64) * 10000, "100 x 10k:  ")
xyz, r, offs = tools.coil_batch(100, 10000, seed0=1000)
run(xyz, r, offs, "coils 100x10k:")
