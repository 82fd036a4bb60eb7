"""DEV: the headline batch through one context (two batches in flight on ONE stream) against two contexts on two
streams, alternating (the next batch's cell sort and the last workgroups of this batch's tile kernel can overlap)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import freesasa_amd as fa
import tools
structs, atoms, steps = 1000, 10000, int(sys.argv[1]) if len(sys.argv) > 1 else 20
xyz, r, offs = tools.coil_batch(structs, atoms, seed0=1000, cache_dir="/tmp")
dev = torch.device("cuda:0")
d_xyz, d_r = torch.from_numpy(xyz).to(dev), torch.from_numpy(r).to(dev)
n = int(offs[-1])
def run(nctx, prio=False):
    streams = [torch.cuda.Stream(device=dev, priority=(-1 if (prio and k % 2) else 0)) for k in range(nctx)]
    ctxs = [fa.GpuContext(0, stream=s.cuda_stream, timing=False) for s in streams]
    outs = [(torch.empty(n, dtype=torch.float64, device=dev), torch.empty(structs, dtype=torch.float64, device=dev)) for _ in range(nctx)]
    def step(k):
        c = ctxs[k % nctx]; o, t = outs[k % nctx]
        c.lee_richards_async(d_xyz.data_ptr(), d_r.data_ptr(), offs, o.data_ptr(), t.data_ptr(), probe=1.4, n_slices=20)
    for k in range(4): step(k)
    for c in ctxs: c.wait()
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for k in range(steps): step(k)
        for c in ctxs: c.wait()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / steps)
    same = all(torch.equal(outs[0][0], o[0]) for o in outs)
    for c in ctxs: c.close()
    return best * 1e3, same
for nctx, prio in ((1, False), (2, False), (2, True), (4, True), (2, False), (2, True)):
    ms, same = run(nctx, prio)
    print(f"contexts {nctx} prio {prio}: {ms:.3f} ms per step  {n / ms * 1e3:.4e} atoms/s  outputs equal {same}", flush=True)
