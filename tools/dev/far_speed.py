"""DEV: throughput of a batch far from the origin (every tile through the second launch's walking build) against the same batch at the origin."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import freesasa_amd as fa, tools
bx, br, offs = tools.coil_batch(200, 10000, seed0=1000)
dev = torch.device("cuda:0")
for shift in (0.0, 5000.0):
    x = bx + np.array([0.0, 0.0, shift])
    d_xyz, d_r = torch.from_numpy(np.ascontiguousarray(x)).to(dev), torch.from_numpy(br).to(dev)
    d_out = torch.empty(len(br), dtype=torch.float64, device=dev)
    for ns in (20, 100):
        ctx = fa.GpuContext(0, timing=True)
        for _ in range(3): ctx.lee_richards(d_xyz.data_ptr(), d_r.data_ptr(), offs, d_out.data_ptr(), 0, probe=1.4, n_slices=ns)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): ctx.lee_richards(d_xyz.data_ptr(), d_r.data_ptr(), offs, d_out.data_ptr(), 0, probe=1.4, n_slices=ns)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        print(f"shift {shift:g} A, {ns} slices: {len(br) / dt:.3e} atoms/s ({1e3 * dt:.2f} ms), fallback tiles {ctx.stats()['fallback_tiles']}", flush=True)
        ctx.close()
