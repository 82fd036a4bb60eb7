"""DEV: where a step of the single 200 000-atom Shrake-Rupley call goes (configs[1] proxy): wall clock per call, the
context's own GPU events (prep / kernel / total), and - under rocprofv3 --kernel-trace - the kernels' timeline."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import freesasa_amd as fa
import tools

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
dev = torch.device("cuda:0")
bx, br = tools.globule(nb, 77)
offs = np.array([0, nb], dtype=np.int64)
dbx, dbr = torch.from_numpy(bx).to(dev), torch.from_numpy(br).to(dev)
dbo = torch.empty(nb, dtype=torch.float64, device=dev)
dbc = torch.empty(nb, dtype=torch.int32, device=dev)
for timing in (True, False):
    ctx = fa.GpuContext(0, timing=timing)
    call = lambda: ctx.shrake_rupley(dbx.data_ptr(), dbr.data_ptr(), offs, dbo.data_ptr(), dbc.data_ptr(), 0, probe=1.4, n_points=100)
    for _ in range(5): call()
    t = []
    for _ in range(40):
        t0 = time.perf_counter(); call(); t.append(time.perf_counter() - t0)
    st = ctx.stats()
    print(f"timing={timing}: wall per call median {1e3 * np.median(t):.4f} ms, best {1e3 * min(t):.4f}; events: prep {st['ms_prep']:.4f} kernel {st['ms_kernel']:.4f} total {st['ms_total']:.4f}")
    ctx.close()
