"""DEV: per-call wall clock of the real_pdb_lr20 batch on one context, event timing toggled as bench.py's run() does."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import freesasa_amd as fa, bench
dev = torch.device("cuda:0")
px, pr, poffs, per, reps = bench.real_pdb_batch()
dpx, dpr = torch.from_numpy(px).to(dev), torch.from_numpy(pr).to(dev)
dpo = torch.empty(len(pr), dtype=torch.float64, device=dev)
ctx = fa.GpuContext(0, timing=True)
def calls(k, tag):
    ts = []
    for _ in range(k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ctx.lee_richards(dpx.data_ptr(), dpr.data_ptr(), poffs, dpo.data_ptr(), 0, probe=1.4, n_slices=20)
        torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
    st = ctx.stats()
    print(f"{tag}: per call ms " + " ".join(f"{t:.2f}" for t in ts) + f" | kernel_ms {st['ms_kernel']:.2f} fallback {st['fallback_tiles']} TA {st['tile_atoms']}", flush=True)
calls(5, "timing on ")
ctx.set_timing(False); calls(6, "timing off")
ctx.set_timing(True); calls(4, "timing on ")
ctx.set_timing(False); calls(4, "timing off")
# back to back without a device synchronize in between (what bench.py's loop does)
t0 = time.perf_counter()
for _ in range(10):
    ctx.lee_richards(dpx.data_ptr(), dpr.data_ptr(), poffs, dpo.data_ptr(), 0, probe=1.4, n_slices=20)
torch.cuda.synchronize(); print(f"10 calls back to back, timing off: {1e2 * (time.perf_counter() - t0):.2f} ms per call")
ctx.close()
