"""DEV: end-to-end rate (host arrays -> host per-atom areas) of freesasa_gpu_calc_batch_pipelined on the headline batch."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import freesasa_amd as fa, tools
structs = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
xyz, r, offs = tools.coil_batch(structs, 10000)
n = len(r)
px, pr = torch.from_numpy(xyz).pin_memory(), torch.from_numpy(r).pin_memory()
ps, pt = torch.empty(n, dtype=torch.float64).pin_memory(), torch.empty(structs, dtype=torch.float64).pin_memory()
ref = None
for pinned in (True, False):
    for lanes, chunk in ((1, 0), (2, 0), (3, 0), (4, 0), (3, 625000), (4, 2500000), (6, 0)):
        a = (px.numpy(), pr.numpy()) if pinned else (xyz, r)
        out = (ps.numpy(), None, pt.numpy()) if pinned else None
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter()
            res = fa.calc_batch_pipelined(a[0], a[1], offs, lanes=lanes, chunk_atoms=chunk, out=out)
            best = min(best, time.perf_counter() - t0)
        if ref is None: ref = res[0].copy()
        print(f"pinned={pinned} lanes={lanes} chunk={chunk or 1250000}: {best*1e3:8.2f} ms  {n/best:.4g} atoms/s  ({40*n/best/1e9:.1f} GB/s over PCIe)  maxdiff {np.max(np.abs(res[0]-ref)):.2g}", flush=True)
