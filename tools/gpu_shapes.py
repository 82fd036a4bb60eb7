"""DEV: one process, many launch shapes of the L&R kernel on the headline batch geometry (kernel ms per step).
usage: python tools/gpu_shapes.py [[g]structs] [spec ...]   spec = LR1 | TA,pool,ds,refill"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import freesasa_amd as fa, tools

geom = "coil"
if len(sys.argv) > 1 and sys.argv[1].startswith("g"):  # g100 = 100 protein-like globules of 10k atoms
    geom = "globule"; sys.argv[1] = sys.argv[1][1:]
elif len(sys.argv) > 1 and sys.argv[1].startswith("p"):  # p84 = the reference's PDB entries x 84 (1e6 atoms; bench.real_pdb_batch)
    geom = "pdb"; sys.argv[1] = sys.argv[1][1:]
structs = int(sys.argv[1]) if len(sys.argv) > 1 else 300
specs = sys.argv[2:] or ["LR1", "0,0,-1,0", "4,0,-1,16", "5,0,-1,16", "6,0,-1,8", "6,0,-1,24", "6,0,2,16", "3,0,-1,16"]
if geom == "coil":
    xyz, r, offs = tools.coil_batch(structs, 10000, seed0=1000)
elif geom == "pdb":
    import bench
    xyz, r, offs, _, reps = bench.real_pdb_batch(structs * 12000)
    structs = len(offs) - 1
else:
    parts = [tools.globule(10000, 500 + k) for k in range(structs)]
    xyz = np.concatenate([p[0] for p in parts]); r = np.concatenate([p[1] for p in parts])
    offs = np.arange(structs + 1, dtype=np.int64) * 10000
dev = torch.device("cuda:0")
dx, dr = torch.from_numpy(xyz).to(dev), torch.from_numpy(r).to(dev)
out = torch.empty(len(r), dtype=torch.float64, device=dev)
tot = torch.empty(structs, dtype=torch.float64, device=dev)
ref = None
for spec in specs:
    os.environ.pop("FREESASA_AMD_LR1", None); os.environ.pop("FREESASA_AMD_LR2", None)
    os.environ.pop("FREESASA_AMD_WPE", None)
    if spec == "LR1": os.environ["FREESASA_AMD_LR1"] = "1"
    else:
        f = spec.split(",")
        os.environ["FREESASA_AMD_LR2"] = ",".join(f[:4])
        if len(f) > 4: os.environ["FREESASA_AMD_WPE"] = f[4]
    ctx = fa.GpuContext(0, timing=True)
    ks = []
    for i in range(5):
        t0 = time.perf_counter()
        ctx.lee_richards(dx.data_ptr(), dr.data_ptr(), offs, out.data_ptr(), tot.data_ptr(), probe=1.4, n_slices=int(os.environ.get("SLICES", "20")))
        dt = time.perf_counter() - t0
        st = ctx.stats()
        ks.append((st["ms_kernel"], st["ms_total"], dt * 1e3))
    res = out.cpu().numpy()
    if ref is None: ref = res
    k = min(x[0] for x in ks[1:]); tt = min(x[1] for x in ks[1:]); wall = min(x[2] for x in ks[1:])
    print(f"{spec:16s} kernel_ms {k:8.3f} (first {ks[0][0]:.3f}) total_ms {tt:8.3f} wall_ms {wall:8.3f} atoms/s(wall) {len(r)/wall*1e3:.4g} "
          f"TA {st['tile_atoms']} lds {st['lds_bytes']} fallback {st['fallback_tiles']} maxnn {st['max_neighbors']} maxdiff_vs_first {np.max(np.abs(res-ref)):.3g}", flush=True)
    ctx.close()
