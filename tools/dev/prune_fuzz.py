"""DEV: the Lee-Richards tile kernel with and without its contained-caps phase (lr2_prune_contained, P1.5) on random inputs (GPU):
structures of random kind and size, radii scalings, shifts, probe radius and slice count; every area must be the same BITS
with the phase off (FREESASA_AMD_PRUNE=0), as shipped, and forced on.  One exception is allowed and counted: inputs with atoms
of more than 100 neighbors (three times protein density: lattices of 1.2 A, radii scaled by 3).  There an atom may have arcs in
so many disjoint pieces that the tile kernel's arc stack is too short in every launch, and the LAST launch does it with the
first-generation kernel's arithmetic; with contained arcs gone the stack may suffice, and the atom gets the tile kernel's bits
(the two agree to ~1e-11 A^2: asserted 1e-9 here).  usage: prune_fuzz.py [rounds] [seed]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import freesasa_amd as fa, tools

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2026)
dev = torch.device("cuda:0")


def run(xyz, r, offs, ns, probe, env):
    if env is None: os.environ.pop("FREESASA_AMD_PRUNE", None)
    else: os.environ["FREESASA_AMD_PRUNE"] = env
    dx, dr = torch.from_numpy(xyz.reshape(-1)).to(dev), torch.from_numpy(r).to(dev)
    out = torch.empty(len(r), dtype=torch.float64, device=dev)
    ctx = fa.GpuContext(0)
    try:
        for _ in range(2):   # (second call: the learnt tile shape)
            ctx.lee_richards(dx.data_ptr(), dr.data_ptr(), offs, out.data_ptr(), 0, probe=probe, n_slices=ns)
        st = ctx.stats()
    finally:
        ctx.close()
    return out.cpu().numpy(), st


atoms = moved = refused = 0
t0 = time.time()
for k in range(rounds):
    kind = rng.integers(0, 4)
    nst = int(rng.integers(1, 12))
    parts = []
    for s in range(nst):
        n = int(rng.integers(1, 6000))
        if kind == 0: x, r = tools.coil_batch(1, n, seed0=int(rng.integers(1, 1 << 30)))[:2]; x = x.reshape(-1, 3)
        elif kind == 1: x, r = tools.globule(n, int(rng.integers(1, 1 << 30))); x = x.reshape(-1, 3)
        elif kind == 2: x, r = tools.globule(n, int(rng.integers(1, 1 << 30)), spacing=float(rng.uniform(1.2, 4.0))); x = x.reshape(-1, 3)
        else:
            x = rng.uniform(0, max(4.0, (n * 18.0) ** (1 / 3)), size=(n, 3)); r = rng.uniform(1.0, 2.2, size=n)
        if rng.random() < 0.3: r = r * rng.uniform(0.3, 3.0, size=len(r))
        if rng.random() < 0.3: x = x + rng.uniform(-1, 1, size=3) * 10 ** rng.uniform(0, 3.5)
        if rng.random() < 0.2: x = np.round(x, int(rng.integers(0, 3)))     # atoms on a lattice: equal coordinates, directions along the axes (beta's cut)
        parts.append((np.ascontiguousarray(x, dtype=np.float64), np.ascontiguousarray(r, dtype=np.float64)))
    xyz = np.concatenate([p[0] for p in parts]); r = np.concatenate([p[1] for p in parts])
    offs = np.concatenate([[0], np.cumsum([len(p[1]) for p in parts])]).astype(np.int64)
    ns = int(rng.choice([1, 7, 20, 20, 20, 32, 33, 50, 64, 100, 100, 128, 200, 256]))
    probe = float(rng.choice([0.0, 1.0, 1.4, 1.4, 2.5]))
    if k < int(os.environ.get("FUZZ_FROM", "0")) or (os.environ.get("FUZZ_ONLY") and str(k) not in os.environ["FUZZ_ONLY"].split(",")): continue
    if os.environ.get("FUZZ_TRACE"): print(f"round {k}: kind {kind} structs {nst} atoms {len(r)} slices {ns} probe {probe}", flush=True)
    try:
        a0, st0 = run(xyz, r, offs, ns, probe, "0")
    except RuntimeError as e:   # an input beyond the engine's stated capacity (4096 neighbors per atom): reported, not computed
        if "more neighbors than" not in str(e): raise
        refused += 1
        continue
    if os.environ.get("FUZZ_TRACE"): print("  off done", flush=True)
    for env in (None, "4"):
        a1, st1 = run(xyz, r, offs, ns, probe, env)
        if not np.array_equal(a0, a1, equal_nan=True):
            bad = np.nonzero(~((a0 == a1) | (np.isnan(a0) & np.isnan(a1))))[0]
            worst = float(np.nanmax(np.abs(a0 - a1)))
            if st0["max_neighbors"] > 100 and worst < 1e-9 and not np.any(np.isnan(a0) != np.isnan(a1)):
                moved += 1
                continue
            print(f"MISMATCH round {k}: kind {kind} structs {nst} atoms {len(r)} slices {ns} probe {probe} FREESASA_AMD_PRUNE={env}: {len(bad)} atoms differ, first {bad[:5]}, "
                  f"max {worst:.3g}; tile atoms {st0['tile_atoms']} / {st1['tile_atoms']}; max neighbors {st0['max_neighbors']}", flush=True)
            sys.exit(1)
    atoms += len(r)
print(f"{rounds} rounds, {atoms} atoms: identical areas with the contained caps dropped (as shipped and forced on) and not; {moved} runs on inputs with more than "
      f"100 neighbors per atom in which atoms moved from the last launch into the tile kernel (< 1e-9 A^2); {refused} inputs refused (an atom with more than 4096 neighbors), {time.time() - t0:.0f} s")
