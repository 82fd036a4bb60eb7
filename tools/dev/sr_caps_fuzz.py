"""DEV: the third Shrake-Rupley arrangement against the second on random inputs (GPU): structures of random kind and size, radii,
probe radius, number of test points and table resolution; counts and areas must be identical.  usage: sr_caps_fuzz.py [rounds] [seed]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import freesasa_amd as fa, tools

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2026)
dev = torch.device("cuda:0")


def run(xyz, r, offs, npts, probe, env):
    if env is None: os.environ.pop("FREESASA_AMD_SR_CAPS", None)
    else: os.environ["FREESASA_AMD_SR_CAPS"] = env
    dx, dr = torch.from_numpy(xyz.reshape(-1)).to(dev), torch.from_numpy(r).to(dev)
    out = torch.empty(len(r), dtype=torch.float64, device=dev); cnt = torch.empty(len(r), dtype=torch.int32, device=dev)
    ctx = fa.GpuContext(0)
    try:
        ctx.shrake_rupley(dx.data_ptr(), dr.data_ptr(), offs, out.data_ptr(), cnt.data_ptr(), probe=probe, n_points=npts)
        ctx.shrake_rupley(dx.data_ptr(), dr.data_ptr(), offs, out.data_ptr(), cnt.data_ptr(), probe=probe, n_points=npts)   # (second call: the learnt tile shape)
    finally:
        ctx.close()
    return out.cpu().numpy(), cnt.cpu().numpy()


atoms = pairs = 0
t0 = time.time()
for k in range(rounds):
    kind = rng.integers(0, 4)
    ns = int(rng.integers(1, 12))
    parts = []
    for s in range(ns):
        n = int(rng.integers(1, 6000))
        if kind == 0: x, r = tools.coil_batch(1, n, seed0=int(rng.integers(1, 1 << 30)))[:2]; x = x.reshape(-1, 3)
        elif kind == 1: x, r = tools.globule(n, int(rng.integers(1, 1 << 30))); x = x.reshape(-1, 3)
        elif kind == 2: x, r = tools.globule(n, int(rng.integers(1, 1 << 30)), spacing=float(rng.uniform(1.2, 4.0))); x = x.reshape(-1, 3)
        else:
            x = rng.uniform(0, max(4.0, (n * 18.0) ** (1 / 3)), size=(n, 3)); r = rng.uniform(1.0, 2.2, size=n)
        if rng.random() < 0.3: r = r * rng.uniform(0.3, 3.0, size=len(r))
        if rng.random() < 0.3: x = x + rng.uniform(-1, 1, size=3) * 10 ** rng.uniform(0, 7)
        parts.append((np.ascontiguousarray(x, dtype=np.float64), np.ascontiguousarray(r, dtype=np.float64)))
    xyz = np.concatenate([p[0] for p in parts]); r = np.concatenate([p[1] for p in parts])
    offs = np.concatenate([[0], np.cumsum([len(p[1]) for p in parts])]).astype(np.int64)
    npts = int(rng.choice([1, 5, 20, 50, 64, 100, 100, 100, 127, 128]))
    probe = float(rng.choice([0.0, 1.0, 1.4, 1.4, 2.5]))
    table = rng.choice([None, None, "16,32", "8,8", "3,5", "24,64"])
    s2, c2 = run(xyz, r, offs, npts, probe, "0")
    s3, c3 = run(xyz, r, offs, npts, probe, table)
    ok = np.array_equal(c2, c3) and np.array_equal(s2, s3, equal_nan=True)
    atoms += len(r)
    if not ok:
        bad = np.nonzero(c2 != c3)[0]
        print(f"MISMATCH round {k}: kind {kind} structs {ns} atoms {len(r)} points {npts} probe {probe} table {table}: {len(bad)} atoms differ, first {bad[:5]}", flush=True)
        sys.exit(1)
print(f"{rounds} rounds, {atoms} atoms: third arrangement == second arrangement (counts and areas), {time.time() - t0:.0f} s")
