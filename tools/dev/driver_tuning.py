"""DEV: lanes / batch sizes of the trajectory-file and cache-sweep drivers on one device."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import tools
import freesasa_amd as fa
from freesasa_amd import ingest
import bench
scratch = "/tmp/drv_tuning"; os.makedirs(scratch, exist_ok=True)
print("usable cpus", ingest.usable_cpus())
# trajectory
n_atoms, n_frames = 100_000, 600
base, r = tools.globule(n_atoms, 5)
f64 = os.path.join(scratch, "frames.f64")
with open(f64, "wb") as a:
    for f in range(n_frames): tools.jitter(base, 100 + f, 0.5).tofile(a)
for lanes in (3, 4, 5, 6, 8, 12):
    os.environ["FREESASA_AMD_TRAJ_LANES"] = str(lanes)
    fa.lib().freesasa_gpu_release_pool()
    fa.trajectory_file(f64, r, os.path.join(scratch, "t.bin"), os.path.join(scratch, "s.bin"), n_frames=26 * lanes)  # (every lane warm)
    best = 1e9
    for _ in range(2):
        for q in ("t.bin", "s.bin"):
            if os.path.exists(os.path.join(scratch, q)): os.unlink(os.path.join(scratch, q))   # (fresh result files: truncating the old ones inside the call costs ~50 ms)
        t0 = time.perf_counter(); fa.trajectory_file(f64, r, os.path.join(scratch, "t.bin"), os.path.join(scratch, "s.bin")); best = min(best, time.perf_counter() - t0)
    print(f"trajectory_file lanes {lanes}: {n_atoms * n_frames / best:.3e} atom-frames/s ({best:.3f} s)", flush=True)
    t0 = time.perf_counter(); fa.trajectory_file(f64, r, os.path.join(scratch, "t.bin"), os.path.join(scratch, "s32.bin"), out_f32=True); dt = time.perf_counter() - t0
    print(f"   fp32 per-atom output: {n_atoms * n_frames / dt:.3e}", flush=True)
    t0 = time.perf_counter(); fa.trajectory_file(f64, r, os.path.join(scratch, "t.bin"), None); dt = time.perf_counter() - t0
    print(f"   totals only: {n_atoms * n_frames / dt:.3e}", flush=True)
os.environ.pop("FREESASA_AMD_TRAJ_LANES")
if len(sys.argv) > 1 and sys.argv[1] == "traj": sys.exit(0)
# cache sweep
px, pr, poffs, per, reps = bench.real_pdb_batch(12_000_000)
pdb_dir = os.path.join(ROOT, "tests", "golden", "pdb")
one = ingest.load_pdb_files([os.path.join(pdb_dir, nm + ".pdb") for nm in bench.PDB_NAMES] * 40)
print("building cache ...", flush=True)
texts = [open(os.path.join(pdb_dir, nm + ".pdb"), "rb").read() for nm in bench.PDB_NAMES]
b = ingest.load_pdb_texts(texts * (12_000_000 // one.n_atoms * 40 + 1))
cache = os.path.join(scratch, "big.fsab"); b.save(cache)
n = int(b.n_atoms); print("cache atoms", n, "bytes", os.path.getsize(cache))
for nt in (1, 2, 4, 8, 16):
    t0 = time.perf_counter(); c = ingest.load_cache(cache, n_threads=nt); dt = time.perf_counter() - t0; del c
    print(f"full load {nt} threads: {n / dt:.3e} atoms/s", flush=True)
for lanes in (2, 3, 4, 6, 8):
    for ba in (500_000, 1_000_000, 2_000_000):
        fa.lib().freesasa_gpu_release_pool()
        fa.sweep_cache(cache, batch_atoms=ba, lanes_per_device=lanes, device=0)
        t0 = time.perf_counter(); fa.sweep_cache(cache, batch_atoms=ba, lanes_per_device=lanes, device=0); dt = time.perf_counter() - t0
        print(f"sweep_cache lanes {lanes} batch {ba}: {n / dt:.3e} atoms/s", flush=True)
