import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, tools, freesasa_amd as fa
scratch = "/tmp/drv_tuning"; os.makedirs(scratch, exist_ok=True)
n_atoms, n_frames = 100_000, 600
base, r = tools.globule(n_atoms, 5)
f64 = os.path.join(scratch, "frames.f64")
with open(f64, "wb") as a:
    for f in range(n_frames): tools.jitter(base, 100 + f, 0.5).tofile(a)
os.environ["FREESASA_AMD_TRAJ_PROFILE"] = "1"
for lanes in (3, 6):
    os.environ["FREESASA_AMD_TRAJ_LANES"] = str(lanes)
    fa.trajectory_file(f64, r, os.path.join(scratch, "t.bin"), os.path.join(scratch, "s.bin"), n_frames=26 * lanes)
    t0 = time.perf_counter(); fa.trajectory_file(f64, r, os.path.join(scratch, "t.bin"), os.path.join(scratch, "s.bin")); dt = time.perf_counter() - t0
    print(f"lanes {lanes}: {n_atoms * n_frames / dt:.3e} ({dt:.3f} s)  [the result file of the run before truncated inside the call]", flush=True)
    for k in range(2):
        for q in ("s.bin", "t.bin"):
            if os.path.exists(os.path.join(scratch, q)): os.unlink(os.path.join(scratch, q))
        t0 = time.perf_counter(); fa.trajectory_file(f64, r, os.path.join(scratch, "t.bin"), os.path.join(scratch, "s.bin")); dt = time.perf_counter() - t0
        print(f"lanes {lanes}: {n_atoms * n_frames / dt:.3e} ({dt:.3f} s)  [fresh result files]", flush=True)
    t0 = time.perf_counter(); fa.trajectory_file(f64, r, os.path.join(scratch, "t.bin"), None); dt = time.perf_counter() - t0
    print(f"lanes {lanes} totals only: {n_atoms * n_frames / dt:.3e} ({dt:.3f} s)", flush=True)
