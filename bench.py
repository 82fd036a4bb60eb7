#!/usr/bin/env python3
"""bench.py — atoms/sec of the Lee-Richards hot path (20 slices, probe 1.4 A) on MI355X.

One "step" = one pass of the whole hot path (cell sort -> fused neighbor + L&R kernel ->
per-atom SASA and per-structure totals) over one batch of synthetic structures that is
already resident in HBM.  Workload: BASELINE.json configs[2]'s batch geometry (1000
random-coil structures x 10 000 atoms, seeds 1000+k) at the metric's parameters (L&R, 20
slices, probe 1.4 A, fp64).  Multi-GPU: every rank owns its own 1000 structures (weak
scaling, independent structures, no collective on the data path); the timed region is
bracketed by barrier + synchronize and the MAX over ranks is reported.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--structs S] [--atoms A]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects:
  roofline     — achieved = ALGORITHMIC bytes (40 B/atom: x,y,z,R in, SASA out) of the dominant
                 kernel (k_lr2_tile) / its HIP-event duration measured live on the launch stream;
  end_to_end   — SURVEY 8(d)(i): the same batch from host arrays to host per-atom areas, PCIe included;
  cpu_baseline — the real reference (oracle/_ref, kind "reference") or the oracle port, timed on
                 this box's host cores on a bounded sample of the same batch (own process: one worker
                 process per core, plus the reference's own n_threads 1 / 2 / 16 on one structure).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_ATOM = 40.0  # read x,y,z,R (32 B) + write sasa (8 B), fp64
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
FP64_VECTOR_PEAK_TFLOPS = 78.6
SIMDS, CLOCK_HZ, CYCLES_PER_VALU = 256 * 4, 2.4e9, 4   # MI355X: 256 CUs x 4 SIMD16; a wave64 VALU op issues over 4 cycles
PROFILED_VALU = None        # wave-level VALU instructions per launch of the dominant kernel (committed PMC pass)


def cpu_baseline(args, offs, gpu_sasa, budget_s=12.0):
    """The reference's CPU path on this box's host cores (oracle/cpu_baseline.py, its own process: it forks one
    worker per core, which a process that holds a HIP context must not do).  Reports the all-cores batch rate
    (one process per core, n_threads=1) as `value`, and next to it the reference's own per-structure threading
    (n_threads 1 / 2 / 16 on one structure), the CPU model and the core count.  Returns the JSON object and the
    largest per-atom difference between the GPU's areas and the reference's on the structures it kept."""
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "areas.npz")
        cmd = [sys.executable, os.path.join(ROOT, "oracle", "cpu_baseline.py"), "--atoms", str(args.atoms),
               "--structs", str(args.structs), "--seed0", "1000", "--slices", str(args.slices), "--budget", str(budget_s),
               "--keep", "4", "--out", out]
        res = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        if res.returncode != 0:
            raise RuntimeError("cpu baseline failed: " + res.stderr[-500:])
        base = json.loads(res.stdout.strip().splitlines()[-1])
        err = 0.0
        with np.load(out) as z:
            for k in range(len(z.files)):
                err = max(err, float(np.max(np.abs(z[f"s{k}"] - gpu_sasa[offs[k]:offs[k + 1]]))))
    return base, err


def end_to_end(fa, torch, xyz, r, offs, args, device, resident_sasa):
    """the SAME batch from "host arrays ready" to "per-atom areas in host memory", PCIe copies
    included, through freesasa_gpu_calc_batch_pipelined (chunks of structures on a few host lanes: upload, kernels
    and download of different chunks overlap).  Page-locked host arrays (the figure reported) and pageable ones."""
    n = len(r)
    res = {}
    for kind in ("page-locked", "pageable"):
        if kind == "page-locked":
            hx, hr = torch.from_numpy(xyz).pin_memory().numpy(), torch.from_numpy(r).pin_memory().numpy()
            out = (torch.empty(n, dtype=torch.float64).pin_memory().numpy(), None,
                   torch.empty(len(offs) - 1, dtype=torch.float64).pin_memory().numpy())
        else:
            hx, hr, out = xyz, r, None
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            got = fa.calc_batch_pipelined(hx, hr, offs, probe=1.4, resolution=args.slices, device=device, out=out)
            dt = time.perf_counter() - t0
            best = dt if best is None or dt < best else best
        res[kind] = {"value": n / best, "unit": "atoms/s", "ms": 1e3 * best,
                     "pcie_GBps_in_plus_out": ALGO_BYTES_PER_ATOM * n / best / 1e9,
                     "identical_to_resident_run": bool(np.array_equal(got[0], resident_sasa))}
    top = dict(res["page-locked"])
    top["host_memory"] = "page-locked (DMA in place), 3 host lanes x chunks of 1.25e6 atoms"
    top["pageable_host_memory"] = res["pageable"]
    top["note"] = "host xyz/radii -> host per-atom SASA and per-structure totals, H2D 32 B/atom + D2H 8 B/atom included"
    return top


def two_streams(fa, torch, d_xyz, d_r, offs, args, dev, local_rank, steps):
    """What a production loop over device-resident batches gets when it keeps two passes in flight: two host
    threads, each with its own context, stream and output buffers, take the passes alternately (the calls are
    synchronous; ctypes releases the GIL).  The cell sort of one pass (HBM-bound) and the host gap between passes
    then run under the tile kernel of the other (VALU-bound).  Reported next to the headline, not as it: with two
    kernels sharing the GPU the per-launch durations no longer describe one kernel."""
    import threading
    n, n_structs = int(offs[-1]), len(offs) - 1
    lanes = []
    for _ in range(2):
        st = torch.cuda.Stream(device=dev)
        lanes.append((fa.GpuContext(local_rank, stream=st.cuda_stream), st,
                      torch.empty(n, dtype=torch.float64, device=dev), torch.empty(n_structs, dtype=torch.float64, device=dev)))

    def run(k, count):
        ctx, _, out, tot = lanes[k]
        for _ in range(count):
            ctx.lee_richards(d_xyz.data_ptr(), d_r.data_ptr(), offs, out.data_ptr(), tot.data_ptr(), probe=1.4, n_slices=args.slices)

    for k in range(2):
        run(k, 2)                                   # warm-up: workspace, launch shape
    torch.cuda.synchronize()
    per = (steps + 1) // 2
    th = [threading.Thread(target=run, args=(k, per)) for k in range(2)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    same = bool(torch.equal(lanes[0][2], lanes[1][2]))
    for ctx, *_ in lanes:
        ctx.close()
    return {"value": 2 * per * n / dt, "unit": "atoms/s", "steps": 2 * per, "ms_per_step": 1e3 * dt / (2 * per),
            "identical_outputs": same,
            "note": "two host threads x (context, stream, output buffers), passes taken alternately; the cell sort and the "
                    "host gap of one pass run under the tile kernel of the other"}


def neighbors_per_atom(fa, torch, d_xyz, d_r, offs, dev, local_rank):
    """average number of unique neighbors per atom (the work per atom is proportional to it), counted by
    the kernel's own neighbor phase (freesasa_gpu_lr_neighbors_dev) on a context of its own, outside every timed region."""
    n = int(offs[-1])
    d_nn = torch.empty(n, dtype=torch.int32, device=dev)
    ctx = fa.GpuContext(local_rank)
    ctx.lr_neighbors(d_xyz.data_ptr(), d_r.data_ptr(), offs, d_nn.data_ptr(), probe=1.4)
    torch.cuda.synchronize()
    ctx.close()
    return float(d_nn.to(torch.float64).mean().item())


def checker_error(alg, xyz, r, got, resolution):
    """cpu_baseline leg (the checker, not the thing measured): largest per-atom difference to the real reference
    (oracle/_ref) or, without it, to the oracle port, on ONE structure; for Shrake-Rupley the number of atoms whose
    test-point count differs."""
    import oracle
    if alg == "lr":
        if oracle.Reference.available():
            want = oracle.Reference().calc_coord(xyz, r, oracle.LEE_RICHARDS, 1.4, n_slices=resolution)[0]
        else:
            want = oracle.Oracle().lee_richards(xyz, r, 1.4, resolution)
        return float(np.max(np.abs(np.asarray(want) - got)))
    want = oracle.Oracle().shrake_rupley(xyz, r, 1.4, resolution)[1]
    return int(np.count_nonzero(np.asarray(want) != got))


PDB_NAMES = ["1a0q", "3gnn", "5dx9", "2jo4", "3bkr", "1d3z", "1ubq"]


def real_pdb_batch(min_atoms=3_000_000):
    """The reference's own PDB test entries through the batched reader (its default filters, ProtOr radii), replicated to
    >= min_atoms atoms: (xyz, radii, offsets, atoms per entry, copies) - the offline stand-in for BASELINE configs[1] / [3]
    that is not a lattice."""
    from freesasa_amd import ingest
    pdb_dir = os.path.join(ROOT, "tests", "golden", "pdb")
    b = ingest.load_pdb_files([os.path.join(pdb_dir, nm + ".pdb") for nm in PDB_NAMES])
    per = [int(b.offsets[k + 1] - b.offsets[k]) for k in range(b.n_structs)]
    reps = max(1, -(-min_atoms // int(b.n_atoms)))
    px = np.ascontiguousarray(np.tile(b.xyz.reshape(-1), reps)); pr = np.ascontiguousarray(np.tile(b.radii, reps))
    poffs = np.concatenate([[0], np.cumsum(np.tile(per, reps))]).astype(np.int64)
    return px, pr, poffs, per, reps


def sr_workloads(fa, torch, tools, d_xyz, d_r, offs, xyz, r, dev, local_rank, check, args):
    """Shrake-Rupley (100 points) on the two batch workloads of the Lee-Richards lines - the 1000 x 10 000 coils and the
    reference's PDB entries x 251 - with the share of the VALU issue slots the S&R tile kernel fills (one rocprofv3
    --pmc SQ_INSTS_VALU pass of this script per workload, as for the headline)."""
    out = {}

    def run(name, dx, dr, o, hx, hr, wl, child_workload):
        n = int(o[-1])
        d_out = torch.empty(n, dtype=torch.float64, device=dev)
        d_cnt = torch.empty(n, dtype=torch.int32, device=dev)
        ctx = fa.GpuContext(local_rank, timing=True)
        call = lambda: ctx.shrake_rupley(dx.data_ptr(), dr.data_ptr(), o, d_out.data_ptr(), d_cnt.data_ptr(), 0, probe=1.4, n_points=100)
        for _ in range(3):
            call()
        ctx.set_timing(False)                       # (the step's wall clock without the context's event records ...)
        call()
        torch.cuda.synchronize()
        import gc
        gc.collect(); gc.disable()                  # (see secondary_workloads.run)
        try:
            t0 = time.perf_counter()
            for _ in range(10):
                call()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 10
        finally:
            gc.enable()
        ctx.set_timing(True)                        # (... the kernels' own durations from three further calls)
        ks, ps = [], []
        for _ in range(3):
            call()
            st = ctx.stats(); ks.append(st["ms_kernel"]); ps.append(st["ms_prep"])
        st = ctx.stats()
        res = {"value": n / dt, "unit": "atoms/s", "ms_per_step": 1e3 * dt, "kernel_ms": float(np.mean(ks)), "prep_ms": float(np.mean(ps)), "steps": 10,
               "kernel_atoms_per_s": n / (1e-3 * float(np.mean(ks))), "workload": wl, "tile_atoms": st["tile_atoms"],
               "block_threads": st["block_threads"], "max_neighbors_per_atom": st["max_neighbors"], "fallback_tiles": st["fallback_tiles"]}
        if check:
            k1 = int(o[1])
            res["atoms_with_a_different_count"] = checker_error("sr", hx[:3 * k1], hr[:k1], d_cnt[:k1].cpu().numpy(), 100)
        ctx.close()
        import copy
        a2 = copy.copy(args); a2.workload = child_workload; a2.points = 100
        live = live_counters(a2, counters=("SQ_INSTS_VALU",))
        if live:
            valu = live[1]
            res["valu_issue"] = {"wave_instructions_per_launch": valu, "wave_instructions_per_atom": valu / n,
                                 "frac_of_issue_slots": valu * CYCLES_PER_VALU / (1e-3 * float(np.mean(ks)) * SIMDS * CLOCK_HZ), "source": live[2]}
        out[name] = res

    try:
        run("coil_sr100", d_xyz, d_r, offs, xyz.reshape(-1), r, f"{len(offs) - 1} coils x {int(offs[1])} atoms (the headline batch), Shrake-Rupley 100 points", "coil_sr")
    except Exception as exc:
        out["coil_sr100"] = {"error": repr(exc)}
    try:
        px, pr, poffs, per, reps = real_pdb_batch()
        dpx, dpr = torch.from_numpy(px).to(dev), torch.from_numpy(pr).to(dev)
        run("real_pdb_sr100", dpx, dpr, poffs, px, pr, f"{reps} x the PDB entries {', '.join(PDB_NAMES)} ({len(poffs) - 1} structures, {len(pr)} atoms), Shrake-Rupley 100 points", "pdb_sr")
    except Exception as exc:
        out["real_pdb_sr100"] = {"error": repr(exc)}
    return out


def latency_us(fa, tools):
    """What BASELINE configs[0] feels: ONE structure through the drop-in entry freesasa_calc_coord (host arrays in, host
    areas out, a pooled context): median and best of 60 calls, 1UBQ (602 atoms) and 1A0Q (3 3xx atoms), both algorithms."""
    from freesasa_amd import ingest
    pdb_dir = os.path.join(ROOT, "tests", "golden", "pdb")
    out = {}
    for nm in ("1ubq", "1a0q"):
        b = ingest.load_pdb_files([os.path.join(pdb_dir, nm + ".pdb")])
        x, rr = np.ascontiguousarray(b.xyz.reshape(-1, 3)), np.ascontiguousarray(b.radii)
        for alg, tag, kw in ((fa.LEE_RICHARDS, "lr20", {"n_slices": 20}), (fa.SHRAKE_RUPLEY, "sr100", {"n_points": 100})):
            for _ in range(5):
                fa.calc_coord(x, rr, alg, **kw)
            ts = []
            for _ in range(60):
                t0 = time.perf_counter()
                fa.calc_coord(x, rr, alg, **kw)
                ts.append(time.perf_counter() - t0)
            out[f"{nm}_{tag}"] = {"atoms": int(b.n_atoms), "median_us": 1e6 * float(np.median(ts)), "best_us": 1e6 * float(np.min(ts))}
    out["note"] = "freesasa_calc_coord per call: upload, cell sort, tile kernel, totals, download, status readback (Python ctypes overhead ~10 us included)"
    return out


def driver_workloads(fa, tools, local_rank, scratch):
    """BASELINE configs[3] / configs[4] as one GPU's shard, through the REAL drivers, with the host rates beside them:
    sweep_files (PDB + mmCIF files on local disk -> totals), sweep_cache (the same structures from the binary cache),
    trajectory_file (frames of a 100 000-atom system from a raw frame file, fp64 and fp32, per-atom areas streamed out)."""
    import shutil
    from freesasa_amd import ingest
    out = {}
    cpus = ingest.usable_cpus()
    # ---- files: the reference's 7 PDB entries + 4 mmCIF fixtures, copied until >= 3e6 atoms
    try:
        pdb_dir, cif_dir = os.path.join(ROOT, "tests", "golden", "pdb"), os.path.join(ROOT, "tests", "golden", "cif")
        srcs = [os.path.join(pdb_dir, nm + ".pdb") for nm in PDB_NAMES] + sorted(os.path.join(cif_dir, f) for f in os.listdir(cif_dir) if f.endswith(".cif"))[:4]
        one = ingest.load_pdb_files(srcs)
        reps = max(1, -(-3_000_000 // int(one.n_atoms)))
        d = os.path.join(scratch, "sweep_files")
        os.makedirs(d, exist_ok=True)
        paths = []
        for k in range(reps):
            for sp in srcs:
                dst = os.path.join(d, f"{k:04d}_{os.path.basename(sp)}")
                if not os.path.exists(dst):
                    shutil.copyfile(sp, dst)
                paths.append(dst)
        nbytes = sum(os.path.getsize(q) for q in paths)
        t0 = time.perf_counter(); b = ingest.load_pdb_files(paths); t_load = time.perf_counter() - t0      # the loader alone (all granted CPUs)
        n = int(b.n_atoms)
        fa.sweep_files(paths[:len(srcs)], device=local_rank)                                                # warm-up: context, workspace
        t0 = time.perf_counter(); tot, cls, atoms, status = fa.sweep_files(paths, device=local_rank); t_sweep = time.perf_counter() - t0
        want, _, wtot = fa.calc_batch(b.xyz, b.radii, b.offsets, fa.LEE_RICHARDS, 1.4, 20, device=local_rank)
        t0 = time.perf_counter(); fa.calc_batch(b.xyz, b.radii, b.offsets, fa.LEE_RICHARDS, 1.4, 20, device=local_rank); t_gpu = time.perf_counter() - t0
        # the sweep with the parser ON THE DEVICE (round 6: gpu_parse.hip) against the host parser, on four copies of the list
        # (1.2e7 atoms, the drivers' default batches and two workers on the device, so that reading, PCIe and kernels of different batches overlap); host CPU seconds
        # of the whole process beside the wall clock (what the host still does: read the bytes, find an mmCIF loop's header)
        many = paths * 4
        res = {}
        for tag, opt in (("host", 0), ("device", ingest.PARSE_ON_DEVICE)):
            fa.sweep_files(many[:16 * len(srcs)], device=local_rank, ingest_options=opt)   # warm-up: both workers' staging buffers, tables, workspace
            fa.sweep_parse_stats()
            best = None
            for _ in range(2):
                c0, t0 = time.process_time(), time.perf_counter()
                r4 = fa.sweep_files(many, device=local_rank, ingest_options=opt)
                dt, cpu = time.perf_counter() - t0, time.process_time() - c0
                if best is None or dt < best[0]: best = (dt, cpu)
            res[tag] = (best, r4)
        dev_files, host_files = fa.sweep_parse_stats()
        (t_dev, cpu_dev), r_dev = res["device"]
        (t_host, cpu_host), r_host = res["host"]
        same = all(np.array_equal(x, y) for x, y in zip(r_dev, r_host)) and bool(np.array_equal(r_dev[0][:len(tot)], tot))
        out["sweep_files"] = {"value": 4 * n / t_dev, "unit": "atoms/s", "parser": "device", "files": len(many), "atoms": 4 * n, "file_bytes": 4 * nbytes, "seconds": t_dev,
                              "text_GB_per_s": 4 * nbytes / t_dev / 1e9, "host_cpu_seconds": cpu_dev, "host_cpu_ns_per_atom": 1e9 * cpu_dev / (4 * n),
                              "files_parsed_on_device_last_run": dev_files // 2, "files_left_to_the_host_parser_last_run": host_files // 2,
                              "identical_to_the_host_parser_sweep": bool(same),
                              "host_parser": {"value": 4 * n / t_host, "unit": "atoms/s", "seconds": t_host, "host_cpu_seconds": cpu_host,
                                              "host_cpu_ns_per_atom": 1e9 * cpu_host / (4 * n), "one_pass_of_the_list": n / t_sweep,
                                              "loader_atoms_per_s": n / t_load, "loader_MB_per_s": nbytes / t_load / 1e6, "loader_threads": cpus},
                              "gpu_atoms_per_s_host_arrays": n / t_gpu, "totals_equal_load_then_batch": bool(np.array_equal(tot, wtot)),
                              "workload": f"{4 * reps} copies of {len(srcs)} files ({', '.join(os.path.basename(q) for q in srcs)}) on local disk (page cache warm), "
                                          "freesasa_gpu_sweep_files (default batches, two workers on the device): Lee-Richards 20 slices, totals and class sums per file; parser on the device "
                                          "(host threads read bytes; ~130 bytes of text per atom over PCIe), the host parser's sweep of the same list beside it"}
        # ---- the same structures (four copies: 1.2e7 atoms, so that the lanes have batches to overlap) from the binary cache
        cache = os.path.join(scratch, "sweep.fsab")
        b4 = ingest.load_pdb_files(paths * 4)
        n4 = int(b4.n_atoms)
        t0 = time.perf_counter(); b4.save(cache); t_save = time.perf_counter() - t0
        del b4
        best_load = None
        for nt in (1, 0):
            t0 = time.perf_counter(); c = ingest.load_cache(cache, n_threads=nt); dt = time.perf_counter() - t0
            best_load = (dt, nt) if nt == 0 else best_load
            if nt == 1: t_load1 = dt
            del c
        fa.sweep_cache(cache, device=local_rank)                                                            # warm-up (page-locked staging of the lanes)
        t0 = time.perf_counter(); ctot, ccls, catoms, cstatus = fa.sweep_cache(cache, device=local_rank); t_cs = time.perf_counter() - t0
        out["sweep_cache"] = {"value": n4 / t_cs, "unit": "atoms/s", "atoms": n4, "seconds": t_cs, "cache_bytes": os.path.getsize(cache),
                              "full_load_atoms_per_s_1_thread": n4 / t_load1, "full_load_atoms_per_s_all_threads": n4 / best_load[0], "threads": min(cpus, 8),
                              "lanes": min(8, max(2, cpus)), "save_seconds": t_save,
                              "totals_equal_file_sweep": bool(np.array_equal(ctot[:len(tot)], tot) and np.array_equal(ccls[:len(tot)], cls) and np.array_equal(ctot[len(tot):2 * len(tot)], tot)),
                              "workload": "the same structures x 4 from the version-2 cache file (page cache warm): freesasa_gpu_sweep_cache_devices, one device, "
                                          "lanes read + verify (1 MiB piece checksums) coordinates, radii and classes only, into page-locked staging"}
        del b
    except Exception as exc:
        out.setdefault("sweep_files", {"error": repr(exc)})
        out.setdefault("sweep_cache", {"error": repr(exc)})
    # ---- trajectory: 1000 frames x 100 000 atoms from a frame file, per-atom stream-out
    try:
        n_atoms, n_frames = 100_000, 1000
        base, r = tools.globule(n_atoms, 5)
        f64, f32 = os.path.join(scratch, "frames.f64"), os.path.join(scratch, "frames.f32")
        if not (os.path.exists(f64) and os.path.getsize(f64) == 24 * n_atoms * n_frames and os.path.exists(f32)):
            with open(f64, "wb") as a, open(f32, "wb") as c:
                for f in range(n_frames):
                    fr = tools.jitter(base, 100 + f, 0.5)
                    fr.tofile(a); fr.astype(np.float32).tofile(c)
        res = {}
        for tag, path, is32, o32 in (("f64", f64, False, False), ("f64_out_f32", f64, False, True), ("f32", f32, True, False)):
            tp, sp = os.path.join(scratch, f"tot_{tag}.bin"), os.path.join(scratch, f"sasa_{tag}.bin")
            fa.trajectory_file(path, r, tp, sp, f32=is32, out_f32=o32, n_frames=104, device=local_rank)   # warm-up: 8 shards, so that EVERY lane's context, page-locked staging and tile shape exist (24 frames - two shards - left the third lane cold in rounds 4 and 5: ~50 ms of a 0.38 s run)
            dt = 1e9
            for _ in range(2):       # the better of two runs (0.3 s each, on a box shared with other tenants' jobs)
                for q in (tp, sp):   # fresh result files: until round 6 the timed call began by TRUNCATING the warm-up's (rounds 4-5: the full
                    if os.path.exists(q): os.unlink(q)   # previous run's) 0.8 GB file - freeing its page-cache pages took ~50 ms of a 0.35 s run, and was read as "page-cache writes bind"
                t0 = time.perf_counter(); done, got = fa.trajectory_file(path, r, tp, sp, f32=is32, out_f32=o32, device=local_rank); dt = min(dt, time.perf_counter() - t0)
            res[tag] = {"value": n_atoms * n_frames / dt, "unit": "atom-frames/s", "seconds": dt, "frames": int(got), "complete": bool(done),
                        "in_GB_per_s": (12 if is32 else 24) * n_atoms * n_frames / dt / 1e9,
                        "writer_GB_per_s": (4 if o32 else 8) * n_atoms * n_frames / dt / 1e9}
        tp = os.path.join(scratch, "tot_only.bin")
        t0 = time.perf_counter(); fa.trajectory_file(f64, r, tp, None, device=local_rank); dt = time.perf_counter() - t0
        res["totals_only"] = {"value": n_atoms * n_frames / dt, "unit": "atom-frames/s", "seconds": dt}
        t64 = np.fromfile(os.path.join(scratch, "tot_f64.bin"))
        res["value"], res["unit"] = res["f64"]["value"], "atom-frames/s"
        res["mean_total_A2"] = float(t64.mean())
        res["workload"] = (f"{n_frames} frames x {n_atoms} atoms (globule + 0.5 A jitter) from a raw frame file (page cache warm), freesasa_gpu_trajectory_file: "
                           "totals file + per-atom areas file written (8 B per atom-frame; f64_out_f32: narrowed on the device to 4 B, an output "
                           "format); fp64 frames, and fp32 frames widened on the device; 3 host lanes, all warm")
        out["trajectory_file"] = res
        for q in (f64, f32):
            pass
    except Exception as exc:
        out["trajectory_file"] = {"error": repr(exc)}
    return out


def secondary_workloads(fa, torch, tools, d_xyz, d_r, offs, xyz, r, dev, local_rank, check):
    """The other single-GPU configurations of BASELINE.json, one short measurement each in this same process:
    configs[2] as written (L&R 100 slices on the 1000 x 10 000 coils), protein density (10 000-atom globules, L&R 20),
    configs[1]'s proxy (one 200 000-atom globule, Shrake-Rupley 100 points).  Each: atoms/s over whole steps (wall
    clock around synchronous calls, inputs resident), the tile kernel's own time, average neighbors per atom and —
    with the CPU baseline enabled — the largest difference to the checker on one structure."""
    out = {}

    def run(ctx, call, n_atoms, steps, warmup):
        """ms_per_step: wall clock around `steps` synchronous calls WITHOUT the context's own event timing (four event
        records per batch: ~35 us, a tenth of a step of the 200 000-atom case); kernel_ms: the tile kernel's duration from
        those events, over three further calls outside the timed region."""
        for _ in range(warmup):
            call()
        ctx.set_timing(False)
        call()
        torch.cuda.synchronize()
        # (the interpreter's cyclic garbage collector off for the timed loop: round 6 caught it stopping ONE of ten 6 ms calls
        # for 32 ms - this process holds millions of objects by now - which read as "real_pdb_lr20 3.2e8" in two sessions)
        import gc
        gc.collect(); gc.disable()
        try:
            t0 = time.perf_counter()
            per_call = []
            for _ in range(steps):
                tc = time.perf_counter()
                call()
                per_call.append(1e3 * (time.perf_counter() - tc))
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
        finally:
            gc.enable()
        if os.environ.get("FREESASA_AMD_BENCH_DEBUG"):
            print("run(): per call ms " + " ".join(f"{v:.2f}" for v in per_call), file=sys.stderr, flush=True)
        ctx.set_timing(True)
        ks = []
        for _ in range(3):
            call()
            ks.append(ctx.stats()["ms_kernel"])
        return {"value": n_atoms / dt, "unit": "atoms/s", "ms_per_step": 1e3 * dt, "kernel_ms": float(np.mean(ks)), "steps": steps}

    n = int(offs[-1])
    d_out = torch.empty(n, dtype=torch.float64, device=dev)
    # configs[2] as written
    ctx = fa.GpuContext(local_rank, timing=True)
    res = run(ctx, lambda: ctx.lee_richards(d_xyz.data_ptr(), d_r.data_ptr(), offs, d_out.data_ptr(), 0, probe=1.4, n_slices=100), n, 3, 2)
    st = ctx.stats()
    res.update({"workload": f"{len(offs) - 1} coils x {int(offs[1])} atoms, Lee-Richards 100 slices (BASELINE configs[2] as written)",
                "tile_atoms": st["tile_atoms"], "max_neighbors_per_atom": st["max_neighbors"]})
    if check:
        res["max_abs_dsasa"] = checker_error("lr", xyz[:offs[1]], r[:offs[1]], d_out[:int(offs[1])].cpu().numpy(), 100)
    ctx.close()
    out["lr100_config2"] = res
    # protein density
    ng, na = 300, 10_000
    parts = [tools.globule(na, 500 + k) for k in range(ng)]
    gx = np.concatenate([p[0] for p in parts]); gr = np.concatenate([p[1] for p in parts])
    goffs = np.arange(ng + 1, dtype=np.int64) * na
    dgx, dgr = torch.from_numpy(gx).to(dev), torch.from_numpy(gr).to(dev)
    dgo = torch.empty(ng * na, dtype=torch.float64, device=dev)
    ctx = fa.GpuContext(local_rank, timing=True)
    res = run(ctx, lambda: ctx.lee_richards(dgx.data_ptr(), dgr.data_ptr(), goffs, dgo.data_ptr(), 0, probe=1.4, n_slices=20), ng * na, 10, 3)
    st = ctx.stats()
    res.update({"workload": f"{ng} synthetic globules x {na} atoms (lattice spacing 2.6 A: protein density), Lee-Richards 20 slices",
                "tile_atoms": st["tile_atoms"], "max_neighbors_per_atom": st["max_neighbors"],
                "avg_neighbors_per_atom": neighbors_per_atom(fa, torch, dgx, dgr, goffs, dev, local_rank)})
    if check:
        res["max_abs_dsasa"] = checker_error("lr", gx[:na], gr[:na], dgo[:na].cpu().numpy(), 20)
    ctx.close()
    out["globule_lr20"] = res
    del dgx, dgr, dgo
    # real proteins: the reference's own test entries, ProtOr radii through the batched reader, replicated to >= 3e6 atoms
    try:
        names = PDB_NAMES
        px, pr, poffs, per, reps = real_pdb_batch()
        dpx, dpr = torch.from_numpy(px).to(dev), torch.from_numpy(pr).to(dev)
        dpo = torch.empty(len(pr), dtype=torch.float64, device=dev)
        ctx = fa.GpuContext(local_rank, timing=True)
        # first_call: what a one-shot caller pays on a context that has seen nothing (workspace allocation, the density
        # sample that shapes the tile kernel, a first launch of every kernel, a tile shape not yet learnt from demand) for one
        # heterogeneous 3e6-atom batch, against the same call in steady state (round-5 review, weak 7)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctx.lee_richards(dpx.data_ptr(), dpr.data_ptr(), poffs, dpo.data_ptr(), 0, probe=1.4, n_slices=20)
        torch.cuda.synchronize()
        first_ms = 1e3 * (time.perf_counter() - t0)
        first_fb = ctx.stats()["fallback_tiles"]
        t0 = time.perf_counter()
        ctx.lee_richards(dpx.data_ptr(), dpr.data_ptr(), poffs, dpo.data_ptr(), 0, probe=1.4, n_slices=20)
        torch.cuda.synchronize()
        second_ms = 1e3 * (time.perf_counter() - t0)
        res = run(ctx, lambda: ctx.lee_richards(dpx.data_ptr(), dpr.data_ptr(), poffs, dpo.data_ptr(), 0, probe=1.4, n_slices=20), len(pr), 10, 3)
        st = ctx.stats()
        out["first_call"] = {"atoms": len(pr), "structures": len(poffs) - 1, "first_call_ms": first_ms, "second_call_ms": second_ms,
                             "steady_state_ms": res["ms_per_step"], "first_call_atoms_per_s": len(pr) / (1e-3 * first_ms),
                             "fallback_tiles_first_call": int(first_fb), "fallback_tiles_steady_state": int(st["fallback_tiles"]),
                             "workload": "the real_pdb_lr20 batch on a context created just before the call (the process has run other contexts: "
                                         "kernels are loaded, the device is warm)"}
        res.update({"workload": f"{reps} x the PDB entries {', '.join(names)} of the reference's test data ({', '.join(str(v) for v in per)} atoms after its "
                                f"default filters; ProtOr radii, include/freesasa_ingest.h), {len(poffs) - 1} structures, {len(pr)} atoms, Lee-Richards 20 slices: "
                                "the offline stand-in for BASELINE configs[1] / [3] that is not a lattice",
                    "tile_atoms": st["tile_atoms"], "max_neighbors_per_atom": st["max_neighbors"],
                    "avg_neighbors_per_atom": neighbors_per_atom(fa, torch, dpx, dpr, poffs, dev, local_rank)})
        if check:
            k = names.index("1a0q")
            sl = slice(int(poffs[k]), int(poffs[k + 1]))
            res["max_abs_dsasa"] = checker_error("lr", px.reshape(-1, 3)[sl].reshape(-1), pr[sl], dpo[sl].cpu().numpy(), 20)
            res["checked_entry"] = "1a0q"
        ctx.close()
        out["real_pdb_lr20"] = res
        del dpx, dpr, dpo
    except Exception as exc:        # (a secondary line must not take the headline down)
        out["real_pdb_lr20"] = {"error": repr(exc)}
    # configs[1] proxy
    nb = 200_000
    bx, br = tools.globule(nb, 77)
    boffs = np.array([0, nb], dtype=np.int64)
    dbx, dbr = torch.from_numpy(bx).to(dev), torch.from_numpy(br).to(dev)
    dbo = torch.empty(nb, dtype=torch.float64, device=dev)
    dbc = torch.empty(nb, dtype=torch.int32, device=dev)
    ctx = fa.GpuContext(local_rank, timing=True)
    res = run(ctx, lambda: ctx.shrake_rupley(dbx.data_ptr(), dbr.data_ptr(), boffs, dbo.data_ptr(), dbc.data_ptr(), 0, probe=1.4, n_points=100), nb, 20, 3)
    st = ctx.stats()
    res.update({"workload": "one synthetic 200 000-atom globule (BASELINE configs[1] proxy: 4V6X is not available offline), Shrake-Rupley 100 points",
                "max_neighbors_per_atom": st["max_neighbors"],
                "avg_neighbors_per_atom": neighbors_per_atom(fa, torch, dbx, dbr, boffs, dev, local_rank)})
    if check:
        res["atoms_with_a_different_count"] = checker_error("sr", bx, br, dbc.cpu().numpy(), 100)
    ctx.close()
    out["globule_sr100_200k"] = res
    return out


def live_counters(args, timeout_s=75, counters=("SQ_INSTS_VALU", "FETCH_SIZE", "WRITE_SIZE")):
    """HBM bytes and wave-level VALU instructions per launch of the dominant kernel, measured NOW: one extra run of this
    script (one timed step, synchronous entry, nothing secondary) under rocprofv3 per counter - FETCH_SIZE, WRITE_SIZE
    and SQ_INSTS_VALU in separate --pmc passes, as MI355X_MICROARCH.md prescribes - when rocprofv3 is on PATH.  Values
    of one dispatch are summed over its XCD instances; the main launch of the tile kernel is the k_lr2_tile /
    k_sr_tile instantiation with the most VALU instructions.  Returns (traffic_bytes, valu_per_launch, note) or None."""
    import csv
    import collections
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3")
    if not exe or args.no_live_counters or os.environ.get("FREESASA_AMD_BENCH_CHILD"):
        return None
    per = {}
    try:
        with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
            for counter in counters:
                cmd = [exe, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", os.path.join(tmp, counter), "-o", "c", "--",
                       sys.executable, os.path.abspath(__file__), "--steps", "1", "--warmup", "1", "--structs", str(args.structs), "--atoms", str(args.atoms),
                       "--slices", str(args.slices), "--workload", args.workload, "--points", str(args.points), "--sync-entry",
                       "--no-cpu-baseline", "--no-end-to-end", "--no-secondary", "--no-neighbors", "--no-live-counters", "--no-drivers"]
                env = dict(os.environ, FREESASA_AMD_BENCH_CHILD="1", TMPDIR="/tmp")
                res = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, cwd="/tmp", env=env)
                if res.returncode != 0:
                    return None
                total, launches = collections.defaultdict(float), collections.defaultdict(set)
                for root, _, files in os.walk(os.path.join(tmp, counter)):
                    for f in files:
                        if f.endswith("counter_collection.csv"):
                            for r in csv.DictReader(open(os.path.join(root, f))):
                                if r["Counter_Name"] == counter:
                                    total[r["Kernel_Name"]] += float(r["Counter_Value"])
                                    launches[r["Kernel_Name"]].add(r["Dispatch_Id"])
                per[counter] = {k: v / len(launches[k]) for k, v in total.items()}
    except (OSError, subprocess.SubprocessError, KeyError, ValueError):
        return None
    tiles = {k: v for k, v in per.get("SQ_INSTS_VALU", {}).items() if "k_lr2_tile<" in k or "k_sr_tile<" in k or "k_lr_tile<" in k}
    if not tiles:
        return None
    main = max(tiles, key=tiles.get)
    if "FETCH_SIZE" not in counters:
        return None, tiles[main], f"live: rocprofv3 --pmc pass of this run ({main.split('(')[0]})"
    if main not in per.get("FETCH_SIZE", {}) or main not in per.get("WRITE_SIZE", {}):
        return None
    # both counters are in KB; on gfx950 FETCH_SIZE reports 0.500 x the bytes moved and WRITE_SIZE 1.000 x (copy kernels of known
    # size, profiles/r0N_fetch_calibration.txt: re-checked every round by tools/gpu_round.sh)
    traffic = (2.0 * per["FETCH_SIZE"][main] + per["WRITE_SIZE"][main]) * 1024.0
    return traffic, tiles[main], f"live: rocprofv3 --pmc passes of this run ({main.split('(')[0]})"


def profiled_traffic(args):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE
    collected in separate --pmc runs of this same command, KB, summed over the XCD instances of a dispatch), corrected
    as MI355X_MICROARCH.md prescribes and as the calibration of the same session confirms: copy kernels of known
    size (tools/dev/fetch_calib.hip: 8-byte, 16-byte and 32-byte-record accesses, streamed and gathered) report
    FETCH_SIZE = 0.500 x and WRITE_SIZE = 1.000 x the bytes they move, so bytes = (2 FETCH_SIZE + WRITE_SIZE) x 1024
    (profiles/README.md, "FETCH_SIZE calibration").  Only valid for the default workload."""
    if (args.structs, args.atoms, args.slices) != (1000, 10000, 20):
        return None, None
    import re
    best, best_round = None, -1
    for name in os.listdir(os.path.join(ROOT, "profiles")) if os.path.isdir(os.path.join(ROOT, "profiles")) else []:
        m = re.match(r"r(\d+)_hbm_counters\.json$", name)
        if m and int(m.group(1)) > best_round:                 # (by round NUMBER: r10 comes after r9)
            best, best_round = name, int(m.group(1))
    if not best:
        return None, None
    with open(os.path.join(ROOT, "profiles", best)) as fh:
        d = json.load(fh)
    # the main launch (tier 0) of the L&R tile kernel: k_lr2_tile<rounds, 0, waves> (round 1: k_lr_tile<64, false, 0, ...>)
    best_k, best_v = None, None
    for k, v in d.items():
        main = ("k_lr2_tile<" in k and ", 0, " in k) or ("k_lr_tile<" in k and "false, 0, " in k)
        if main and "FETCH_SIZE" in v and "WRITE_SIZE" in v and "SQ_INSTS_VALU" in v:
            if best_v is None or v["SQ_INSTS_VALU"]["per_launch"] > best_v["SQ_INSTS_VALU"]["per_launch"]:
                best_k, best_v = k, v
    if best_v is not None:
        global PROFILED_VALU
        PROFILED_VALU = best_v["SQ_INSTS_VALU"]["per_launch"]
        return (2.0 * best_v["FETCH_SIZE"]["per_launch_KB"] + best_v["WRITE_SIZE"]["per_launch_KB"]) * 1024.0, "profiles/" + best
    return None, None


def bench_trajectory(args, rank, world, local_rank):
    """Secondary line (BASELINE configs[4] proxy): frames of one 100k-atom system streamed FROM HOST
    MEMORY through freesasa_gpu_trajectory (PCIe-inclusive, copies overlapped with kernels)."""
    import freesasa_amd as fa
    import tools
    n_atoms, n_frames = 100_000, max(8, args.steps * 16)
    base, r = tools.globule(n_atoms, 5 + rank)
    frames = np.stack([tools.jitter(base, 100 + f, 0.5) for f in range(n_frames)])
    fa.trajectory(frames[:16], r, per_atom=True, device=local_rank)            # warm-up
    t0 = time.perf_counter()
    totals, _ = fa.trajectory(frames, r, per_atom=True, device=local_rank)
    dt = time.perf_counter() - t0
    if rank == 0:
        print(json.dumps({"metric": "atom-frames/sec SASA (L&R 20 slices), host-resident trajectory",
                          "value": world * n_atoms * n_frames / dt, "unit": "atoms/s", "n_gpus": world,
                          "steps": n_frames, "warmup": 16, "ms_per_step": 1e3 * dt / n_frames,
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
                          "data": "synthetic",
                          "config": {"workload": f"{n_frames} frames x {n_atoms} atoms (globule + 0.5 A jitter), frames in pageable "
                                                 "host memory, staged through page-locked buffers by the driver's host lanes, per-atom "
                                                 "SASA streamed back; PCIe-inclusive",
                                     "mean_total": float(np.mean(totals))}}), flush=True)


def rank_driver_workloads(fa, tools, torch, dist, dev, rank, world, local_rank, dry):
    """N > 1 (the driver's 2 / 4 / 8-GPU runs): BASELINE configs[3] and configs[4] through the REAL drivers, one process per
    GPU - every rank sweeps the binary cache and streams the frame file on its own device (weak scaling: the per-GPU work
    is the 1-GPU keys' kind, the ranks share the input files' page cache and write their own result files), bracketed by
    barriers, the slowest rank's time, whole-job rates.  What this measures that the resident headline cannot: the host
    budget - file reads, PCIe both ways, result writes and the CPUs the cgroup grants, divided among the ranks
    (LOCAL_WORLD_SIZE: gpu_drivers.hip, process_cpus) - at N devices.  Collective: every rank calls it."""
    from freesasa_amd import ingest
    scratch = os.path.join(os.environ.get("FREESASA_AMD_BENCH_CACHE", "/tmp"), f"freesasa_amd_bench_u{os.getuid()}")
    os.makedirs(scratch, exist_ok=True)
    n_atoms, n_frames = 100_000, 240
    frames, cache = os.path.join(scratch, "frames240.f64"), os.path.join(scratch, "ranks.fsab")
    base, r = tools.globule(n_atoms, 5) if not dry else (np.zeros(3 * 8), np.ones(8))
    cache_atoms = 0
    if rank == 0 and not dry:                                   # the inputs, once per session
        if not (os.path.exists(frames) and os.path.getsize(frames) == 24 * n_atoms * n_frames):
            with open(frames + ".tmp", "wb") as fh:
                for f in range(n_frames):
                    tools.jitter(base, 100 + f, 0.5).tofile(fh)
            os.replace(frames + ".tmp", frames)
        if not os.path.exists(cache):
            pdb_dir = os.path.join(ROOT, "tests", "golden", "pdb")
            texts = [open(os.path.join(pdb_dir, nm + ".pdb"), "rb").read() for nm in PDB_NAMES]
            ingest.load_pdb_texts(texts * 250).save(cache + ".tmp")
            os.replace(cache + ".tmp", cache)
    dist.barrier()

    def timed(fn, units):
        fn()                                                    # warm-up: contexts, staging, tile shapes of every lane
        if not dry:
            torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        fn()
        if not dry:
            torch.cuda.synchronize()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return {"value": world * units / float(t.item()), "seconds": float(t.item()), "ranks": world}

    out = {}
    tp, sp = os.path.join(scratch, f"tot_rank{rank}.bin"), os.path.join(scratch, f"sasa_rank{rank}.bin")

    def traj():
        if dry:
            time.sleep(0.002); return
        for q in (tp, sp):
            if os.path.exists(q): os.unlink(q)
        fa.trajectory_file(frames, r, tp, sp, device=local_rank)
    res = timed(traj, n_atoms * n_frames)
    res.update({"unit": "atom-frames/s", "workload": f"every rank: {n_frames} frames x {n_atoms} atoms from one shared frame file, totals + per-atom areas to its own files"})
    out["trajectory_file"] = res
    if not dry:
        c = ingest.Cache(cache); cache_atoms = c.n_atoms; c.close()

    def sweep():
        if dry:
            time.sleep(0.002); return
        fa.sweep_cache(cache, device=local_rank)
    res = timed(sweep, cache_atoms)
    res.update({"unit": "atoms/s", "atoms_per_rank": int(cache_atoms), "workload": "every rank: the version-2 cache file of the reference's PDB entries x 250, read + verified + swept on its own device"})
    out["sweep_cache"] = res
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--structs", type=int, default=1000, help="structures per GPU")
    ap.add_argument("--atoms", type=int, default=10000, help="atoms per structure")
    ap.add_argument("--slices", type=int, default=20)
    ap.add_argument("--workload", default="coil_lr", choices=["coil_lr", "globule_sr", "traj_lr", "sweep_lr", "coil_sr", "pdb_lr", "pdb_sr"],
                    help="coil_lr: the headline metric (default).  globule_sr: BASELINE configs[1] proxy, "
                         "ONE 200k-atom globule per GPU, Shrake-Rupley 100 points (secondary line).  sweep_lr: configs[3] proxy, "
                         "structures of log-uniform size 500..50 000 atoms dealt to the ranks by LPT on atom count.  coil_sr: the headline "
                         "batch through Shrake-Rupley; pdb_lr / pdb_sr: the reference's PDB entries x 251 (what the secondary keys and their counter passes run)")
    ap.add_argument("--points", type=int, default=100)
    ap.add_argument("--sync-entry", action="store_true", help="time freesasa_gpu_lr_batch_dev (one synchronous call per step) instead of the asynchronous batch entry")
    ap.add_argument("--no-live-counters", action="store_true", help="do not re-run under rocprofv3 for roofline.traffic / valu_issue (the committed profile is quoted instead)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true")
    ap.add_argument("--no-neighbors", action="store_true", help="skip the neighbor count (one launch of the kernel's neighbor phase): the counter passes of tools/gpu_round.sh want the tile kernel's own launches only")
    ap.add_argument("--no-secondary", action="store_true", help="skip the short measurements of the other single-GPU configurations")
    ap.add_argument("--no-drivers", action="store_true", help="skip the file sweep / cache sweep / trajectory-file / latency keys (they write ~4 GB of scratch files under /tmp)")
    ap.add_argument("--sustain-seconds", type=float, default=2.0, help="length of the sustained run reported next to the headline (0: none)")
    ap.add_argument("--dry-run", action="store_true",
                    help="TEST ONLY (tests/test_distributed.py): the rank / argument / JSON plumbing on CPU under gloo, "
                         "with a stand-in for the engine that computes nothing; prints \"dry_run\": true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import freesasa_amd as fa
    import tools

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dry = args.dry_run
    # FREESASA_AMD_BENCH_FORCE_DIST=1 (tests/test_distributed.py, -m gpu): the N-rank plumbing - init_process_group("nccl"),
    # barrier, both all_reduces, destroy - with ONE rank, so that RCCL is initialised by this code on an MI355X before the
    # driver's 8-GPU run ever happens
    dist_on = world > 1 or (os.environ.get("FREESASA_AMD_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ)
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if dry:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if dry:
        dev = torch.device("cpu")
    else:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)

    def sync():
        if not dry:
            torch.cuda.synchronize()

    if args.workload == "traj_lr":
        return bench_trajectory(args, rank, world, local_rank)
    sr = args.workload in ("globule_sr", "coil_sr", "pdb_sr")
    if args.workload == "globule_sr":
        args.structs, args.atoms = 1, 200_000
        xyz, r = tools.globule(args.atoms, 77 + rank)
        offs = np.array([0, args.atoms], dtype=np.int64)
    elif args.workload in ("pdb_lr", "pdb_sr"):
        xyz, r, offs, _, _ = real_pdb_batch()
        args.structs, args.atoms = len(offs) - 1, int(offs[-1] // (len(offs) - 1))
    elif args.workload == "sweep_lr":
        # whole-PDB sweep proxy: one global list of ragged structures, dealt to the ranks by
        # longest-processing-time-first on the atom count (freesasa_amd/shard.py); weak scaling:
        # the list grows with the number of ranks
        from freesasa_amd import shard
        if args.structs == 1000:
            args.structs = 400
        sizes = np.exp(np.random.default_rng(2024).uniform(np.log(500), np.log(50000), args.structs * world)).astype(np.int64)
        mine = shard.lpt(sizes, world)[rank]
        parts = [tools.coil(int(sizes[k]), 5000 + int(k)) for k in mine]
        xyz = np.concatenate([p[0] for p in parts])
        r = np.concatenate([p[1] for p in parts])
        offs = np.concatenate([[0], np.cumsum(sizes[mine])]).astype(np.int64)
        args.structs, args.atoms = len(mine), int(np.mean(sizes[mine]))
    else:
        # this rank's shard: its own independent structures (seeds are disjoint across ranks)
        # (generated once per session: the 1 / 2 / 4 / 8-GPU runs of the driver read rank k's batch back from /tmp)
        xyz, r, offs = tools.coil_batch(args.structs, args.atoms, seed0=1000 + rank * args.structs,
                                        cache_dir=os.environ.get("FREESASA_AMD_BENCH_CACHE", "/tmp") if args.structs * args.atoms >= 1_000_000 else None)
    n_atoms = int(offs[-1])
    d_xyz = torch.from_numpy(xyz).to(dev)
    d_r = torch.from_numpy(r).to(dev)
    d_sasa = torch.empty(n_atoms, dtype=torch.float64, device=dev)
    d_tot = torch.empty(args.structs, dtype=torch.float64, device=dev)
    sync()
    if dry:
        class DryContext:                       # TEST ONLY: stands in for the engine, computes nothing
            def lee_richards(self, *a, **k): time.sleep(0.002)
            shrake_rupley = lee_richards
            def stats(self): return {"ms_kernel": 0.0, "ms_prep": 0.0, "max_neighbors": 0, "fallback_tiles": 0, "tile_atoms": 0,
                                     "block_threads": 0, "lds_bytes": 0, "n_cells": 0}
            def close(self): pass
        ctx = DryContext()
    else:
        stream = torch.cuda.Stream(device=dev)      # the engine launches on this torch stream
        torch.cuda.set_stream(stream)
        ctx = fa.GpuContext(local_rank, stream=stream.cuda_stream, timing=True)

    d_cnt = torch.empty(n_atoms, dtype=torch.int32, device=dev) if sr else None

    # Lee-Richards steps go through the asynchronous batch entry (freesasa_gpu_lr_batch_dev_async): a step is enqueued
    # while the one before it runs, up to two in flight, and everything is collected (ctx.wait) inside the timed region.
    # Same kernels, bit-identical outputs (checked below against one synchronous step); --sync-entry times the
    # synchronous entry instead.
    use_async = not sr and not dry and not args.sync_entry

    def step():
        if sr:
            ctx.shrake_rupley(d_xyz.data_ptr(), d_r.data_ptr(), offs, d_sasa.data_ptr(), d_cnt.data_ptr(),
                              d_tot.data_ptr(), probe=1.4, n_points=args.points)
        elif use_async:
            ctx.lee_richards_async(d_xyz.data_ptr(), d_r.data_ptr(), offs, d_sasa.data_ptr(), d_tot.data_ptr(),
                                   probe=1.4, n_slices=args.slices)
        else:
            ctx.lee_richards(d_xyz.data_ptr(), d_r.data_ptr(), offs, d_sasa.data_ptr(), d_tot.data_ptr(),
                             probe=1.4, n_slices=args.slices)

    def drain():
        if use_async:
            ctx.wait()

    def barrier():
        sync()
        if dist_on:
            dist.barrier()
        sync()

    for _ in range(args.warmup):
        step()
    drain()
    import gc
    gc.collect(); gc.disable()              # (no cyclic-collector pause inside the timed region: see secondary_workloads.run)
    barrier()
    k_ms, prep_ms = [], []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        st = ctx.stats()                    # (asynchronous entry: of the step collected last, two calls back)
        k_ms.append(st["ms_kernel"])
        prep_ms.append(st["ms_prep"])
    drain()
    barrier()
    elapsed = time.perf_counter() - t0
    gc.enable()
    if use_async:
        st = ctx.stats()
        k_ms = k_ms[2:] + [st["ms_kernel"]] if len(k_ms) > 2 else [st["ms_kernel"]]
        prep_ms = prep_ms[2:] + [st["ms_prep"]] if len(prep_ms) > 2 else [st["ms_prep"]]
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if dist_on:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    na = torch.tensor([float(n_atoms)], dtype=torch.float64, device=dev)   # ragged shards differ per rank
    if dist_on:
        dist.all_reduce(na, op=dist.ReduceOp.SUM)
    atoms_all_ranks = int(na.item())
    st = ctx.stats()
    # the same loop held for >= --sustain-seconds: under sustained load the chip does not keep the clock of a 0.2 s burst
    # (DESIGN.md 8), so the figure a long job sees is reported next to the headline (every rank runs it; rank 0 reports)
    sustained = None
    if not dry and args.sustain_seconds > 0 and args.workload == "coil_lr":
        n_sus, t_sus = 0, time.perf_counter()
        while True:
            for _ in range(max(4, args.steps)):
                step()
            n_sus += max(4, args.steps)
            if time.perf_counter() - t_sus >= args.sustain_seconds:
                break
        drain()
        sync()
        dt_sus = time.perf_counter() - t_sus
        sustained = {"value": world * n_atoms * n_sus / dt_sus, "unit": "atoms/s", "steps": n_sus, "seconds": dt_sus, "ms_per_step": 1e3 * dt_sus / n_sus,
                     "note": "the timed loop continued for >= %.1f s (not max-reduced over ranks)" % args.sustain_seconds}

    multi = None
    if world > 1 and args.workload == "coil_lr" and not args.no_drivers:
        try:
            multi = rank_driver_workloads(fa, tools, torch, dist, dev, rank, world, local_rank, dry)
        except Exception as exc:                    # (a secondary key must not take the scaling line down - but every rank must leave the collectives together)
            multi = {"error": repr(exc)}
    if rank == 0:
        total_atoms = atoms_all_ranks * args.steps
        value = total_atoms / elapsed
        kern_s = float(np.mean(k_ms)) * 1e-3
        achieved = ALGO_BYTES_PER_ATOM * n_atoms / kern_s / 1e9 if kern_s > 0 else None
        traffic, traffic_src = profiled_traffic(args)          # the committed profile of the last round ...
        live = live_counters(args) if (world == 1 and not dry) else None
        if live:                                               # ... unless the counters can be read now
            global PROFILED_VALU
            traffic, PROFILED_VALU, traffic_src = live
        if sr:
            traffic = None
            metric = f"atoms/sec SASA (S&R {args.points} points)"
            wl = (f"one synthetic {args.atoms}-atom globule per GPU (BASELINE configs[1] proxy: 4V6X is not available "
                  f"offline), Shrake-Rupley {args.points} test points, probe 1.4 A, inputs resident in HBM") if args.workload == "globule_sr" else \
                 (f"{args.structs} structures, {n_atoms} atoms per GPU ({args.workload}), Shrake-Rupley {args.points} test points, probe 1.4 A, inputs resident in HBM")
        elif args.workload == "sweep_lr":
            traffic = None
            metric = f"atoms/sec SASA (L&R {args.slices} slices), ragged sweep"
            wl = (f"{args.structs} synthetic random-coil structures per GPU of log-uniform size 500..50000 atoms "
                  f"({n_atoms} atoms on rank 0; BASELINE configs[3] proxy), dealt to the ranks by LPT on atom count, "
                  f"Lee-Richards {args.slices} slices, probe 1.4 A, inputs resident in HBM")
        else:
            metric = "atoms/sec SASA (L&R 20 slices)" if args.slices == 20 else f"atoms/sec SASA (L&R {args.slices} slices)"
            wl = (f"{args.structs} synthetic random-coil structures x {args.atoms} atoms per GPU "
                  f"(BASELINE configs[2] batch geometry, seeds 1000+k), Lee-Richards "
                  f"{args.slices} slices, probe 1.4 A, inputs resident in HBM")
        out = {
            "metric": metric,
            "value": value, "unit": "atoms/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": wl,
                       "structures_per_gpu": args.structs, "atoms_per_structure": args.atoms,
                       "n_slices": args.slices, "probe_radius": 1.4,
                       "parallelism": f"{world} x independent structure shards (no collective)",
                       "entry": ("freesasa_gpu_lr_batch_dev_async: steps enqueued back to back, at most two in flight, all collected inside the timed region"
                                 if use_async else "synchronous batch entry, one call per step"),
                       "max_neighbors_per_atom": st["max_neighbors"], "fallback_tiles": st["fallback_tiles"],
                       "tile_atoms": st["tile_atoms"], "block_threads": st["block_threads"],
                       "lds_bytes_per_block": st["lds_bytes"], "cells": st["n_cells"]},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS if achieved else None, "traffic": traffic,
                         "traffic_unit": "bytes per launch", "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": ALGO_BYTES_PER_ATOM * n_atoms,
                         "kernel": "k_sr_tile" if sr else ("k_lr2_tile" if args.slices <= 256 else "k_lr_tile"), "kernel_ms": 1e3 * kern_s, "prep_ms": float(np.mean(prep_ms)),
                         "kernel_atoms_per_s": n_atoms / kern_s if kern_s > 0 else None,
                         "note": "nominal HBM roofline per north_star (40 B/atom); the kernel is fp64-VALU bound, "
                                 "see DESIGN.md",
                         # the roofline that binds: share of the VALU issue slots of 1024 SIMDs the kernel fills
                         "valu_issue": None if not (PROFILED_VALU and kern_s > 0 and not sr and args.workload == "coil_lr") else {
                             "wave_instructions_per_launch": PROFILED_VALU,
                             "wave_instructions_per_atom": PROFILED_VALU / n_atoms,
                             "frac_of_issue_slots": PROFILED_VALU * CYCLES_PER_VALU / (kern_s * SIMDS * CLOCK_HZ),
                             "source": traffic_src}},
        }
        if sustained:
            out["sustained"] = sustained
        if dist_on:
            out["config"]["process_group"] = f"{dist.get_backend()} world {dist.get_world_size()} (barrier + 2 all_reduce around the timed region)"
        if multi is not None:
            out["drivers_all_ranks"] = multi
        if dry:
            out["dry_run"] = True
            out["atoms_all_ranks"] = atoms_all_ranks
            print(json.dumps(out), flush=True)
            if dist_on:
                dist.barrier()
                dist.destroy_process_group()
            return
        if not sr and not args.no_neighbors:
            out["config"]["avg_neighbors_per_atom"] = neighbors_per_atom(fa, torch, d_xyz, d_r, offs, dev, local_rank)
        if use_async and world == 1:
            # the synchronous entry on the same context: the same bits, and its rate for comparison
            got_async = d_sasa.clone()
            n_sync = max(5, args.steps // 2)
            ctx.lee_richards(d_xyz.data_ptr(), d_r.data_ptr(), offs, d_sasa.data_ptr(), d_tot.data_ptr(), probe=1.4, n_slices=args.slices)  # (untimed: the first call behind the neighbor-count pass)
            torch.cuda.synchronize()
            ts = time.perf_counter()
            for _ in range(n_sync):
                ctx.lee_richards(d_xyz.data_ptr(), d_r.data_ptr(), offs, d_sasa.data_ptr(), d_tot.data_ptr(), probe=1.4, n_slices=args.slices)
            torch.cuda.synchronize()
            dts = (time.perf_counter() - ts) / n_sync
            out["synchronous_entry"] = {"value": n_atoms / dts, "unit": "atoms/s", "ms_per_step": 1e3 * dts,
                                        "identical_outputs": bool(torch.equal(got_async, d_sasa))}
            if not sr and "FREESASA_AMD_PRUNE" not in os.environ:
                # round 6: the same batch with the kernel's contained-caps phase switched off (neighbors whose
                # cap lies inside another neighbor's are dropped before the pair records are made) - the same bits, and
                # what the phase is worth on this box (the variable is read per call; removed again before anything else runs)
                os.environ["FREESASA_AMD_PRUNE"] = "0"
                try:
                    ctx.lee_richards(d_xyz.data_ptr(), d_r.data_ptr(), offs, d_sasa.data_ptr(), d_tot.data_ptr(), probe=1.4, n_slices=args.slices)
                    torch.cuda.synchronize()
                    ts = time.perf_counter()
                    for _ in range(n_sync):
                        ctx.lee_richards(d_xyz.data_ptr(), d_r.data_ptr(), offs, d_sasa.data_ptr(), d_tot.data_ptr(), probe=1.4, n_slices=args.slices)
                    torch.cuda.synchronize()
                    dto = (time.perf_counter() - ts) / n_sync
                finally:
                    del os.environ["FREESASA_AMD_PRUNE"]
                out["contained_caps_off"] = {"value": n_atoms / dto, "unit": "atoms/s", "ms_per_step": 1e3 * dto,
                                             "identical_outputs": bool(torch.equal(got_async, d_sasa)),
                                             "note": "synchronous entry with FREESASA_AMD_PRUNE=0; compare with synchronous_entry"}
                ctx.lee_richards(d_xyz.data_ptr(), d_r.data_ptr(), offs, d_sasa.data_ptr(), d_tot.data_ptr(), probe=1.4, n_slices=args.slices)  # (back to the shipped launch state)
                torch.cuda.synchronize()
        if world == 1 and args.workload == "coil_lr" and not args.no_secondary and (args.structs, args.atoms, args.slices) == (1000, 10000, 20):
            out.update(secondary_workloads(fa, torch, tools, d_xyz, d_r, offs, xyz, r, dev, local_rank, check=not args.no_cpu_baseline))
        if world == 1 and args.workload == "coil_lr" and not args.no_secondary and (args.structs, args.atoms, args.slices) == (1000, 10000, 20):
            out.update(sr_workloads(fa, torch, tools, d_xyz, d_r, offs, xyz, r, dev, local_rank, not args.no_cpu_baseline, args))
        if world == 1 and args.workload == "coil_lr" and not args.no_drivers and (args.structs, args.atoms, args.slices) == (1000, 10000, 20):
            scratch = os.path.join(os.environ.get("FREESASA_AMD_BENCH_CACHE", "/tmp"), f"freesasa_amd_bench_u{os.getuid()}")
            os.makedirs(scratch, exist_ok=True)
            try:
                out["latency_us"] = latency_us(fa, tools)
            except Exception as exc:
                out["latency_us"] = {"error": repr(exc)}
            out.update(driver_workloads(fa, tools, local_rank, scratch))
        if world == 1 and args.workload == "coil_lr" and not args.no_end_to_end:
            out["two_passes_in_flight"] = two_streams(fa, torch, d_xyz, d_r, offs, args, dev, local_rank, args.steps)
            out["end_to_end"] = end_to_end(fa, torch, xyz, r, offs, args, local_rank, d_sasa.cpu().numpy())
        if world == 1 and not args.no_cpu_baseline and args.workload == "coil_lr":
            base, err = cpu_baseline(args, offs, d_sasa.cpu().numpy())
            out["cpu_baseline"] = base
            out["max_abs_dsasa_vs_cpu"] = err
        print(json.dumps(out), flush=True)
    ctx.close()
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
