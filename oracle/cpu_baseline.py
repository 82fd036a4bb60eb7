#!/usr/bin/env python3
"""CPU baseline of bench.py: the reference's own CPU path timed on this box's host cores.
TEST/MEASUREMENT INFRASTRUCTURE ONLY (executed by bench.py's cpu_baseline leg; never by the product).

The library timed is the real reference (oracle/_ref/libfreesasa_ref.so, kind "reference") when it was
built, else the plain-C port (oracle/libsasa_oracle.so, kind "port").  Two ways of using the host cores,
as SURVEY.md 8(d) lists them:

  per-structure threads   ONE structure, the reference's own pthreads split of the atoms
                          (src/sasa_lr.c:219-253), n_threads in {1, 2 (its default), 16 (its maximum)};
  all cores, batched      one PROCESS per core (fork; every process loads the library itself, so the
                          reference's ~4 mallocs per atom, src/nb.c:260-321, do not meet in one glibc
                          arena), each running whole structures with n_threads = 1.

This script never imports torch or HIP: it can fork.  Structures are the bench's own (tools.coil, seeds
seed0 + k), regenerated here.  Prints one JSON object; with --out writes the per-atom areas of the
first --keep structures (bench.py compares them with the GPU's).
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

_calc = None


def _init():
    """Per process: load the library (after the fork, so that it is this process's own copy of its state)."""
    global _calc
    import oracle
    if oracle.Reference.available():
        ref = oracle.Reference()
        ref.lib.freesasa_set_verbosity(2)

        def calc(xyz, r, threads, slices):
            return ref.calc_coord(xyz, r, oracle.LEE_RICHARDS, 1.4, n_slices=slices, n_threads=threads)[0]
        _calc = (calc, "reference")
    else:
        orc = oracle.Oracle()

        def calc(xyz, r, threads, slices):
            return orc.lee_richards(xyz, r, 1.4, slices)
        _calc = (calc, "port")


def _work(job):
    """Structures k0..k1 of the batch, one after the other; returns (atoms, seconds, areas of the kept ones)."""
    import tools
    k0, k1, atoms, seed0, slices, keep = job
    if _calc is None:
        _init()
    structs = [tools.coil(atoms, seed0 + k) for k in range(k0, k1)]  # generation is not timed
    t0 = time.perf_counter()
    out = [_calc[0](x, r, 1, slices) for x, r in structs]
    dt = time.perf_counter() - t0
    return (k1 - k0) * atoms, dt, [out[k - k0] for k in range(k0, min(k1, keep))]


def cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_limit():
    """CPUs this process may actually use at once: the affinity mask, capped by the cgroup's CPU quota (a
    container can show 256 logical CPUs and be allowed ten of them)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:                      # cgroup v2: "<quota> <period>" or "max <period>"
            q, p = fh.read().split()
            if q != "max":
                quota = float(q) / float(p)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fq, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fp:
                q, p = float(fq.read()), float(fp.read())
                if q > 0:
                    quota = q / p
        except (OSError, ValueError):
            pass
    return n, quota


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--atoms", type=int, default=10000)
    ap.add_argument("--structs", type=int, default=1000, help="structures of the batch the sample is taken from")
    ap.add_argument("--seed0", type=int, default=1000)
    ap.add_argument("--slices", type=int, default=20)
    ap.add_argument("--budget", type=float, default=12.0, help="seconds of wall time for the all-cores mode")
    ap.add_argument("--keep", type=int, default=2, help="structures whose per-atom areas are written to --out")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()

    import tools
    logical, quota = cpu_limit()
    cores = logical if quota is None else max(1, min(logical, int(quota + 0.5)))
    if os.environ.get("FREESASA_CPU_BASELINE_PROCS"):  # dev: scaling curve (tools/dev/cpu_scaling.py)
        cores = max(1, int(os.environ["FREESASA_CPU_BASELINE_PROCS"]))
    _init()
    kind = _calc[1]
    xyz, r = tools.coil(args.atoms, args.seed0)

    # mode 1: one structure, the reference's own thread split
    per_structure = {}
    for nt in (1, 2, 16):
        best = None
        for _ in range(2):
            t0 = time.perf_counter()
            _calc[0](xyz, r, nt, args.slices)
            dt = time.perf_counter() - t0
            best = dt if best is None or dt < best else best
        per_structure[str(nt)] = args.atoms / best
        if kind == "port":
            break  # the port has no threads
    single = per_structure["1"]

    # mode 2: all cores, one process per core, whole structures with n_threads = 1
    t_struct = args.atoms / single
    per_proc = max(1, int(args.budget / t_struct))
    per_proc = min(per_proc, max(1, args.structs // cores))
    sample = min(args.structs, per_proc * cores)
    procs = min(cores, sample)
    jobs = []
    for p in range(procs):
        k0, k1 = p * sample // procs, (p + 1) * sample // procs
        if k1 > k0:
            jobs.append((k0, k1, args.atoms, args.seed0, args.slices, args.keep))
    ctx = mp.get_context("fork")
    with ctx.Pool(processes=len(jobs), initializer=_init) as pool:
        pool.map(_work, [(0, 0, args.atoms, args.seed0, args.slices, 0)] * len(jobs))  # start every process, load the library
        t0 = time.perf_counter()
        res = pool.map(_work, jobs, chunksize=1)
        wall = time.perf_counter() - t0
    atoms = sum(x[0] for x in res)
    busy = max(x[1] for x in res)  # the slowest process (structure generation excluded)
    kept = [a for x in res for a in x[2]]
    if args.out and kept:
        np.savez(args.out, **{f"s{k}": a for k, a in enumerate(kept)})
    all_cores = atoms / busy
    print(json.dumps({
        "value": all_cores, "unit": "atoms/s", "cores": len(jobs), "kind": kind,
        "sample": f"first {sample} of {args.structs} structures ({atoms} atoms), one process per core "
                  f"({len(jobs)} processes, each with its own copy of the library), n_threads=1, L&R {args.slices} slices; "
                  f"slowest process {busy:.2f} s (wall incl. structure generation {wall:.2f} s)",
        "cpu_model": cpu_model(), "host_logical_cpus": logical, "cgroup_cpu_quota": quota, "host_cores": cores,
        "single_thread": single,
        "per_structure_threads": {"n_threads": per_structure, "unit": "atoms/s",
                                  "note": "one 10k-atom structure, the reference's own pthreads split (src/sasa_lr.c:219-253)"},
        "all_cores_vs_cores_x_single": all_cores / (len(jobs) * single),
    }))


if __name__ == "__main__":
    main()
