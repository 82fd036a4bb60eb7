#!/usr/bin/env python3
"""Whole-directory sweep: PDB files -> batched ingestion on host threads -> SASA on the GPU ->
per-structure totals (BASELINE configs[3] in miniature; SURVEY §8f N1 + the batch entry point).

    python tools/sweep.py [--replicate N] [--threads T] [--batch-atoms A] [--devices 0,1,...] [--done FILE] [--cache FILE] [paths ...]

Without paths it sweeps the PDB fixtures under tests/golden/pdb, replicated N times.  Loading of
batch k+1 runs on host threads while the GPU computes batch k.  Prints one JSON line with the
end-to-end rate, the loader-only rate and (if oracle/_ref is present) the reference reader's
single-thread rate on the same files.  --devices: the GPUs that share the batches (freesasa_gpu_sweep_files_devices;
default: every visible device; entries may repeat); --done: a done-list, so that an interrupted sweep resumes — on any
device list; --cache FILE: sweep the binary cache FILE instead (written first from the files if it does not exist)."""
import argparse
import glob
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("paths", nargs="*")
    ap.add_argument("--replicate", type=int, default=200)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--batch-atoms", type=int, default=0, help="atoms per batch (0: the driver's default: 5e5 with the parser on the device, 1e6 with the host parser)")
    ap.add_argument("--host-parser", action="store_true", help="parse the files with the host loader (default: ON THE DEVICE, csrc/gpu_parse.hip; files it refuses go to the host parser anyway)")
    ap.add_argument("--slices", type=int, default=20)
    ap.add_argument("--no-gpu", action="store_true", help="time the loader only")
    ap.add_argument("--devices", default="", help="comma-separated device list (default: all visible devices)")
    ap.add_argument("--done", default=None, help="done-list file: resume an interrupted sweep")
    ap.add_argument("--cache", default=None, help="binary cache file to sweep (created from the files when missing)")
    ap.add_argument("--engine", choices=["python", "c"], default="c",
                    help="c: freesasa_gpu_sweep_files (loader thread || GPU inside the library); "
                         "python: the same pipeline written with the two-step Python API")
    args = ap.parse_args()
    import freesasa_amd as fa
    from freesasa_amd import ingest

    paths = args.paths or [p for p in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "pdb", "*.pdb")))
                           if os.path.getsize(p) > 10_000]
    paths = paths * args.replicate
    sizes = np.array([os.path.getsize(p) for p in paths])
    # batches of roughly equal atom count (~81 bytes per ATOM line)
    per_batch = max(1, int(args.batch_atoms * 81 / max(1.0, sizes.mean())))
    chunks = [paths[i:i + per_batch] for i in range(0, len(paths), per_batch)]

    t0 = time.perf_counter()
    probe = ingest.load_pdb_files(chunks[0], n_threads=args.threads)
    t_load1 = time.perf_counter() - t0
    out = {"files": len(paths), "batches": len(chunks), "threads": args.threads or os.cpu_count(),
           "loader_atoms_per_s": probe.n_atoms / t_load1, "loader_MB_per_s": sum(os.path.getsize(p) for p in chunks[0]) / t_load1 / 1e6}

    devices = [int(d) for d in args.devices.split(",") if d != ""] or list(range(max(1, fa.device_count())))
    out["devices"] = devices
    if not args.no_gpu and args.cache:
        if not os.path.exists(args.cache):
            t0 = time.perf_counter()
            ingest.load_pdb_files(paths, n_threads=args.threads).save(args.cache)
            out["cache_build_seconds"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        totals, _, atoms, status = fa.sweep_cache(args.cache, fa.LEE_RICHARDS, resolution=args.slices, batch_atoms=args.batch_atoms, devices=devices)
        dt = time.perf_counter() - t0
        ok = status == 0
        out.update({"engine": "freesasa_gpu_sweep_cache_devices", "atoms": int(atoms.sum()), "structures": int(ok.sum()),
                    "failed_inputs": int((~ok).sum()), "seconds": dt, "end_to_end_atoms_per_s": float(atoms.sum()) / dt,
                    "structures_per_s": float(ok.sum()) / dt, "mean_total_A2": float(totals[ok].mean())})
    elif not args.no_gpu and args.engine == "c":
        popt = 0 if args.host_parser else ingest.PARSE_ON_DEVICE
        fa.sweep_files(chunks[0][:20], fa.LEE_RICHARDS, resolution=args.slices, ingest_options=popt)                      # warm-up
        fa.sweep_parse_stats()
        t0 = time.perf_counter()
        if args.done:
            complete, totals, _, atoms, status = fa.sweep_files_resumable(paths, args.done, fa.LEE_RICHARDS, resolution=args.slices, n_threads=args.threads,
                                                                         batch_atoms=args.batch_atoms, devices=devices, ingest_options=popt)
            out["complete"] = bool(complete)
        else:
            totals, _, atoms, status = fa.sweep_files(paths, fa.LEE_RICHARDS, resolution=args.slices, n_threads=args.threads,
                                                      batch_atoms=args.batch_atoms, class_sums=True, devices=devices, ingest_options=popt)
        dt = time.perf_counter() - t0
        ok = status == 0
        on_dev, on_host = fa.sweep_parse_stats()
        out.update({"engine": "freesasa_gpu_sweep_files_devices", "parser": "host" if args.host_parser else "device",
                    "files_parsed_on_device": on_dev, "files_left_to_the_host_parser": on_host, "atoms": int(atoms.sum()), "structures": int(ok.sum()),
                    "failed_inputs": int((~ok).sum()), "seconds": dt, "end_to_end_atoms_per_s": float(atoms.sum()) / dt,
                    "structures_per_s": float(ok.sum()) / dt, "mean_total_A2": float(totals[ok].mean())})
    elif not args.no_gpu:
        totals, n_atoms, n_bad = [], 0, 0
        fa.calc_batch(probe.xyz[:3000], probe.radii[:3000], [0, 3000], fa.LEE_RICHARDS, resolution=args.slices)  # warm-up
        t0 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=1) as pool:              # the C loader brings its own threads
            fut = pool.submit(ingest.load_pdb_files, chunks[0], 0, args.threads)
            for k in range(len(chunks)):
                b = fut.result()
                if k + 1 < len(chunks):
                    fut = pool.submit(ingest.load_pdb_files, chunks[k + 1], 0, args.threads)
                ok = b.status == 0
                n_bad += int((~ok).sum())
                # empty structures are legal batch members only for the loader; drop them here
                keep = np.nonzero(ok)[0]
                offs = np.concatenate([[0], np.cumsum(np.diff(b.offsets)[keep])]).astype(np.int64)
                _, _, tot = fa.calc_batch(b.xyz, b.radii, offs, fa.LEE_RICHARDS, resolution=args.slices)
                totals.append(tot)
                n_atoms += b.n_atoms
        dt = time.perf_counter() - t0
        totals = np.concatenate(totals)
        out.update({"atoms": n_atoms, "structures": int(len(totals)), "failed_inputs": n_bad, "seconds": dt,
                    "end_to_end_atoms_per_s": n_atoms / dt, "structures_per_s": len(totals) / dt,
                    "mean_total_A2": float(totals.mean())})

    try:
        sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
        import make_ingest_golden as mg                                # needs oracle/_ref (build container)
        sample = paths[:min(len(paths), 40)]
        t0 = time.perf_counter()
        n = sum(mg.reference_view_unsafe(p, 0).get("n_atoms", 0) for p in sample)
        out["reference_reader_atoms_per_s_1thread"] = n / (time.perf_counter() - t0)
    except Exception as e:                                             # noqa: BLE001 - optional comparison
        out["reference_reader"] = f"not available here ({type(e).__name__})"
    print(json.dumps(out))


if __name__ == "__main__":
    main()
