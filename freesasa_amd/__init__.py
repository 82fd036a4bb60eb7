"""freesasa_amd — thin ctypes front-end of libfreesasa_amd.so (the MI355X SASA engine).

The product is the C library (include/freesasa_amd.h = the reference's calculation API,
include/freesasa_gpu.h = the additive batch API).  This module only loads it and wraps the
C entry points for the tests, the bench and Python callers; it contains no SASA arithmetic
and no CPU fallback — if the library or a HIP device is missing, calls fail loudly.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB_PATH = os.environ.get("FREESASA_AMD_LIB") or os.path.join(HERE, "lib", "libfreesasa_amd.so")

LEE_RICHARDS, SHRAKE_RUPLEY = 0, 1
SUCCESS, FAIL, WARN = 0, -1, -2
V_NORMAL, V_NOWARNINGS, V_SILENT, V_DEBUG = 0, 1, 2, 3

_dp, _ip, _lp = C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_int64)


class Parameters(C.Structure):
    """freesasa_parameters (include/freesasa_amd.h; reference src/freesasa.h:232-238)."""
    _fields_ = [("alg", C.c_int), ("probe_radius", C.c_double),
                ("shrake_rupley_n_points", C.c_int), ("lee_richards_n_slices", C.c_int),
                ("n_threads", C.c_int)]


class Result(C.Structure):
    """freesasa_result (reference src/freesasa.h:267-272)."""
    _fields_ = [("total", C.c_double), ("sasa", _dp), ("n_atoms", C.c_int),
                ("parameters", Parameters)]


class CoordT(C.Structure):
    """coord_t (reference src/coord.h:26-38)."""
    _fields_ = [("n", C.c_int), ("is_linked", C.c_int), ("xyz", _dp)]


class Stats(C.Structure):
    _fields_ = [("n_atoms", C.c_longlong), ("n_cells", C.c_longlong), ("n_structs", C.c_int),
                ("max_neighbors", C.c_int), ("fallback_tiles", C.c_int), ("tile_atoms", C.c_int),
                ("block_threads", C.c_int), ("lds_bytes", C.c_int), ("ms_prep", C.c_double),
                ("ms_kernel", C.c_double), ("ms_total", C.c_double)]


_lib = None


def build():
    """Compile the library in-tree (hipcc --offload-arch=gfx950; works without a GPU)."""
    subprocess.run(["make", "-C", ROOT, "all"], check=True, stdout=subprocess.DEVNULL)


def _preload_torch_hip_runtime():
    """libfreesasa_amd.so needs libamdhip64.so.7.  PyTorch-ROCm wheels bundle their own copy under
    the same soname; a process must not end up with ROCm's copy loaded first and torch's other
    runtime libraries later ("No HIP GPUs are available").  If torch is installed and not yet
    imported, load ITS libamdhip64 first so that both this library and a later `import torch`
    share one runtime.  Pure C callers are unaffected (no torch in the process)."""
    import importlib.util
    import sys
    if "torch" in sys.modules or os.environ.get("FREESASA_AMD_NO_TORCH_PRELOAD"):
        return
    try:
        spec = importlib.util.find_spec("torch")
        if spec and spec.origin:
            cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
            if os.path.exists(cand):
                C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except Exception:
        pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OSError(f"{LIB_PATH} is not built (run `make` or __graft_entry__.build()); "
                          "freesasa_amd has no pure-Python or CPU path")
        _preload_torch_hip_runtime()
        L = C.CDLL(LIB_PATH)
        L.freesasa_calc_coord.argtypes = [_dp, _dp, C.c_int, C.POINTER(Parameters)]
        L.freesasa_calc_coord.restype = C.POINTER(Result)
        L.freesasa_calc.argtypes = [C.POINTER(CoordT), _dp, C.POINTER(Parameters)]
        L.freesasa_calc.restype = C.POINTER(Result)
        L.freesasa_calc_structure.argtypes = [C.c_void_p, C.POINTER(Parameters)]
        L.freesasa_calc_structure.restype = C.POINTER(Result)
        L.freesasa_result_free.argtypes = [C.POINTER(Result)]
        L.freesasa_result_free.restype = None
        L.freesasa_lee_richards.argtypes = [_dp, C.POINTER(CoordT), _dp, C.POINTER(Parameters)]
        L.freesasa_shrake_rupley.argtypes = [_dp, C.POINTER(CoordT), _dp, C.POINTER(Parameters)]
        L.freesasa_set_verbosity.argtypes = [C.c_int]
        L.freesasa_get_verbosity.restype = C.c_int
        L.freesasa_gpu_device_count.restype = C.c_int
        L.freesasa_gpu_ctx_create.argtypes = [C.c_int, C.c_void_p]
        L.freesasa_gpu_ctx_create.restype = C.c_void_p
        L.freesasa_gpu_ctx_destroy.argtypes = [C.c_void_p]
        L.freesasa_gpu_ctx_destroy.restype = None
        L.freesasa_gpu_ctx_set_timing.argtypes = [C.c_void_p, C.c_int]
        L.freesasa_gpu_ctx_set_timing.restype = None
        L.freesasa_gpu_ctx_get_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
        L.freesasa_gpu_ctx_get_stats.restype = None
        L.freesasa_gpu_ctx_last_error.argtypes = [C.c_void_p]
        L.freesasa_gpu_ctx_last_error.restype = C.c_char_p
        L.freesasa_gpu_lr_batch_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, _lp, C.c_int,
                                                C.c_double, C.c_int, C.c_void_p, C.c_void_p]
        L.freesasa_gpu_lr_batch_dev_async.argtypes = L.freesasa_gpu_lr_batch_dev.argtypes
        L.freesasa_gpu_wait.argtypes = [C.c_void_p]
        L.freesasa_gpu_sr_batch_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, _lp, C.c_int,
                                                C.c_double, C.c_int, _dp, C.c_void_p, C.c_void_p,
                                                C.c_void_p]
        L.freesasa_gpu_segment_sums_dev.argtypes = [C.c_void_p, C.c_void_p, _lp, C.c_int, C.c_void_p]
        L.freesasa_gpu_class_sums_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, _lp, C.c_int, C.c_void_p]
        L.freesasa_gpu_residue_areas_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, _lp, C.c_int,
                                                     C.POINTER(C.c_short), _dp, C.c_int, C.c_void_p, C.c_void_p]
        L.freesasa_gpu_test_points.argtypes = [C.c_int, _dp]
        L.freesasa_gpu_test_points.restype = None
        L.freesasa_gpu_calc_batch.argtypes = [_dp, _dp, _lp, C.c_int, C.c_int, C.c_double, C.c_int,
                                              _dp, _ip, _dp, C.c_int, C.c_char_p, C.c_int]
        L.freesasa_gpu_calc_batch_devices.argtypes = [_dp, _dp, _lp, C.c_int, C.c_int, C.c_double, C.c_int, _dp, _ip, _dp,
                                                      _ip, C.c_int, C.c_char_p, C.c_int]
        L.freesasa_gpu_calc_batch_pipelined.argtypes = [_dp, _dp, _lp, C.c_int, C.c_int, C.c_double, C.c_int, _dp, _ip, _dp,
                                                        C.c_int, C.c_int, C.c_longlong, C.c_char_p, C.c_int]
        L.freesasa_gpu_lr_neighbors_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, _lp, C.c_int, C.c_double, C.c_void_p, C.c_void_p, C.c_int]
        L.freesasa_gpu_arc_union_dev.argtypes = [C.c_void_p, _dp, _ip, C.c_int, _dp]
        L.freesasa_gpu_release_pool.argtypes = []
        L.freesasa_gpu_release_pool.restype = None
        L.freesasa_gpu_test_fail_after.argtypes = [C.c_int]
        L.freesasa_gpu_test_fail_after.restype = None
        L.freesasa_host_test_fail_after.argtypes = [C.c_int]
        L.freesasa_host_test_fail_after.restype = C.c_int
        L.freesasa_gpu_shard_cuts.argtypes = [_lp, C.c_int, C.c_int, _ip]
        L.freesasa_gpu_shard_cuts.restype = None
        L.freesasa_gpu_sweep_files.argtypes = [C.POINTER(C.c_char_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int,
                                               C.c_longlong, _dp, _dp, _lp, _ip, C.c_int, C.c_char_p, C.c_int]
        L.freesasa_gpu_sweep_files_resumable.argtypes = [C.POINTER(C.c_char_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int,
                                                         C.c_longlong, _dp, _dp, _lp, _ip, C.c_char_p, C.c_longlong, C.c_int, C.c_char_p, C.c_int]
        L.freesasa_gpu_trajectory.argtypes = [_dp, _dp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int,
                                              C.c_int, _dp, _dp, C.c_int, C.c_char_p, C.c_int]
        L.freesasa_gpu_trajectory_file.argtypes = [C.c_char_p, C.c_int, C.c_longlong, _dp, C.c_int, C.c_longlong, C.c_int, C.c_double,
                                                   C.c_int, C.c_int, C.c_char_p, C.c_char_p, C.c_char_p, C.c_longlong, C.c_int,
                                                   C.POINTER(C.c_longlong), C.c_char_p, C.c_int]
        L.freesasa_gpu_sweep_files_devices.argtypes = [C.POINTER(C.c_char_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int,
                                                       C.c_longlong, _dp, _dp, _lp, _ip, C.c_char_p, C.c_longlong, _ip, C.c_int, C.c_char_p, C.c_int]
        L.freesasa_gpu_sweep_cache_devices.argtypes = [C.c_char_p, C.c_int, C.c_double, C.c_int, C.c_longlong, _dp, _dp, _lp, _ip, C.c_int,
                                                       _ip, C.c_int, C.c_int, C.c_char_p, C.c_int]
        L.freesasa_gpu_trajectory_devices.argtypes = [_dp, _dp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int,
                                                      C.c_int, _dp, _dp, _ip, C.c_int, C.c_char_p, C.c_int]
        L.freesasa_gpu_trajectory_file_devices.argtypes = [C.c_char_p, C.c_int, C.c_longlong, _dp, C.c_int, C.c_longlong, C.c_int, C.c_double,
                                                           C.c_int, C.c_int, C.c_char_p, C.c_char_p, C.c_char_p, C.c_longlong, _ip, C.c_int,
                                                           C.POINTER(C.c_longlong), C.c_char_p, C.c_int]
        _lib = L
    return _lib


def _devs(devices, device):
    """the device list of a driver call: `devices` (a list; entries may repeat) or the single `device`"""
    d = np.ascontiguousarray([device] if devices is None else list(devices), dtype=np.int32)
    return d, d.ctypes.data_as(_ip), int(d.size)


def device_count():
    return lib().freesasa_gpu_device_count()


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def calc_coord(xyz, radii, alg=LEE_RICHARDS, probe=1.4, n_points=100, n_slices=20, n_threads=1):
    """freesasa_calc_coord(): returns (per-atom sasa, total) or raises on NULL."""
    xyz, radii = _f64(xyz).reshape(-1), _f64(radii)
    p = Parameters(alg, probe, n_points, n_slices, n_threads)
    res = lib().freesasa_calc_coord(xyz.ctypes.data_as(_dp), radii.ctypes.data_as(_dp),
                                    radii.size, C.byref(p))
    if not res:
        raise RuntimeError("freesasa_calc_coord returned NULL (see the library's error output)")
    sasa = np.ctypeslib.as_array(res.contents.sasa, (radii.size,)).copy()
    total = res.contents.total
    lib().freesasa_result_free(res)
    return sasa, total


def calc_batch(xyz, radii, offsets, alg=LEE_RICHARDS, probe=1.4, resolution=20, device=-1):
    """freesasa_gpu_calc_batch() on host arrays: (sasa, counts-or-None, totals)."""
    xyz, radii = _f64(xyz).reshape(-1), _f64(radii)
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    n, ns = radii.size, offsets.size - 1
    sasa, totals = np.empty(n), np.empty(ns)
    counts = np.empty(n, dtype=np.int32) if alg == SHRAKE_RUPLEY else None
    err = C.create_string_buffer(512)
    ret = lib().freesasa_gpu_calc_batch(xyz.ctypes.data_as(_dp), radii.ctypes.data_as(_dp),
                                        offsets.ctypes.data_as(_lp), ns, alg, probe, resolution,
                                        sasa.ctypes.data_as(_dp),
                                        counts.ctypes.data_as(_ip) if counts is not None else None,
                                        totals.ctypes.data_as(_dp), device, err, 512)
    if ret:
        raise RuntimeError("freesasa_gpu_calc_batch: " + err.value.decode())
    return sasa, counts, totals


def calc_batch_devices(xyz, radii, offsets, devices, alg=LEE_RICHARDS, probe=1.4, resolution=20):
    """freesasa_gpu_calc_batch_devices(): (sasa, counts-or-None, totals); contiguous runs of structures of
    about equal atom count go to the listed devices (one host thread each; a device may repeat)."""
    xyz, radii = _f64(xyz).reshape(-1), _f64(radii)
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    devs = np.ascontiguousarray(devices, dtype=np.int32)
    n, ns = radii.size, offsets.size - 1
    sasa, totals = np.zeros(n), np.zeros(ns)
    counts = np.zeros(n, dtype=np.int32) if alg == SHRAKE_RUPLEY else None
    err = C.create_string_buffer(512)
    ret = lib().freesasa_gpu_calc_batch_devices(xyz.ctypes.data_as(_dp), radii.ctypes.data_as(_dp), offsets.ctypes.data_as(_lp), ns,
                                                alg, probe, resolution, sasa.ctypes.data_as(_dp),
                                                counts.ctypes.data_as(_ip) if counts is not None else None,
                                                totals.ctypes.data_as(_dp), devs.ctypes.data_as(_ip), devs.size, err, 512)
    if ret:
        raise RuntimeError("freesasa_gpu_calc_batch_devices: " + err.value.decode())
    return sasa, counts, totals


def calc_batch_pipelined(xyz, radii, offsets, alg=LEE_RICHARDS, probe=1.4, resolution=20, device=-1, lanes=0,
                         chunk_atoms=0, out=None):
    """freesasa_gpu_calc_batch_pipelined() on host arrays (numpy, or anything with .ctypes / data_ptr() such as a
    pinned torch tensor via its numpy view): (sasa, counts-or-None, totals).  `out` = (sasa, counts, totals) arrays
    to write into (e.g. page-locked ones); by default fresh numpy arrays."""
    xyz, radii = _f64(xyz).reshape(-1), _f64(radii)
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    n, ns = radii.size, offsets.size - 1
    if out is None:
        sasa, totals = np.empty(n), np.empty(ns)
        counts = np.empty(n, dtype=np.int32) if alg == SHRAKE_RUPLEY else None
    else:
        sasa, counts, totals = out
    err = C.create_string_buffer(512)
    ret = lib().freesasa_gpu_calc_batch_pipelined(xyz.ctypes.data_as(_dp), radii.ctypes.data_as(_dp), offsets.ctypes.data_as(_lp), ns,
                                                  alg, probe, resolution, sasa.ctypes.data_as(_dp),
                                                  counts.ctypes.data_as(_ip) if counts is not None else None,
                                                  totals.ctypes.data_as(_dp) if totals is not None else None,
                                                  device, lanes, chunk_atoms, err, 512)
    if ret:
        raise RuntimeError("freesasa_gpu_calc_batch_pipelined: " + err.value.decode())
    return sasa, counts, totals


def shard_cuts(offsets, n_parts):
    """freesasa_gpu_shard_cuts(): first structure of each of n_parts contiguous, atom-balanced runs (+ the end)."""
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    cuts = np.zeros(n_parts + 1, dtype=np.int32)
    lib().freesasa_gpu_shard_cuts(offsets.ctypes.data_as(_lp), offsets.size - 1, n_parts, cuts.ctypes.data_as(_ip))
    return cuts


def sweep_files(paths, alg=LEE_RICHARDS, probe=1.4, resolution=20, ingest_options=0, n_threads=0, batch_atoms=0,
                class_sums=True, device=-1, devices=None):
    """freesasa_gpu_sweep_files[_devices](): PDB / mmCIF files -> (totals[n], class_sums[n,3] or None, n_atoms[n],
    status[n]); loading of the next batch overlaps the GPU work on the current one.  devices: a list of devices
    (entries may repeat) that share the batches, largest first."""
    n = len(paths)
    arr = (C.c_char_p * n)(*[str(p).encode() for p in paths])
    totals, atoms, status = np.zeros(n), np.zeros(n, dtype=np.int64), np.zeros(n, dtype=np.int32)
    cls = np.zeros((n, 3)) if class_sums else None
    err = C.create_string_buffer(512)
    if devices is None:
        ret = lib().freesasa_gpu_sweep_files(arr, n, ingest_options, n_threads, alg, probe, resolution, batch_atoms,
                                             totals.ctypes.data_as(_dp), cls.ctypes.data_as(_dp) if cls is not None else None,
                                             atoms.ctypes.data_as(_lp), status.ctypes.data_as(_ip), device, err, 512)
    else:
        keep, dp_, nd = _devs(devices, device)
        ret = lib().freesasa_gpu_sweep_files_devices(arr, n, ingest_options, n_threads, alg, probe, resolution, batch_atoms,
                                                     totals.ctypes.data_as(_dp), cls.ctypes.data_as(_dp) if cls is not None else None,
                                                     atoms.ctypes.data_as(_lp), status.ctypes.data_as(_ip), None, 0, dp_, nd, err, 512)
    if ret:
        raise RuntimeError("freesasa_gpu_sweep_files: " + err.value.decode())
    return totals, cls, atoms, status


def sweep_files_resumable(paths, done_path, alg=LEE_RICHARDS, probe=1.4, resolution=20, ingest_options=0, n_threads=0,
                          batch_atoms=0, max_new_batches=0, device=-1, devices=None):
    """freesasa_gpu_sweep_files_resumable(): like sweep_files with a done-list at done_path (+ done_path.bin):
    returns (complete, totals, class_sums, n_atoms, status); batches listed there are not computed again."""
    n = len(paths)
    arr = (C.c_char_p * n)(*[str(p).encode() for p in paths])
    totals, atoms, status = np.zeros(n), np.zeros(n, dtype=np.int64), np.zeros(n, dtype=np.int32)
    cls = np.zeros((n, 3))
    err = C.create_string_buffer(512)
    if devices is None:
        ret = lib().freesasa_gpu_sweep_files_resumable(arr, n, ingest_options, n_threads, alg, probe, resolution, batch_atoms,
                                                       totals.ctypes.data_as(_dp), cls.ctypes.data_as(_dp), atoms.ctypes.data_as(_lp),
                                                       status.ctypes.data_as(_ip), str(done_path).encode(), max_new_batches, device, err, 512)
    else:
        keep, dp_, nd = _devs(devices, device)
        ret = lib().freesasa_gpu_sweep_files_devices(arr, n, ingest_options, n_threads, alg, probe, resolution, batch_atoms,
                                                     totals.ctypes.data_as(_dp), cls.ctypes.data_as(_dp), atoms.ctypes.data_as(_lp),
                                                     status.ctypes.data_as(_ip), str(done_path).encode(), max_new_batches, dp_, nd, err, 512)
    if ret < 0:
        raise RuntimeError("freesasa_gpu_sweep_files_resumable: " + err.value.decode())
    return ret == 0, totals, cls, atoms, status


def sweep_cache(cache_path, alg=LEE_RICHARDS, probe=1.4, resolution=20, batch_atoms=0, class_sums=True, device=-1, devices=None,
                lanes_per_device=0):
    """freesasa_gpu_sweep_cache_devices(): the sweep of a binary cache file (ingest.Batch.save) -> (totals[n],
    class_sums[n,3] or None, n_atoms[n], status[n]): only coordinates, radii and classes are read, verified piece by
    piece, by a few lanes per device."""
    from . import ingest
    c = ingest.Cache(cache_path)
    n = c.n_structs
    c.close()
    totals, atoms, status = np.zeros(n), np.zeros(n, dtype=np.int64), np.zeros(n, dtype=np.int32)
    cls = np.zeros((n, 3)) if class_sums else None
    err = C.create_string_buffer(512)
    keep, dp_, nd = _devs(devices, device)
    ret = lib().freesasa_gpu_sweep_cache_devices(str(cache_path).encode(), alg, probe, resolution, batch_atoms, totals.ctypes.data_as(_dp),
                                                 cls.ctypes.data_as(_dp) if cls is not None else None, atoms.ctypes.data_as(_lp),
                                                 status.ctypes.data_as(_ip), n, dp_, nd, lanes_per_device, err, 512)
    if ret:
        raise RuntimeError("freesasa_gpu_sweep_cache_devices: " + err.value.decode())
    return totals, cls, atoms, status


def trajectory(xyz_frames, radii, alg=LEE_RICHARDS, probe=1.4, resolution=20, frames_per_batch=0,
               per_atom=True, device=-1, devices=None):
    """freesasa_gpu_trajectory() on host arrays: xyz_frames [n_frames, n_atoms, 3] -> (totals
    [n_frames], per-atom [n_frames, n_atoms] or None)."""
    xyz_frames = np.ascontiguousarray(xyz_frames, dtype=np.float64)
    radii = _f64(radii)
    n_frames, n_atoms = xyz_frames.shape[0], radii.size
    assert xyz_frames.size == n_frames * n_atoms * 3
    totals = np.empty(n_frames)
    sasa = np.empty((n_frames, n_atoms)) if per_atom else None
    err = C.create_string_buffer(512)
    if devices is None:
        ret = lib().freesasa_gpu_trajectory(xyz_frames.ctypes.data_as(_dp), radii.ctypes.data_as(_dp), n_atoms,
                                            n_frames, alg, probe, resolution, frames_per_batch,
                                            totals.ctypes.data_as(_dp),
                                            sasa.ctypes.data_as(_dp) if sasa is not None else None,
                                            device, err, 512)
    else:
        keep, dp_, nd = _devs(devices, device)
        ret = lib().freesasa_gpu_trajectory_devices(xyz_frames.ctypes.data_as(_dp), radii.ctypes.data_as(_dp), n_atoms,
                                                    n_frames, alg, probe, resolution, frames_per_batch, totals.ctypes.data_as(_dp),
                                                    sasa.ctypes.data_as(_dp) if sasa is not None else None, dp_, nd, err, 512)
    if ret:
        raise RuntimeError("freesasa_gpu_trajectory: " + err.value.decode())
    return totals, sasa


def trajectory_file(frames_path, radii, totals_path, sasa_path=None, done_path=None, f32=False, header_bytes=0,
                    n_frames=0, alg=LEE_RICHARDS, probe=1.4, resolution=20, frames_per_batch=0, max_new_shards=0, device=-1,
                    devices=None, out_f32=False):
    """freesasa_gpu_trajectory_file(): raw frame file -> totals file (+ per-atom file), resumable through the
    done-list at done_path.  Returns (complete, n_frames): complete is False when max_new_shards stopped the run.
    f32: the frames are floats (an input format); out_f32: the per-atom file holds floats (an output format)."""
    radii = _f64(radii)
    f32 = (1 if f32 else 0) | (2 if out_f32 else 0)
    err = C.create_string_buffer(512)
    total = C.c_longlong(0)
    enc = lambda p: None if p is None else str(p).encode()
    if devices is None:
        ret = lib().freesasa_gpu_trajectory_file(enc(frames_path), f32, header_bytes, radii.ctypes.data_as(_dp), radii.size,
                                                 n_frames, alg, probe, resolution, frames_per_batch, enc(totals_path), enc(sasa_path),
                                                 enc(done_path), max_new_shards, device, C.byref(total), err, 512)
    else:
        keep, dp_, nd = _devs(devices, device)
        ret = lib().freesasa_gpu_trajectory_file_devices(enc(frames_path), f32, header_bytes, radii.ctypes.data_as(_dp), radii.size,
                                                         n_frames, alg, probe, resolution, frames_per_batch, enc(totals_path), enc(sasa_path),
                                                         enc(done_path), max_new_shards, dp_, nd, C.byref(total), err, 512)
    if ret < 0:
        raise RuntimeError("freesasa_gpu_trajectory_file: " + err.value.decode())
    return ret == 0, int(total.value)


def parse_files_dev(paths, ingest_options=0, n_threads=0, device=0):
    """freesasa_gpu_parse_files(): the device-side PDB / mmCIF parser on its own -> (xyz [atoms, 3], radii, classes,
    offsets [n + 1], status [n], refused [n]); a refused file (the sweep hands it to the host parser) contributes no atoms."""
    L = lib()
    L.freesasa_gpu_parse_files.argtypes = [C.POINTER(C.c_char_p), C.c_int, C.c_int, C.c_int, C.c_int, _dp, _dp, C.POINTER(C.c_ubyte),
                                           C.c_longlong, _lp, _ip, _ip, C.c_char_p, C.c_int]
    L.freesasa_gpu_parse_files.restype = C.c_longlong
    n = len(paths)
    arr = (C.c_char_p * n)(*[str(p).encode() for p in paths])
    offs, status, host = np.zeros(n + 1, dtype=np.int64), np.zeros(n, dtype=np.int32), np.zeros(n, dtype=np.int32)
    err = C.create_string_buffer(512)
    cap = 1 << 16
    while True:
        xyz, r, cls = np.empty(3 * cap), np.empty(cap), np.empty(cap, dtype=np.uint8)
        got = L.freesasa_gpu_parse_files(arr, n, ingest_options, n_threads, device, xyz.ctypes.data_as(_dp), r.ctypes.data_as(_dp),
                                         cls.ctypes.data_as(C.POINTER(C.c_ubyte)), cap, offs.ctypes.data_as(_lp), status.ctypes.data_as(_ip),
                                         host.ctypes.data_as(_ip), err, 512)
        if got == -2:
            cap = int(offs[-1]) + 16
            continue
        if got < 0:
            raise RuntimeError("freesasa_gpu_parse_files: " + err.value.decode())
        return xyz[:3 * got].reshape(-1, 3).copy(), r[:got].copy(), cls[:got].copy(), offs, status, host


def sweep_parse_stats():
    """(files parsed on the device, files left to the host parser) by this process's sweeps since the last call."""
    a, b = C.c_longlong(0), C.c_longlong(0)
    lib().freesasa_gpu_sweep_parse_stats(C.byref(a), C.byref(b))
    return int(a.value), int(b.value)


def host_test_fail_after(n):
    """freesasa_host_test_fail_after(): the n-th host allocation / thread creation of the library's own code fails
    (n <= 0: off); returns what was left of the previous countdown (tests/test_hostfault.py)."""
    return int(lib().freesasa_host_test_fail_after(int(n)))


def test_points(n_points):
    tp = np.empty(3 * n_points)
    lib().freesasa_gpu_test_points(n_points, tp.ctypes.data_as(_dp))
    return tp.reshape(n_points, 3)


class GpuContext:
    """Device-resident batches: pointers are raw device addresses (e.g. tensor.data_ptr())."""

    def __init__(self, device=0, stream=None, timing=False):
        self._h = lib().freesasa_gpu_ctx_create(device, C.c_void_p(stream) if stream else None)
        if not self._h:
            raise RuntimeError("freesasa_gpu_ctx_create failed: no usable HIP device "
                               "(libfreesasa_amd has no CPU path)")
        if timing:
            lib().freesasa_gpu_ctx_set_timing(self._h, 1)
        self._tp = {}

    def set_timing(self, on):
        """freesasa_gpu_ctx_set_timing(): HIP events around the cell sort and the tile kernel of every batch (stats():
        ms_prep, ms_kernel, ms_total); off by default - the four event records cost a small batch ~35 us of its step."""
        lib().freesasa_gpu_ctx_set_timing(self._h, 1 if on else 0)

    def close(self):
        if self._h:
            lib().freesasa_gpu_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def error(self):
        return lib().freesasa_gpu_ctx_last_error(self._h).decode()

    def stats(self):
        s = Stats()
        lib().freesasa_gpu_ctx_get_stats(self._h, C.byref(s))
        return {k: getattr(s, k) for k, _ in Stats._fields_}

    def lee_richards(self, d_xyz, d_radii, offsets, d_sasa, d_totals=0, probe=1.4, n_slices=20):
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        ret = lib().freesasa_gpu_lr_batch_dev(self._h, d_xyz, d_radii, offsets.ctypes.data_as(_lp),
                                              offsets.size - 1, probe, n_slices, d_sasa,
                                              d_totals or None)
        if ret:
            raise RuntimeError("freesasa_gpu_lr_batch_dev: " + self.error())

    def lee_richards_async(self, d_xyz, d_radii, offsets, d_sasa, d_totals=0, probe=1.4, n_slices=20):
        """Enqueue the batch and return (freesasa_gpu_lr_batch_dev_async): up to two in flight; wait() collects them."""
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        if lib().freesasa_gpu_lr_batch_dev_async(self._h, d_xyz, d_radii, offsets.ctypes.data_as(_lp),
                                                 offsets.size - 1, probe, n_slices, d_sasa, d_totals or None):
            raise RuntimeError("freesasa_gpu_lr_batch_dev_async: " + self.error())

    def wait(self):
        if lib().freesasa_gpu_wait(self._h):
            raise RuntimeError("freesasa_gpu_wait: " + self.error())

    def lr_neighbors(self, d_xyz, d_radii, offsets, d_nn, d_nb=0, nb_cap=0, probe=1.4):
        """Test hook: the neighbor sets the L&R kernel finds (counts, optionally the first nb_cap neighbors per atom)."""
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        if lib().freesasa_gpu_lr_neighbors_dev(self._h, d_xyz, d_radii, offsets.ctypes.data_as(_lp), offsets.size - 1,
                                               probe, d_nn, d_nb or None, nb_cap):
            raise RuntimeError("freesasa_gpu_lr_neighbors_dev: " + self.error())

    def arc_union(self, sets):
        """Test hook: exposed arc length of every set of (start, end) arcs through the kernel's arc union."""
        first = np.concatenate([[0], np.cumsum([len(x) // 2 for x in sets])]).astype(np.int32)
        arcs = np.ascontiguousarray(np.concatenate([np.asarray(x, dtype=np.float64).ravel() for x in sets]))
        out = np.empty(len(sets))
        if lib().freesasa_gpu_arc_union_dev(self._h, arcs.ctypes.data_as(_dp), first.ctypes.data_as(_ip), len(sets),
                                            out.ctypes.data_as(_dp)):
            raise RuntimeError("freesasa_gpu_arc_union_dev: " + self.error())
        return out

    def segment_sums(self, d_sasa, seg_offsets, d_out):
        seg = np.ascontiguousarray(seg_offsets, dtype=np.int64)
        if lib().freesasa_gpu_segment_sums_dev(self._h, d_sasa, seg.ctypes.data_as(_lp), seg.size - 1, d_out):
            raise RuntimeError("freesasa_gpu_segment_sums_dev: " + self.error())

    def class_sums(self, d_sasa, d_class, offsets, d_out):
        """d_out[3*s + c] <- sum of d_sasa over structure s's atoms of class c (0 apolar, 1 polar, 2 unknown)."""
        offs = np.ascontiguousarray(offsets, dtype=np.int64)
        if lib().freesasa_gpu_class_sums_dev(self._h, d_sasa, d_class, offs.ctypes.data_as(_lp), offs.size - 1, d_out):
            raise RuntimeError("freesasa_gpu_class_sums_dev: " + self.error())

    def residue_areas(self, d_sasa, d_class, d_backbone, res_first, d_abs, res_ref=None, ref_table=None, d_rel=0):
        """Per-residue node areas (d_abs[6*r+..]) and, given res_ref / ref_table, relative areas (d_rel[5*r+..])."""
        rf = np.ascontiguousarray(res_first, dtype=np.int64)
        rr = np.ascontiguousarray(res_ref, dtype=np.int16) if res_ref is not None else None
        rt = np.ascontiguousarray(ref_table, dtype=np.float64).reshape(-1) if ref_table is not None else None
        ret = lib().freesasa_gpu_residue_areas_dev(
            self._h, d_sasa, d_class, d_backbone, rf.ctypes.data_as(_lp), rf.size - 1,
            rr.ctypes.data_as(C.POINTER(C.c_short)) if rr is not None else None,
            rt.ctypes.data_as(_dp) if rt is not None else None, (rt.size // 5) if rt is not None else 0, d_abs, d_rel)
        if ret:
            raise RuntimeError("freesasa_gpu_residue_areas_dev: " + self.error())

    def shrake_rupley(self, d_xyz, d_radii, offsets, d_sasa, d_counts=0, d_totals=0, probe=1.4,
                      n_points=100):
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        if n_points not in self._tp:
            self._tp[n_points] = np.ascontiguousarray(test_points(n_points).reshape(-1))
        tp = self._tp[n_points]
        ret = lib().freesasa_gpu_sr_batch_dev(self._h, d_xyz, d_radii, offsets.ctypes.data_as(_lp),
                                              offsets.size - 1, probe, n_points,
                                              tp.ctypes.data_as(_dp), d_sasa, d_counts or None,
                                              d_totals or None)
        if ret:
            raise RuntimeError("freesasa_gpu_sr_batch_dev: " + self.error())
