"""TESTS ONLY: ctypes front-end of the CPU emulation of the kernel phase functions."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
_SO = os.environ.get("SASA_EMU_SO") or os.path.join(HERE, "libsasa_emu.so")  # (SASA_EMU_SO: a variant build, see test_emulation.py)
_dp, _ip, _lp = C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_int64)
_lib = None


def _load():
    global _lib
    if _lib is None:
        if not os.environ.get("SASA_EMU_SO"):
            subprocess.run(["make", "-C", ROOT, "emu"], check=True, stdout=subprocess.DEVNULL)
        _lib = C.CDLL(_SO)
        _lib.emu_run_batch.argtypes = [C.c_int, _dp, _dp, _lp, C.c_int, C.c_double, C.c_int, _dp,
                                       _dp, _ip, _dp, C.POINTER(C.c_longlong)] + [C.c_int] * 9
        _lib.emu_set_lr2_opts.argtypes = [C.c_int]
    return _lib


def set_lr2_opts(shape_builds=False, compact_cells=False):
    """Run the Lee-Richards tile kernel's shape-specialised builds where the device would, and / or look cells up in the
    compact cell table (single-structure batches)."""
    _load().emu_set_lr2_opts((1 if shape_builds else 0) | (2 if compact_cells else 0))


def run_batch(lr, xyz, radii, offsets=None, probe=1.4, resolution=20, unit_pts=None,
              cap_idx=0, pool=0, ds=-1, fb_cap_idx=0, fb_pool=0, fb_ds=0, mid_cap_idx=0, mid_pool=0,
              mid_ds=-1, check=True):
    xyz = np.ascontiguousarray(xyz, dtype=np.float64).reshape(-1)
    radii = np.ascontiguousarray(radii, dtype=np.float64)
    n = radii.size
    if offsets is None:
        offsets = [0, n]
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    sasa = np.full(n, np.nan)
    counts = np.full(n, -1, dtype=np.int32)
    totals = np.zeros(offsets.size - 1)
    stats = (C.c_longlong * 10)()
    up = None
    if unit_pts is not None:
        up = np.ascontiguousarray(unit_pts, dtype=np.float64).reshape(-1)
    ret = _load().emu_run_batch(1 if lr else 0, xyz.ctypes.data_as(_dp), radii.ctypes.data_as(_dp),
                                offsets.ctypes.data_as(_lp), offsets.size - 1, probe, resolution,
                                up.ctypes.data_as(_dp) if up is not None else None,
                                sasa.ctypes.data_as(_dp), counts.ctypes.data_as(_ip),
                                totals.ctypes.data_as(_dp), stats, cap_idx, pool, ds,
                                fb_cap_idx, fb_pool, fb_ds, mid_cap_idx, mid_pool, mid_ds)
    st = dict(zip(("error", "fallback_tiles", "max_nn", "TA", "B", "lds", "cells", "items", "slab_tiles", "unused"), list(stats)))
    if check and ret:
        raise RuntimeError(f"emulated batch failed: {st}")
    return sasa, counts, totals, st
