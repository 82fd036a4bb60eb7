"""Workload tooling: seeded synthetic structure generators (tools/synth.c) via ctypes."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(HERE, "libsasa_synth.so")
_dp = C.POINTER(C.c_double)
_lib = None


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            subprocess.run(["make", "-C", HERE], check=True, stdout=subprocess.DEVNULL)
        _lib = C.CDLL(_SO)
        _lib.synth_coil.argtypes = [C.c_int, C.c_uint64, _dp, _dp]
        _lib.synth_globule.argtypes = [C.c_int, C.c_uint64, C.c_double, _dp, _dp]
        _lib.synth_coil_batch.argtypes = [C.c_int, C.c_int, C.c_uint64, _dp, _dp]
        _lib.synth_jitter.argtypes = [_dp, C.c_int, C.c_uint64, C.c_double, _dp]
    return _lib


def coil(n, seed):
    """Self-avoiding random-walk coil: (xyz[n,3], radii[n])."""
    xyz, r = np.empty((n, 3)), np.empty(n)
    if _load().synth_coil(n, seed, xyz.ctypes.data_as(_dp), r.ctypes.data_as(_dp)):
        raise RuntimeError("synth_coil failed")
    return xyz, r


def globule(n, seed, spacing=2.6):
    """Jittered simple-cubic lattice (protein-like packing at spacing 2.6 A)."""
    xyz, r = np.empty((n, 3)), np.empty(n)
    if _load().synth_globule(n, seed, spacing, xyz.ctypes.data_as(_dp), r.ctypes.data_as(_dp)):
        raise RuntimeError("synth_globule failed")
    return xyz, r


def coil_batch(n_structs, n, seed0=1000):
    """n_structs coils of n atoms, seeds seed0+k, concatenated: (xyz, radii, offsets)."""
    xyz, r = np.empty((n_structs * n, 3)), np.empty(n_structs * n)
    if _load().synth_coil_batch(n_structs, n, seed0, xyz.ctypes.data_as(_dp), r.ctypes.data_as(_dp)):
        raise RuntimeError("synth_coil_batch failed")
    return xyz, r, np.arange(n_structs + 1, dtype=np.int64) * n


def jitter(base, seed, amp=0.5):
    base = np.ascontiguousarray(base, dtype=np.float64)
    out = np.empty_like(base)
    _load().synth_jitter(base.ctypes.data_as(_dp), base.size // 3, seed, amp, out.ctypes.data_as(_dp))
    return out
