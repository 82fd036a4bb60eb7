"""ctypes loaders for the CPU oracle.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``
may import this package.  Nothing under ``freesasa_amd/`` does.

* ``Oracle``  — our plain-C restatement (``oracle/libsasa_oracle.so``, built from
  ``oracle/sasa_oracle.c``).
* ``Reference`` — the real reference compiled from ``/root/reference/src`` into the
  git-ignored ``oracle/_ref/libfreesasa_ref.so`` (see ``oracle/Makefile``); present in the
  build container and shipped to the GPU box as a built artefact, never as source.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "libsasa_oracle.so")
REF_SO = os.path.join(HERE, "_ref", "libfreesasa_ref.so")
REF_CLI = os.path.join(HERE, "_ref", "freesasa_ref")

LEE_RICHARDS, SHRAKE_RUPLEY = 0, 1
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def build(quiet=True):
    """Compile the oracle (and the reference, when /root/reference exists)."""
    subprocess.run(["make", "-C", HERE, "all"], check=True,
                   stdout=subprocess.DEVNULL if quiet else None)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class Parameters(C.Structure):
    """Layout of the reference's freesasa_parameters (src/freesasa.h:232-238)."""
    _fields_ = [("alg", C.c_int), ("probe_radius", C.c_double),
                ("shrake_rupley_n_points", C.c_int), ("lee_richards_n_slices", C.c_int),
                ("n_threads", C.c_int)]


class Result(C.Structure):
    """Layout of the reference's freesasa_result (src/freesasa.h:267-272)."""
    _fields_ = [("total", C.c_double), ("sasa", _dp), ("n_atoms", C.c_int),
                ("parameters", Parameters)]


class Oracle:
    def __init__(self, path=ORACLE_SO):
        if not os.path.exists(path):
            build()
        self.lib = lib = C.CDLL(path)
        lib.oracle_test_points.argtypes = [C.c_int, _dp]
        lib.oracle_test_points.restype = None
        lib.oracle_exposed_arc_length.argtypes = [_dp, C.c_int]
        lib.oracle_exposed_arc_length.restype = C.c_double
        lib.oracle_neighbors.argtypes = [_dp, _dp, C.c_int, C.POINTER(_ip), C.POINTER(_ip)]
        lib.oracle_shrake_rupley.argtypes = [_dp, _dp, C.c_int, C.c_double, C.c_int, _dp, _ip]
        lib.oracle_lee_richards.argtypes = [_dp, _dp, C.c_int, C.c_double, C.c_int, _dp]
        lib.oracle_lr_work_stats.argtypes = [_dp, _dp, C.c_int, C.c_double, C.c_int, _dp]
        lib.oracle_total.argtypes = [_dp, C.c_int]
        lib.oracle_total.restype = C.c_double
        self.libc = C.CDLL(None)
        self.libc.free.argtypes = [C.c_void_p]

    def test_points(self, n):
        tp = np.empty(3 * n)
        self.lib.oracle_test_points(n, tp.ctypes.data_as(_dp))
        return tp.reshape(n, 3)

    def exposed_arc_length(self, arcs):
        a = _f64(arcs).copy().ravel()
        return self.lib.oracle_exposed_arc_length(a.ctypes.data_as(_dp), a.size // 2)

    def neighbors(self, xyz, r_ext):
        xyz, r_ext = _f64(xyz).ravel(), _f64(r_ext)
        n = r_ext.size
        ps, pi = _ip(), _ip()
        if self.lib.oracle_neighbors(xyz.ctypes.data_as(_dp), r_ext.ctypes.data_as(_dp), n,
                                     C.byref(ps), C.byref(pi)):
            raise RuntimeError("oracle_neighbors failed")
        start = np.ctypeslib.as_array(ps, (n + 1,)).copy()
        idx = np.ctypeslib.as_array(pi, (max(int(start[-1]), 1),)).copy()[:start[-1]]
        self.libc.free(ps)
        self.libc.free(pi)
        return start, idx

    def shrake_rupley(self, xyz, radii, probe=1.4, n_points=100):
        xyz, radii = _f64(xyz).ravel(), _f64(radii)
        n = radii.size
        sasa, counts = np.empty(n), np.empty(n, dtype=np.int32)
        if self.lib.oracle_shrake_rupley(xyz.ctypes.data_as(_dp), radii.ctypes.data_as(_dp), n,
                                         probe, n_points, sasa.ctypes.data_as(_dp),
                                         counts.ctypes.data_as(_ip)):
            raise RuntimeError("oracle_shrake_rupley failed")
        return sasa, counts

    def lee_richards(self, xyz, radii, probe=1.4, n_slices=20):
        xyz, radii = _f64(xyz).ravel(), _f64(radii)
        n = radii.size
        sasa = np.empty(n)
        if self.lib.oracle_lee_richards(xyz.ctypes.data_as(_dp), radii.ctypes.data_as(_dp), n,
                                        probe, n_slices, sasa.ctypes.data_as(_dp)):
            raise RuntimeError("oracle_lee_richards failed")
        return sasa

    def lr_work_stats(self, xyz, radii, probe=1.4, n_slices=20):
        xyz, radii = _f64(xyz).ravel(), _f64(radii)
        st = np.zeros(6)
        self.lib.oracle_lr_work_stats(xyz.ctypes.data_as(_dp), radii.ctypes.data_as(_dp),
                                      radii.size, probe, n_slices, st.ctypes.data_as(_dp))
        return dict(zip(("tests", "zpass", "arcs", "buried", "max_arcs", "nn_sum"), st))

    def total(self, sasa):
        s = _f64(sasa)
        return self.lib.oracle_total(s.ctypes.data_as(_dp), s.size)


class Reference:
    """The real reference library (freesasa_calc_coord et al.) through ctypes."""

    def __init__(self, path=REF_SO):
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        self.lib = lib = C.CDLL(path)
        lib.freesasa_calc_coord.argtypes = [_dp, _dp, C.c_int, C.POINTER(Parameters)]
        lib.freesasa_calc_coord.restype = C.POINTER(Result)
        lib.freesasa_result_free.argtypes = [C.POINTER(Result)]
        lib.freesasa_result_free.restype = None
        lib.freesasa_set_verbosity.argtypes = [C.c_int]

    @staticmethod
    def available():
        return os.path.exists(REF_SO)

    def calc_coord(self, xyz, radii, alg=LEE_RICHARDS, probe=1.4, n_points=100, n_slices=20,
                   n_threads=1):
        xyz, radii = _f64(xyz).ravel(), _f64(radii)
        p = Parameters(alg, probe, n_points, n_slices, n_threads)
        res = self.lib.freesasa_calc_coord(xyz.ctypes.data_as(_dp), radii.ctypes.data_as(_dp),
                                           radii.size, C.byref(p))
        if not res:
            raise RuntimeError("reference freesasa_calc_coord returned NULL")
        sasa = np.ctypeslib.as_array(res.contents.sasa, (radii.size,)).copy()
        total = res.contents.total
        self.lib.freesasa_result_free(res)
        return sasa, total
