#!/usr/bin/env python3
"""Mint the committed golden vectors from the REAL reference (run in the build container).

Inputs  : /root/reference/tests/data/*.pdb, parsed and given ProtOr radii by the reference's
          own parser/classifier (oracle/_ref/libfreesasa_ref.so: freesasa_structure_from_pdb).
Outputs : tests/golden/<name>.npz — xyz, radii and the reference's per-atom results
          (S&R-100, L&R-20, L&R-100, plus a parameter sweep for 1UBQ), produced by the
          reference's freesasa_calc_structure with n_threads = 1;
          tests/golden/synthetic.npz — seeded synthetic coil/globule cases and edge cases,
          results from the reference's freesasa_calc_coord;
          tests/golden/1ubq.B.pdb — the reference's own per-atom S&R golden file
          (tests/data/1ubq.B.pdb, a DATA file: coordinates, radius in the occupancy column,
          SASA in the B-factor column), copied verbatim.
Nothing here travels to the GPU box except the .npz/.pdb data it writes.
"""
import ctypes as C
import os
import shutil
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
import tools   # noqa: E402

REFDATA = "/root/reference/tests/data"
OUT = os.path.dirname(os.path.abspath(__file__))
_dp = C.POINTER(C.c_double)

ref = oracle.Reference()
lib = ref.lib
libc = C.CDLL(None)
libc.fopen.restype = C.c_void_p
libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
libc.fclose.argtypes = [C.c_void_p]
lib.freesasa_structure_from_pdb.restype = C.c_void_p
lib.freesasa_structure_from_pdb.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
lib.freesasa_structure_n.argtypes = [C.c_void_p]
lib.freesasa_structure_radius.restype = _dp
lib.freesasa_structure_radius.argtypes = [C.c_void_p]
lib.freesasa_structure_coord_array.restype = _dp
lib.freesasa_structure_coord_array.argtypes = [C.c_void_p]
lib.freesasa_structure_free.argtypes = [C.c_void_p]
lib.freesasa_calc_structure.restype = C.POINTER(oracle.Result)
lib.freesasa_calc_structure.argtypes = [C.c_void_p, C.POINTER(oracle.Parameters)]
lib.freesasa_set_verbosity(2)  # silent: unknown-atom warnings are not our business


def load_pdb(name, options=0):
    fp = libc.fopen(os.path.join(REFDATA, name).encode(), b"r")
    assert fp
    st = lib.freesasa_structure_from_pdb(fp, None, options)  # NULL = default (ProtOr) classifier
    libc.fclose(fp)
    assert st, name
    n = lib.freesasa_structure_n(st)
    xyz = np.ctypeslib.as_array(lib.freesasa_structure_coord_array(st), (3 * n,)).copy()
    radii = np.ctypeslib.as_array(lib.freesasa_structure_radius(st), (n,)).copy()
    return st, xyz.reshape(n, 3), radii


def calc_structure(st, n, alg, probe=1.4, n_points=100, n_slices=20):
    p = oracle.Parameters(alg, probe, n_points, n_slices, 1)
    res = lib.freesasa_calc_structure(st, C.byref(p))
    assert res
    sasa = np.ctypeslib.as_array(res.contents.sasa, (n,)).copy()
    total = res.contents.total
    lib.freesasa_result_free(res)
    return sasa, total


def counts_from_sasa(sasa, radii, probe, n_points):
    """Invert sasa = 4*pi*r^2*count/N (src/sasa_sr.c:337) — exact for integer counts."""
    r = radii + probe
    c = np.rint(sasa * n_points / (4.0 * np.pi * r * r)).astype(np.int32)
    back = (4.0 * np.pi * r * r * c) / n_points
    assert np.array_equal(back, sasa), "count inversion is not exact"
    return c


def mint_structure(tag, pdb, options=0, sweep=False):
    st, xyz, radii = load_pdb(pdb, options)
    n = radii.size
    out = {"xyz": xyz, "radii": radii}
    s, t = calc_structure(st, n, oracle.SHRAKE_RUPLEY)
    out["sr100"], out["sr100_total"] = s, t
    out["sr100_counts"] = counts_from_sasa(s, radii, 1.4, 100)
    for ns in (20, 100):
        s, t = calc_structure(st, n, oracle.LEE_RICHARDS, n_slices=ns)
        out[f"lr{ns}"], out[f"lr{ns}_total"] = s, t
    if sweep:
        for probe in (1.0, 2.0):
            s, _ = calc_structure(st, n, oracle.SHRAKE_RUPLEY, probe=probe)
            out[f"sr100_p{probe}_counts"] = counts_from_sasa(s, radii, probe, 100)
            out[f"lr20_p{probe}"], _ = calc_structure(st, n, oracle.LEE_RICHARDS, probe=probe)
        for npts in (20, 1000):
            s, _ = calc_structure(st, n, oracle.SHRAKE_RUPLEY, n_points=npts)
            out[f"sr{npts}_counts"] = counts_from_sasa(s, radii, 1.4, npts)
        out["lr5"], _ = calc_structure(st, n, oracle.LEE_RICHARDS, n_slices=5)
    lib.freesasa_structure_free(st)
    np.savez_compressed(os.path.join(OUT, tag + ".npz"), **out)
    print(f"{tag}: n={n} SR100={out['sr100_total']:.6f} LR20={out['lr20_total']:.6f}")


def mint_synthetic():
    out = {}

    def add(tag, xyz, radii, lr_slices=(20,), sr_points=(100,), probe=1.4):
        out[tag + "_xyz"], out[tag + "_radii"] = xyz, radii
        for ns in lr_slices:
            out[f"{tag}_lr{ns}"], _ = ref.calc_coord(xyz, radii, oracle.LEE_RICHARDS, probe, n_slices=ns)
        for npts in sr_points:
            s, _ = ref.calc_coord(xyz, radii, oracle.SHRAKE_RUPLEY, probe, n_points=npts)
            out[f"{tag}_sr{npts}_counts"] = counts_from_sasa(s, radii, probe, npts)

    add("coil1000", *tools.coil(1000, 1000), lr_slices=(20, 100))
    add("coil3000", *tools.coil(3000, 1001))
    add("globule1000", *tools.globule(1000, 7), lr_slices=(20, 100))
    add("dense512", *tools.globule(512, 11, spacing=2.05))           # avg nn ~ 80, max > 100
    # reference's analytic fixtures (tests/test_freesasa.c:59-78): two spheres r=1,2 at d=2
    add("two_x", np.array([[0, 0, 0], [2.0, 0, 0]]), np.array([1.0, 2.0]), (20, 20000), (100, 5000))
    add("two_y", np.array([[0, 0, 0], [0, 2.0, 0]]), np.array([1.0, 2.0]), (20, 20000), (100, 5000))
    add("two_z", np.array([[0, 0, 0], [0, 0, 2.0]]), np.array([1.0, 2.0]), (20, 20000), (100, 5000))
    # four spheres (tests/test_freesasa.c:66-78 style): mixed overlaps
    add("four", np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1.0]]),
        np.array([1.0, 1.0, 2.0, 1.0]), (20, 1000))
    # exactly touching spheres: d == R1+R2 -> strict '<' says not neighbors (src/nb.c:491)
    # (L&R only: the reference's S&R reads uninitialised memory when an atom has no neighbor)
    add("touch", np.array([[0, 0, 0], [0.0, 0, 6.0]]), np.array([1.5, 1.7]), (20,), ())
    # a fully buried atom inside a big one (both algorithms give exactly 0.0 for it)
    add("buried", np.array([[0, 0, 0], [0.1, 0.2, 0.3], [7.0, 0, 0]]), np.array([4.0, 1.0, 1.5]))
    # nb-list KAT geometry of tests/test_nb.c:7-8 (6 atoms on a line-ish arrangement)
    add("nbkat", np.array([[0, 0, 0], [1, 0, 0], [3, 0, 0], [4, 0, 0], [4.5, 0.2, 0.1], [20.0, 0, 0]]),
        np.array([1.0, 1.0, 1.0, 1.0, 1.0, 1.0]), (20,), ())   # last atom has no neighbors: L&R only
    np.savez_compressed(os.path.join(OUT, "synthetic.npz"), **out)
    print("synthetic:", len(out), "arrays")


if __name__ == "__main__":
    mint_structure("1ubq", "1ubq.pdb", sweep=True)
    mint_structure("1a0q", "1a0q.pdb")
    mint_structure("3bzd_trimmed", "3bzd_trimmed.pdb")
    mint_structure("1d3z", "1d3z.pdb")
    mint_structure("1d3z_H", "1d3z.pdb", options=1 << 2)  # FREESASA_INCLUDE_HYDROGEN
    mint_synthetic()
    shutil.copy(os.path.join(REFDATA, "1ubq.B.pdb"), os.path.join(OUT, "1ubq.B.pdb"))


# RSA fixtures: the reference CLI's own --format=rsa output, generated here by
#   oracle/_ref/freesasa_ref --format=rsa --shrake-rupley tests/golden/pdb/1ubq.pdb > tests/golden/1ubq.sr100.rsa
#   oracle/_ref/freesasa_ref --format=rsa                 tests/golden/pdb/1ubq.pdb > tests/golden/1ubq.lr20.rsa
#   oracle/_ref/freesasa_ref --format=rsa --shrake-rupley tests/golden/pdb/3bkr.pdb > tests/golden/3bkr.sr100.rsa
