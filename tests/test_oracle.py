"""CPU tests pinning the oracle (oracle/sasa_oracle.c) to the reference.

Three anchors:
  1. the reference's own golden numbers (tests/test_freesasa.c, tests/data/1ubq.B.pdb,
     src/sasa_lr.c:455-475) through committed fixtures;
  2. per-atom vectors minted from the real reference (tests/golden/make_golden.py);
  3. when oracle/_ref/libfreesasa_ref.so is present: live bit-for-bit comparison.
"""
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_golden, read_bfactor_pdb

TWOPI = 2 * np.pi
STRUCTS = ["1ubq", "1a0q", "3bzd_trimmed", "1d3z", "1d3z_H"]


# ---- reference golden totals (tests/test_freesasa.c:161,175,305,442,451) ----
REF_TOTALS = {
    ("1ubq", "sr100"): 4834.716265, ("1ubq", "lr20"): 4804.055641,
    ("3bzd_trimmed", "sr100"): 16133.867124,
    ("1d3z", "sr100"): 5000.340175, ("1d3z_H", "sr100"): 5035.614493,
}


@pytest.mark.parametrize("key", sorted(REF_TOTALS))
def test_reference_totals(oracle_lib, key):
    name, what = key
    g = load_golden(name)
    if what == "sr100":
        sasa, _ = oracle_lib.shrake_rupley(g["xyz"], g["radii"], 1.4, 100)
    else:
        sasa = oracle_lib.lee_richards(g["xyz"], g["radii"], 1.4, 20)
    assert abs(oracle_lib.total(sasa) - REF_TOTALS[key]) < 1e-5  # the reference's own tolerance


@pytest.mark.parametrize("name", STRUCTS)
def test_per_atom_bit_exact_vs_minted_vectors(oracle_lib, name):
    g = load_golden(name)
    sasa, counts = oracle_lib.shrake_rupley(g["xyz"], g["radii"], 1.4, 100)
    assert np.array_equal(counts, g["sr100_counts"])
    assert np.array_equal(sasa, g["sr100"])
    assert oracle_lib.total(sasa) == float(g["sr100_total"])
    for ns in (20, 100):
        lr = oracle_lib.lee_richards(g["xyz"], g["radii"], 1.4, ns)
        assert np.array_equal(lr, g[f"lr{ns}"])
        assert oracle_lib.total(lr) == float(g[f"lr{ns}_total"])


def test_parameter_sweep_1ubq(oracle_lib):
    g = load_golden("1ubq")
    for probe in (1.0, 2.0):
        _, c = oracle_lib.shrake_rupley(g["xyz"], g["radii"], probe, 100)
        assert np.array_equal(c, g[f"sr100_p{probe}_counts"])
        assert np.array_equal(oracle_lib.lee_richards(g["xyz"], g["radii"], probe, 20), g[f"lr20_p{probe}"])
    for npts in (20, 1000):
        _, c = oracle_lib.shrake_rupley(g["xyz"], g["radii"], 1.4, npts)
        assert np.array_equal(c, g[f"sr{npts}_counts"])
    assert np.array_equal(oracle_lib.lee_richards(g["xyz"], g["radii"], 1.4, 5), g["lr5"])


def test_1ubq_B_pdb_golden_file(oracle_lib):
    """tests/data/1ubq.B.pdb is the reference CLI's per-atom S&R output (test-cli.in:214-215)."""
    xyz, rad, sasa_ref = read_bfactor_pdb(os.path.join(GOLDEN, "1ubq.B.pdb"))
    assert len(rad) == 602
    sasa, _ = oracle_lib.shrake_rupley(xyz, rad, 1.4, 100)
    assert np.max(np.abs(sasa - sasa_ref)) <= 0.005 + 1e-9  # file has two decimals


def test_synthetic_and_edge_cases(oracle_lib):
    g = load_golden("synthetic")
    tags = sorted({k[:-4] for k in g.files if k.endswith("_xyz")})
    assert len(tags) >= 10
    for tag in tags:
        xyz, radii = g[tag + "_xyz"], g[tag + "_radii"]
        for k in g.files:
            if not k.startswith(tag + "_") or k.endswith(("_xyz", "_radii")):
                continue
            what = k[len(tag) + 1:]
            if what.startswith("lr"):
                assert np.array_equal(oracle_lib.lee_richards(xyz, radii, 1.4, int(what[2:])), g[k]), k
            else:
                _, c = oracle_lib.shrake_rupley(xyz, radii, 1.4, int(what[2:].split("_")[0]))
                assert np.array_equal(c, g[k]), k


def test_analytic_two_spheres(oracle_lib):
    """Closed-form two-sphere area, the reference's own check and tolerance
    (tests/test_freesasa.c:22-43, 59-101: rel_err = |a-b|/(|a|+|b|) on the TOTAL)."""
    r1, r2, d, probe = 1.0 + 1.4, 2.0 + 1.4, 2.0, 1.4
    hidden = np.pi / d * (r1 * (r2 * r2 - (d - r1) ** 2) + r2 * (r1 * r1 - (d - r2) ** 2))
    exact = 4 * np.pi * (r1 * r1 + r2 * r2) - hidden

    def rel_err(a, b):
        return abs(a - b) / (abs(a) + abs(b))
    for axis in range(3):
        xyz = np.zeros((2, 3))
        xyz[1, axis] = d
        lr = oracle_lib.lee_richards(xyz, [1.0, 2.0], probe, 20000)
        assert rel_err(oracle_lib.total(lr), exact) < 1e-5
        sr, _ = oracle_lib.shrake_rupley(xyz, [1.0, 2.0], probe, 5000)
        assert rel_err(oracle_lib.total(sr), exact) < 1e-3


def test_exposed_arc_length_known_answers(oracle_lib):
    """The nine hand cases of src/sasa_lr.c:455-475."""
    T = TWOPI
    cases = [
        ([0, 0.1 * T, 0.9 * T, T], 0.8 * T), ([0.9 * T, T, 0, 0.1 * T], 0.8 * T),
        ([0, T, 1, 2], 0.0), ([1, 2, 0, T], 0.0),
        ([0.1 * T, 0.2 * T, 0.5 * T, 0.6 * T], 0.8 * T),
        ([0.1 * T, 0.3 * T, 0.15 * T, 0.2 * T], 0.8 * T),
        ([0.15 * T, 0.2 * T, 0.1 * T, 0.3 * T], 0.8 * T),
        ([0.05, 0.1, 0.5, 0.6, 0, 0.15, 0.7, 0.8, 0.75, T], 0.45),
    ]
    for arcs, want in cases:
        assert abs(oracle_lib.exposed_arc_length(arcs) - want) < 1e-10
    assert oracle_lib.exposed_arc_length([]) == TWOPI


def test_single_atom_and_isolated(oracle_lib):
    """Single sphere (tests/test_freesasa.c:138-153): area 4*pi*R^2; S&R with no neighbor is
    DEFINED here (all points exposed) where the reference reads uninitialised memory."""
    R = 1.0 + 1.4
    lr = oracle_lib.lee_richards([[1.0, 2.0, 3.0]], [1.0], 1.4, 20)
    assert abs(lr[0] - 4 * np.pi * R * R) < 1e-10
    sr, c = oracle_lib.shrake_rupley([[1.0, 2.0, 3.0]], [1.0], 1.4, 100)
    assert c[0] == 100 and sr[0] == (4.0 * np.pi * R * R * 100) / 100


def test_neighbor_list_kat(oracle_lib):
    """tests/test_nb.c:7-27 contact pattern on its six-atom geometry."""
    xyz = np.array([[0, 0, 0], [1, 0, 0], [3, 0, 0], [4, 0, 0], [4.5, 0.2, 0.1], [20.0, 0, 0]])
    start, idx = oracle_lib.neighbors(xyz, np.ones(6))
    nb = [set(idx[start[i]:start[i + 1]]) for i in range(6)]
    assert nb[0] == {1} and nb[1] == {0} and nb[2] == {3, 4} and nb[5] == set()
    for i in range(6):          # symmetric, irreflexive, unique
        assert i not in nb[i] and len(nb[i]) == start[i + 1] - start[i]
        for j in nb[i]:
            assert i in nb[j]


def test_neighbors_match_bruteforce(oracle_lib):
    import tools
    xyz, r = tools.globule(700, 3)
    r_ext = r + 1.4
    start, idx = oracle_lib.neighbors(xyz, r_ext)
    d = xyz[None, :, :] - xyz[:, None, :]
    d2 = d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1] + d[..., 2] * d[..., 2]
    cut = (r_ext[:, None] + r_ext[None, :]) ** 2
    want = (d2 < cut) & ~np.eye(len(r), dtype=bool)
    for i in range(len(r)):
        assert set(idx[start[i]:start[i + 1]]) == set(np.nonzero(want[i])[0])


# ---- live against the real reference, when it is built (this container) ----
def test_live_bit_exact_vs_reference(oracle_lib, reference_lib):
    import oracle
    import tools
    cases = [tools.coil(2000, 5), tools.globule(1500, 9), tools.globule(400, 2, 2.05)]
    g = load_golden("1a0q")
    cases.append((g["xyz"], g["radii"]))
    for xyz, r in cases:
        for probe in (1.4, 0.7):
            a, _ = reference_lib.calc_coord(xyz, r, oracle.LEE_RICHARDS, probe, n_slices=20)
            assert np.array_equal(a, oracle_lib.lee_richards(xyz, r, probe, 20))
            a, _ = reference_lib.calc_coord(xyz, r, oracle.SHRAKE_RUPLEY, probe, n_points=100)
            assert np.array_equal(a, oracle_lib.shrake_rupley(xyz, r, probe, 100)[0])


def test_reference_thread_count_is_result_neutral(reference_lib):
    """tests/test_freesasa.c:404-429: threads do not change results (so the GPU path may
    ignore n_threads)."""
    import oracle
    g = load_golden("1ubq")
    for alg in (oracle.LEE_RICHARDS, oracle.SHRAKE_RUPLEY):
        a1, t1 = reference_lib.calc_coord(g["xyz"], g["radii"], alg, n_threads=1)
        a4, t4 = reference_lib.calc_coord(g["xyz"], g["radii"], alg, n_threads=4)
        assert np.array_equal(a1, a4) and t1 == t4
