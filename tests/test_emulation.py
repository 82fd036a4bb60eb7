"""CPU tests of the KERNEL LOGIC: the phase functions of freesasa_amd/csrc/sasa_kernels.h are
driven thread-by-thread by tests/emu (never shipped) in the production launch sequence and
compared with the oracle.  S&R must be bit-identical.  L&R shares the reference's arc-union
result bit for bit given equal arc end points, but computes cos(alpha) with reciprocals and
acos with its own polynomial.  Since round 3 the arc pass spends two orders of magnitude of the 1e-4 A^2
contract on speed (degree-12 acos, 6e-13 relative; 1/(2 Ri') to 4e-15; slice planes in closed form, where the
reference's accumulated z drifts by ~ns ulp(z)): held to LR_TOL = 1e-9 A^2 per atom here (observed 4.5e-11 at worst
on these inputs, 3e-12 rms; 4e-10 with coordinates near 1e4 A); on this box only the device's v_rsq_f64 seed remains untested."""
import numpy as np
import pytest

import tools
from conftest import load_golden
import emu
from emu import run_batch

LR_TOL = 1e-9


def close(a, b, tol=LR_TOL):
    return np.max(np.abs(np.asarray(a) - np.asarray(b))) <= tol


def _sr(oracle_lib, xyz, r, n_points=100, probe=1.4, **kw):
    return run_batch(False, xyz, r, probe=probe, resolution=n_points,
                     unit_pts=oracle_lib.test_points(n_points), **kw)


@pytest.mark.parametrize("name", ["1ubq", "3bzd_trimmed"])
def test_golden_structures(oracle_lib, name):
    g = load_golden(name)
    sasa, _, tot, st = run_batch(True, g["xyz"], g["radii"], resolution=20)
    assert close(sasa, g["lr20"]) and abs(tot[0] - float(g["lr20_total"])) < 1e-9
    # the kernel variant without bucketed beta ranking (sparse inputs) must agree bit for bit
    emu._load().emu_set_bucket(0)
    try:
        plain, _, _, _ = run_batch(True, g["xyz"], g["radii"], resolution=20)
    finally:
        emu._load().emu_set_bucket(1)
    assert np.array_equal(plain, sasa)
    sasa, counts, tot, _ = _sr(oracle_lib, g["xyz"], g["radii"])
    assert np.array_equal(counts, g["sr100_counts"])
    assert np.array_equal(sasa, g["sr100"]) and abs(tot[0] - float(g["sr100_total"])) < 1e-9


def test_lr100_and_probe_sweep(oracle_lib):
    g = load_golden("1ubq")
    sasa, *_ = run_batch(True, g["xyz"], g["radii"], resolution=100)
    assert close(sasa, g["lr100"])
    for probe in (1.0, 2.0):
        sasa, *_ = run_batch(True, g["xyz"], g["radii"], probe=probe, resolution=20)
        assert close(sasa, g[f"lr20_p{probe}"])
        _, c, _, _ = _sr(oracle_lib, g["xyz"], g["radii"], probe=probe)
        assert np.array_equal(c, g[f"sr100_p{probe}_counts"])
    for npts in (20, 1000):
        _, c, _, _ = _sr(oracle_lib, g["xyz"], g["radii"], n_points=npts)
        assert np.array_equal(c, g[f"sr{npts}_counts"])


def test_ragged_batch_with_empty_and_single_atom_structures(oracle_lib):
    parts = [tools.coil(700, 1), tools.globule(333, 2), (np.zeros((0, 3)), np.zeros(0)),
             (np.array([[5.0, 5.0, 5.0]]), np.array([1.7])), tools.coil(64, 3),
             tools.globule(250, 4, 2.05), tools.coil(9000, 5)]   # the last one spans 3 bounds chunks
    xyz = np.concatenate([p[0] for p in parts])
    # translate every structure to the same place: neighbors must never cross structures
    r = np.concatenate([p[1] for p in parts])
    offsets = np.concatenate([[0], np.cumsum([len(p[1]) for p in parts])])
    lr, _, tot, st = run_batch(True, xyz, r, offsets)
    sr, cnt, stot, _ = _sr(oracle_lib, xyz, r, offsets=offsets)
    for k, (px, pr) in enumerate(parts):
        sl = slice(offsets[k], offsets[k + 1])
        if len(pr) == 0:
            assert tot[k] == 0.0
            continue
        want = oracle_lib.lee_richards(px, pr)
        assert close(lr[sl], want) and abs(tot[k] - oracle_lib.total(want)) < 1e-9
        ws, wc = oracle_lib.shrake_rupley(px, pr)
        assert np.array_equal(cnt[sl], wc) and np.array_equal(sr[sl], ws)
    # isolated atom: L&R = slices of a free sphere, S&R = all points (defined; reference is UB)
    assert cnt[offsets[3]] == 100


def test_many_slices_strided_mode(oracle_lib):
    """resolution > 640 switches to one atom per workgroup with strided partial sums and
    z = z0 + (k+1)*delta: no longer bit-identical, must stay at rounding level."""
    g = load_golden("synthetic")
    for tag in ("two_x", "two_y", "two_z"):
        sasa, *_ = run_batch(True, g[tag + "_xyz"], g[tag + "_radii"], resolution=20000)
        assert np.max(np.abs(sasa - g[tag + "_lr20000"])) < 1e-9
        _, c, _, _ = _sr(oracle_lib, g[tag + "_xyz"], g[tag + "_radii"], n_points=5000)
        assert np.array_equal(c, g[tag + "_sr5000_counts"])


def test_edge_fixtures(oracle_lib):
    g = load_golden("synthetic")
    for tag in ("touch", "buried", "nbkat", "four"):
        sasa, *_ = run_batch(True, g[tag + "_xyz"], g[tag + "_radii"])
        assert close(sasa, g[tag + "_lr20"]), tag
    sasa, *_ = run_batch(True, g["buried_xyz"], g["buried_radii"])
    assert sasa[1] == 0.0


def test_coincident_atoms_are_nan_like_the_reference(oracle_lib):
    """Duplicate atom records (same centre, same radius): the reference divides 0 by 0 in every slice and returns
    NaN for such atoms (src/sasa_lr.c:335 with dij = 0 and equal slice radii).  The kernel's record for such a
    pair is NaN and cuts no arc, and the atom's area comes back NaN as well - a duplicate record does not pass for
    an area; everything else in the structure is unaffected.  S&R is defined for both."""
    xyz = np.array([[0.0, 0, 0], [0, 0, 0], [0, 0, 0], [9.0, 0, 0], [9.0, 2.5, 0]])
    r = np.array([1.8, 1.8, 1.8, 1.6, 1.7])
    want = oracle_lib.lee_richards(xyz, r)
    assert np.all(np.isnan(want[:3])) and np.all(np.isfinite(want[3:]))
    sasa, *_ = run_batch(True, xyz, r)
    assert np.all(np.isnan(sasa[:3]))
    assert close(sasa[3:], want[3:])
    _, c, _, _ = _sr(oracle_lib, xyz, r)
    assert np.array_equal(c, oracle_lib.shrake_rupley(xyz, r)[1])


def test_overflow_goes_to_later_launches_and_matches(oracle_lib):
    """Shrunken capacities: most tiles overflow (neighbor list, pool, arc stack) in the first
    launch and are redone by the second (larger LDS lists) and, when that is shrunk too, by the
    third (slab-backed) launch; results must not change."""
    xyz, r = tools.globule(900, 5)
    want = oracle_lib.lee_richards(xyz, r)
    for kw in (dict(cap_idx=24), dict(pool=100), dict(ds=0), dict(ds=1)):
        sasa, _, _, st = run_batch(True, xyz, r, **kw)
        assert st["fallback_tiles"] > 0 and st["slab_tiles"] == 0, kw
        assert close(sasa, want), kw
    for kw in (dict(cap_idx=24, mid_cap_idx=30), dict(pool=100, mid_pool=120), dict(ds=0, mid_ds=0)):
        sasa, _, _, st = run_batch(True, xyz, r, **kw)
        assert st["fallback_tiles"] > 0 and st["slab_tiles"] > 0, kw
        assert close(sasa, want), kw
    ws, wc = oracle_lib.shrake_rupley(xyz, r)
    _, c, _, st = _sr(oracle_lib, xyz, r, cap_idx=24, mid_cap_idx=30)
    assert st["slab_tiles"] > 0 and np.array_equal(c, wc)


def test_last_launch_capacity_errors_are_reported(oracle_lib):
    xyz, r = tools.globule(300, 6)
    _, _, _, st = run_batch(True, xyz, r, cap_idx=8, mid_cap_idx=12, fb_cap_idx=16, check=False)
    assert st["error"] == 4          # ERR_NEIGHBOR_CAP
    xyz, r = tools.coil(400, 8)
    _, _, _, st = run_batch(True, xyz, r, ds=0, mid_ds=0, fb_ds=1, check=False)
    assert st["error"] in (0, 5)     # ERR_STACK_CAP only if some slice needs depth > 2


def test_bad_inputs_flag_errors():
    xyz, r = tools.coil(50, 1)
    bad = xyz.copy()
    bad[7, 1] = np.nan
    assert run_batch(True, bad, r, check=False)[3]["error"] == 3    # ERR_BAD_COORD
    assert run_batch(True, xyz, -2.0 * np.ones(50), check=False)[3]["error"] == 1  # ERR_BAD_RADIUS
    far = xyz.copy()
    far[0, 0] = 1e12
    assert run_batch(True, far, r, check=False)[3]["error"] == 2    # ERR_GRID_TOO_BIG


def test_launch_configurations():
    """Every branch of the launch-configuration chooser on one small structure."""
    import oracle
    o = oracle.Oracle()
    xyz, r = tools.globule(150, 9)
    for ns in (1, 3, 7, 20, 33, 64, 100, 333, 640, 641):
        sasa, _, _, st = run_batch(True, xyz, r, resolution=ns)
        want = o.lee_richards(xyz, r, 1.4, ns)
        assert close(sasa, want, 1e-10), (ns, st)
    for npts in (1, 13, 100, 257, 4096, 4097):
        _, c, _, st = run_batch(False, xyz, r, resolution=npts, unit_pts=o.test_points(npts))
        assert np.array_equal(c, o.shrake_rupley(xyz, r, 1.4, npts)[1]), (npts, st)
    # beyond 8192 points the tile runs without its survivor table (sr_survivors_fit): any point count works
    xs, rs = xyz[:3 * 40], r[:40]
    for npts in (8192, 8193, 45000):
        _, c, _, st = run_batch(False, xs, rs, resolution=npts, unit_pts=o.test_points(npts))
        assert np.array_equal(c, o.shrake_rupley(xs, rs, 1.4, npts)[1]), (npts, st)


def test_lr_launch_shape_follows_density_and_demand():
    """Host side of the L&R launch: atoms per tile from slices and density, the pool from the demand histogram of
    the last batch weighed against the occupancy step it costs (lr2_choose_cfg, lr2_pool_from_hist)."""
    import ctypes as C
    lib = emu._load()
    lib.emu_lr2_shape.argtypes = [C.c_int, C.c_double, C.c_int, C.c_int, C.c_double, C.POINTER(C.c_int)]
    lib.emu_lr2_pool_from_hist.argtypes = [C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.c_int]
    lib.emu_lr2_lds.argtypes = [C.c_int] * 5

    def shape(ns, nn=0.0, nn_max=0, last_ta=0, last_split=0.0):
        out = (C.c_int * 6)()
        lib.emu_lr2_shape(ns, nn, nn_max, last_ta, last_split, out)
        return dict(zip(("TA", "pool", "mw", "ds", "lds", "rmax"), out))

    def pool(hist, TA=6, ns=20, mw=2, ds=2):
        return lib.emu_lr2_pool_from_hist((C.c_int * 64)(*hist), TA, ns, mw, ds)

    coil, dense = shape(20, 27.0, 40), shape(20, 75.0, 110)
    assert coil["TA"] == 6 and coil["mw"] == 2 and coil["lds"] * 16 <= 160 * 1024 < (coil["lds"] + 96) * 16
    assert dense["TA"] < coil["TA"] and dense["mw"] >= 3 and dense["pool"] <= 256
    # atoms per tile: as many as keep ~320 items AND 16 tiles per CU with a pool for ~90 % of the tiles
    assert [shape(ns, 30.0, 45)["TA"] for ns in (20, 40, 60, 100, 200, 256)] == [6, 5, 4, 3, 1, 1]
    for s in (coil, dense, shape(40, 30.0, 45), shape(60, 30.0, 45), shape(100, 30.0, 45), shape(100), shape(1)):
        assert s["pool"] <= 64 * s["rmax"] and s["lds"] * 16 <= 160 * 1024
    # with history: stay, one atom less when too many tiles were split, one more only when it clearly fits
    assert shape(20, 33.7, 77, last_ta=6, last_split=0.03)["TA"] == 6       # coils: 3 % split, halves of 60 items
    assert shape(20, 58.0, 74, last_ta=4, last_split=0.12)["TA"] == 3       # globules at TA 4: 12 % split
    assert shape(20, 62.0, 74, last_ta=3, last_split=0.005)["TA"] == 3      # ... and TA 3 stays
    assert shape(100, 33.7, 77, last_ta=3, last_split=0.075)["TA"] == 3     # 100 slices: halves fill the wave
    assert shape(20, 12.0, 30, last_ta=4, last_split=0.0)["TA"] == 5        # sparse: one more fits
    # protein density at 20 slices: a fourth atom only where the pool has a tenth to spare (round 5, measured: the
    # reference's PDB entries and the bench's globules both run faster with 3 than with the 4 that just fit)
    assert shape(20, 58.0, 74, last_ta=4, last_split=0.014)["TA"] == 3 and shape(20, 56.0, 74, last_ta=3, last_split=0.0)["TA"] == 3
    assert shape(20, 47.0, 70, last_ta=3, last_split=0.0)["TA"] == 4 and shape(20, 47.0, 70, last_ta=4, last_split=0.01)["TA"] == 4
    very_dense = shape(20, 400.0, 500)     # one atom per tile and still more records than 16 tiles per CU leave room for
    assert very_dense["TA"] == 1 and very_dense["pool"] == 256 and very_dense["lds"] * 16 > 160 * 1024

    width = 2 * 6 + 2  # hist_bin_width(6)
    p16 = max(p for p in range(16, 258, 2) if lib.emu_lr2_lds(6, 20, p, 2, 2) * 16 <= 160 * 1024)
    assert pool([0] * 64) == 0
    low = [0] * 64; low[5] = 1000                       # every tile needs ~75 records: the smallest step that holds them all
    assert pool(low) >= 6 * width and lib.emu_lr2_lds(6, 20, pool(low), 2, 2) * 16 <= 160 * 1024
    few_above = [0] * 64; few_above[10] = 980; few_above[p16 // width + 1] = 20   # 2 % above the 16-tile pool: stay
    assert pool(few_above) == p16
    most_above = [0] * 64; most_above[p16 // width + 1] = 1000                   # nearly all above it: give up a tile of occupancy
    assert pool(most_above) > p16


def test_device_math_helpers_accuracy():
    """acos_fast / sqrt_rh (the only transcendental code of the L&R arc pass besides atan2) against
    correctly rounded references everywhere on (-1,1), including next to +-1 and +-0.5: acos within
    2.5e-14 relative (its degree-9 polynomial) + 4e-16 absolute (pi/2 carried in one word)."""
    import ctypes as C
    import emu
    lib = emu._load()
    dp = C.POINTER(C.c_double)
    lib.emu_acos_fast.argtypes = [dp, dp, C.c_int]
    lib.emu_sqrt_rh.argtypes = [dp, dp, dp, C.c_int]
    rng = np.random.default_rng(3)
    edge = np.concatenate([1 - np.logspace(-16, -1, 400), -(1 - np.logspace(-16, -1, 400)),
                           0.5 + np.linspace(-1e-9, 1e-9, 101), -0.5 + np.linspace(-1e-9, 1e-9, 101),
                           np.linspace(-1e-9, 1e-9, 101)])
    x = np.concatenate([rng.uniform(-1, 1, 200_000), edge])
    x = x[(x > -1) & (x < 1)]
    out = np.empty_like(x)
    lib.emu_acos_fast(x.ctypes.data_as(dp), out.ctypes.data_as(dp), x.size)
    want = np.arccos(x)
    assert np.all(np.abs(out - want) <= 2.5e-14 * want + 4e-16)
    lib.emu_atan2_fast.argtypes = [dp, dp, dp, C.c_int]
    ang = np.concatenate([rng.uniform(-np.pi, np.pi, 200_000), np.pi / 8 * np.arange(-8, 9) + 1e-12,
                          np.pi / 8 * np.arange(-8, 9) - 1e-12, np.array([0.0, 1e-300, -1e-300])])
    rad = rng.uniform(1e-3, 30.0, ang.size)
    yy, xx = rad * np.sin(ang), rad * np.cos(ang)
    got = np.empty_like(yy)
    lib.emu_atan2_fast(yy.ctypes.data_as(dp), xx.ctypes.data_as(dp), got.ctypes.data_as(dp), yy.size)
    ref = np.arctan2(yy, xx)
    assert np.max(np.abs(got - ref)) <= 3 * np.spacing(np.pi)        # a few ulp of pi, absolute
    small = np.abs(ref) < 1e-3
    assert np.all(np.abs(got[small] - ref[small]) <= 4 * np.spacing(np.abs(ref[small]) + 1e-300))
    v = np.concatenate([rng.uniform(1e-12, 1e4, 100_000), np.logspace(-30, 30, 1000)])
    g, h = np.empty_like(v), np.empty_like(v)
    lib.emu_sqrt_rh(v.ctypes.data_as(dp), g.ctypes.data_as(dp), h.ctypes.data_as(dp), v.size)
    assert np.max(np.abs(g - np.sqrt(v)) / np.spacing(np.sqrt(v))) <= 1.0
    assert np.max(np.abs(h - 0.5 / np.sqrt(v)) / np.spacing(0.5 / np.sqrt(v))) <= 2.0


def test_aggregation_kernels_logic():
    """Per-residue (one thread per segment, strict atom order) and per-class sums."""
    import ctypes as C
    lib = emu._load()
    rng = np.random.default_rng(3)
    n = 5000
    v = rng.uniform(0, 50, n)
    cuts = np.sort(rng.choice(np.arange(1, n), 700, replace=False))
    seg = np.concatenate([[0], cuts, [n]]).astype(np.int64)
    out = np.full(len(seg) - 1, np.nan)
    dp, lp = C.POINTER(C.c_double), C.POINTER(C.c_int64)
    lib.emu_segsum_small(v.ctypes.data_as(dp), seg.ctypes.data_as(lp), len(seg) - 1, out.ctypes.data_as(dp))
    want = []
    for a, b in zip(seg[:-1], seg[1:]):
        t = 0.0
        for x in v[a:b]:
            t += x
        want.append(t)
    assert np.array_equal(out, np.array(want))
    cls = rng.integers(0, 3, n).astype(np.uint8)
    offs = np.array([0, 0, 1, 300, 4999, n], dtype=np.int64)      # empty, single-atom and long structures
    got = np.full(3 * (len(offs) - 1), np.nan)
    lib.emu_class_sums(v.ctypes.data_as(dp), cls.ctypes.data_as(C.POINTER(C.c_ubyte)), offs.ctypes.data_as(lp),
                       len(offs) - 1, got.ctypes.data_as(dp))
    got = got.reshape(-1, 3)
    for s in range(len(offs) - 1):
        sl = slice(offs[s], offs[s + 1])
        for c in range(3):
            assert abs(got[s, c] - v[sl][cls[sl] == c].sum()) <= 1e-9 * max(1.0, got[s, c])
    assert np.all(got[0] == 0)


def test_residue_area_kernel_logic():
    """Per-residue node areas and relative areas: sequential sums in atom order, 100*abs/ref, NaN without reference."""
    import ctypes as C
    lib = emu._load()
    rng = np.random.default_rng(5)
    n, nres = 3000, 400
    v = rng.uniform(0, 40, n)
    cls = rng.integers(0, 3, n).astype(np.uint8)
    bb = rng.integers(0, 2, n).astype(np.uint8)
    first = np.concatenate([[0], np.sort(rng.choice(np.arange(1, n), nres - 1, replace=False)), [n]]).astype(np.int64)
    table = rng.uniform(50, 250, (7, 5))
    table[3, 2] = 0.0                                                   # a zero reference (GLY side chain)
    rows = rng.integers(-1, 7, nres).astype(np.int16)
    A = np.full(6 * nres, np.nan); R = np.full(5 * nres, np.nan)
    dp = C.POINTER(C.c_double)
    lib.emu_residue_areas(v.ctypes.data_as(dp), cls.ctypes.data_as(C.POINTER(C.c_ubyte)), bb.ctypes.data_as(C.POINTER(C.c_ubyte)),
                          first.ctypes.data_as(C.POINTER(C.c_int64)), rows.ctypes.data_as(C.POINTER(C.c_short)),
                          table.ctypes.data_as(dp), nres, A.ctypes.data_as(dp), R.ctypes.data_as(dp))
    A, R = A.reshape(-1, 6), R.reshape(-1, 5)
    with np.errstate(all="ignore"):
        for r in range(nres):
            sl = slice(first[r], first[r + 1])
            want = [0.0] * 6
            for a, c, m in zip(v[sl], cls[sl], bb[sl]):                 # strictly sequential, like the kernel
                want[0] += a
                want[1 if m else 2] += a
                want[{0: 4, 1: 3, 2: 5}[int(c)]] += a
            assert A[r].tolist() == want
            if rows[r] < 0:
                assert np.all(np.isnan(R[r]))
            else:
                expect = 100.0 * np.array(want[:5]) / table[rows[r]]
                assert np.array_equal(R[r], expect, equal_nan=True)


def test_closed_form_sweep_against_the_oracle(oracle_lib):
    """lr2_sweep (round 4: closed form over the final components) through the arc union, on arcs with raw end points
    (start < 0, end > 2 pi, half-widths up to pi, up to 13 arcs, deep stacks) against exposed_arc_length of the
    oracle on the same arcs split at the origin the way the reference stores them (src/sasa_lr.c:340-351)."""
    import ctypes as C
    lib = emu._load()
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
    lib.emu_arc_union.argtypes = [dp, ip, C.c_int, C.c_int, dp]
    T = 2 * np.pi
    rng = np.random.default_rng(7)
    worst = 0.0
    for trial in range(4000):
        k = int(rng.integers(1, 14))
        mids = np.sort(rng.uniform(0, T, k))
        halves = np.minimum(rng.uniform(0.001, rng.choice([0.05, 0.3, 1.0, 3.1]), k), np.pi)
        raw = np.stack([mids - halves, mids + halves], 1).ravel()
        split = []
        for lo, hi in zip(mids - halves, mids + halves):
            if lo < 0 and hi > T: split += [0, T]
            elif lo < 0: split += [0, hi, lo + T, T]
            elif hi > T: split += [0, hi - T, lo, T]
            else: split += [lo, hi]
        first = np.array([0, k], dtype=np.int32)
        out = np.zeros(1)
        lib.emu_arc_union(raw.ctypes.data_as(dp), first.ctypes.data_as(ip), 1, 8, out.ctypes.data_as(dp))
        worst = max(worst, abs(out[0] - oracle_lib.exposed_arc_length(split)))
    assert worst < 1e-13


def test_exact_arithmetic_build_holds_a_tighter_bar(tmp_path):
    """Where the L&R tolerance goes (round-3 advisor: keep the exact build honest).  The shipped kernel spends part of
    the 1e-4 A^2 contract on speed: a degree-12 acos (6e-13) and one Goldschmidt step for 1/(2 Ri') and 1/(2 dij) (4e-15).
    The same sources built with -DACOS2_DEG=14 -DLR2_EXACT_H2 (degree-14 acos, reciprocal roots to ~1 ulp) must sit
    within 5e-11 A^2 of the reference's golden areas (observed 1.6e-11; the fast build 4.5e-11 on the same inputs, held to
    1e-9) - what is left then is the closed-form slice plane against the reference's accumulated z - so a regression of the LOGIC cannot hide behind the
    looser bar of the fast arithmetic.  Runs in a child process (the emulation library is loaded once per process)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = str(tmp_path / "libsasa_emu_exact.so")
    subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-DSASA_EMU", "-DACOS2_DEG=14", "-DLR2_EXACT_H2",
                    "-shared", "-o", so, os.path.join(root, "tests", "emu", "emu.cpp"), "-lm"], check=True)
    code = ("import sys, numpy as np; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "from conftest import load_golden; from emu import run_batch\n"
            "worst = 0.0\n"
            "for name, key, res in (('1ubq', 'lr20', 20), ('1ubq', 'lr100', 100), ('3bzd_trimmed', 'lr20', 20)):\n"
            "    g = load_golden(name); s, *_ = run_batch(True, g['xyz'], g['radii'], resolution=res)\n"
            "    worst = max(worst, float(np.max(np.abs(s - g[key]))))\n"
            "print('WORST %%.3e' %% worst)\n") % (os.path.join(root, "tests"), root)
    out = subprocess.run([sys.executable, "-c", code], check=True, capture_output=True, text=True,
                         env=dict(os.environ, SASA_EMU_SO=so)).stdout
    worst = float(out.split("WORST")[1])
    print(f"\n[exact build] max |dSASA| vs the golden areas: {worst:.3g}")
    assert worst < 5e-11


def test_tiles_with_too_many_candidate_items_are_handed_on(tmp_path):
    """Round-4 advisor (high): P1 decodes a work item's place in its row with a 15-bit index, so a tile whose candidate
    rows hold 2^15 work items or more (a giant radius that puts tens of thousands of small atoms into one cell) must
    not be decoded at all - it is handed on like a tile whose lists do not fit (halves, the second launch, then atom by
    atom to the slab launch, which walks its candidates one by one).  The limit itself needs > 16 000 atoms in a cell
    (a GPU test: tests/test_adversarial.py); here the same sources are built with a limit of 300 / 40 items, so that
    many / nearly all atoms of an ordinary structure take the hand-on path down to the slab launch (whose first-generation
    arithmetic differs from the tile kernel's in the last bits: both are held to LR_TOL against the reference's areas)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for limit in (0, 300, 40):
        so = str(tmp_path / f"libsasa_emu_lim{limit}.so")
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-DSASA_EMU"] +
                       ([f"-DLR2_P1_ITEMS_MAX={limit}"] if limit else []) +
                       ["-shared", "-o", so, os.path.join(root, "tests", "emu", "emu.cpp"), "-lm"], check=True)
        code = ("import sys, numpy as np; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
                "from conftest import load_golden; from emu import run_batch\n"
                "g = load_golden('1ubq'); s, _, _, st = run_batch(True, g['xyz'], g['radii'], resolution=20)\n"
                "print('ERR', float(np.max(np.abs(s - g['lr20']))), 'FB', st['fallback_tiles'], 'SLAB', st['slab_tiles'])\n"
                ) % (os.path.join(root, "tests"), root)
        out = subprocess.run([sys.executable, "-c", code], check=True, capture_output=True, text=True,
                             env=dict(os.environ, SASA_EMU_SO=so)).stdout.split()
        outs[limit] = dict(err=float(out[1]), fb=int(out[3]), slab=int(out[5]))
    assert outs[0]["err"] < LR_TOL and outs[0]["slab"] == 0
    assert outs[300]["fb"] > outs[0]["fb"] and 0 < outs[300]["slab"] < 100 and outs[300]["err"] < LR_TOL  # halves, second launch, a few atoms to the slab
    assert outs[40]["slab"] > 500 and outs[40]["err"] < LR_TOL                                            # nearly every atom through the slab launch


def test_arc_pass_lane_dealing_does_not_change_a_bit(tmp_path):
    """Round 6: on dense tiles the arc pass deals its 64 lanes out by the items' arc counts (n_i = ceil(arcs_i / g) lanes per
    item, every lane a run of consecutive arcs of the item's beta-sorted list, partial unions merged pairwise).  The union's
    components are minima and maxima of the same end points however the arcs are dealt out, so three builds of the same
    sources must give identical bits on protein-density input: the shipped one, one with the dealing switched off (rounds
    4 - 5: two or four lanes per item by list position, or the queue), and one whose lanes see their arcs through windows
    of 8 list positions instead of 64, so that every lane walks the several-windows path that the shipped build takes only
    for an item with a list beyond 64 neighbors, few arcs and a single lane."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for tag, flags in (("shipped", []), ("fixed", ["-DLR2_ARC_BALANCED=0"]), ("window8", ["-DLR2_ARC_WINDOW=8"])):
        so = str(tmp_path / f"libsasa_emu_{tag}.so")
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-DSASA_EMU"] + flags +
                       ["-shared", "-o", so, os.path.join(root, "tests", "emu", "emu.cpp"), "-lm"], check=True)
        code = ("import sys, numpy as np; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
                "from conftest import load_golden; import emu; from emu import run_batch\n"
                "L = emu._load(); res = []\n"
                "for name, ta in (('1ubq', 3), ('1ubq', 4), ('1a0q', 3)):\n"
                "    g = load_golden(name); L.emu_set_lr2(1, ta, 0); emu.set_lr2_opts(True, False)\n"
                "    s, _, _, st = run_batch(True, g['xyz'], g['radii'], resolution=20, cap_idx=96)\n"
                "    assert np.max(np.abs(s - g['lr20'])) < 1e-8 and st['TA'] == ta\n"
                "    res.append(s)\n"
                "np.save(sys.argv[1], np.concatenate(res))\n") % (os.path.join(root, "tests"), root)
        out = str(tmp_path / f"{tag}.npy")
        subprocess.run([sys.executable, "-c", code, out], check=True, env=dict(os.environ, SASA_EMU_SO=so))
        outs[tag] = np.load(out)
    assert np.array_equal(outs["shipped"], outs["fixed"])
    assert np.array_equal(outs["shipped"], outs["window8"])


def test_contained_caps_chooser_and_the_room_of_its_lists():
    """Host side of P1.5 (lr2_prune_want / lr2_prune_arg, what gpu_engine.hip puts into Lr2Args::prune): 4 caps wanted per list
    from 32 slices on, 3 on sparse input below, none at protein density below (measured: DESIGN.md 4); never more than a
    list holds; and off when the two lists per atom (a histogram word and four caps of 20 bytes each) do not fit where
    P3's sort keys will be (8 bytes per record of the pool)."""
    L = emu._load()
    assert L.emu_lr2_prune(100, 0, 3, 104) == 4 and L.emu_lr2_prune(100, 1, 2, 120) == 4 and L.emu_lr2_prune(32, 0, 6, 224) == 4
    assert L.emu_lr2_prune(20, 0, 6, 224) == 3 and L.emu_lr2_prune(31, 0, 4, 200) == 3
    assert L.emu_lr2_prune(20, 1, 3, 230) == 0 and L.emu_lr2_prune(20, 1, 4, 230) == 0
    for ta in range(1, 8):
        need = (16 + 20 * 4) * 2 * ta      # bytes of the lists
        assert L.emu_lr2_prune(100, 0, ta, need // 8) == 4 and L.emu_lr2_prune(100, 0, ta, need // 8 - 1) == 0
        # ... and the layout really has that room: the keys' region is 8 bytes per pool record
        assert 8 * (need // 8) >= need


def test_dropping_contained_caps_does_not_change_a_bit(tmp_path):
    """Round 6, P1.5 (lr2_prune_contained): a neighbor whose cap on the atom's sphere lies inside another neighbor's cap cuts,
    on every slice, an arc inside that neighbor's arc; its record is dropped before the pair records are made (45 % of all
    records on coils and proteins alike).  The union's components are minima and maxima of end points that the dropped arcs
    never supply - provided the two arcs are nested AS NUMBERS, hence the phase's same-side-of-the-cut rule -, so every
    area must keep its bits whatever the phase drops: off, 1, 2, 4 caps wanted per list; at 20 and at 100 slices; on a
    coil, on the reference's 1a0q at protein density (tile of three atoms: the cover filter and the dealt arc pass behind
    it), and on hostile geometry - neighbors placed just either side of beta's cut (the negative x axis), spheres wholly
    inside a neighbor's, twins at 1e-9 A, a ring of equal caps."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, numpy as np; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "from conftest import load_golden; import emu, tools; from emu import run_batch\n"
            "L = emu._load(); res = []\n"
            "xyz, r = tools.coil(1500, 77)[:2]\n"
            "for ns in (20, 100):\n"
            "    emu.set_lr2_opts(True, False); L.emu_set_lr2(1, 3, 0)\n"
            "    res.append(run_batch(True, xyz, r, resolution=ns)[0])\n"
            "g = load_golden('1a0q')\n"
            "s, _, _, st = run_batch(True, g['xyz'], g['radii'], resolution=20, cap_idx=96)\n"
            "assert np.max(np.abs(s - g['lr20'])) < 1e-8 and st['TA'] == 3\n"
            "res.append(s)\n"
            "rng = np.random.default_rng(5)\n"
            "pts = [(0.0, 0.0, 0.0, 1.8)]\n"
            "for k in range(40):                      # around the cut of beta: directions (-1, +-tiny, z)\n"
            "    d = 2.0 + 2.5 * rng.random(); e = (1 if k %% 2 else -1) * 10.0 ** rng.uniform(-12, -1)\n"
            "    pts.append((-d, e * d, rng.uniform(-1.5, 1.5), rng.choice([1.2, 1.6, 1.9])))\n"
            "pts += [(0.3, 0.1, 0.2, 3.5), (0.31, 0.1, 0.2, 3.5 + 1e-9), (0.3 + 1e-9, 0.1, 0.2, 3.5)]   # spheres that hold atom 0; twins\n"
            "for k in range(12): pts.append((3.0 * np.cos(k * np.pi / 6), 3.0 * np.sin(k * np.pi / 6), 0.0, 1.7))   # a ring of equal caps\n"
            "pts += [(6.0 + 3 * rng.random(), 4 * rng.random() - 2, 4 * rng.random() - 2, 1.5 + 0.5 * rng.random()) for _ in range(60)]\n"
            "p = np.array(pts)\n"
            "for ns in (20, 100):\n"
            "    res.append(run_batch(True, p[:, :3].copy(), p[:, 3].copy(), resolution=ns)[0])\n"
            "np.save(sys.argv[1], np.concatenate(res))\n") % (os.path.join(root, "tests"), root)
    outs = {}
    for want in ("0", "1", "2", "4"):
        out = str(tmp_path / f"prune{want}.npy")
        subprocess.run([sys.executable, "-c", code, out], check=True, env=dict(os.environ, EMU_LR2_PRUNE=want))
        outs[want] = np.load(out)
    assert np.all(np.isfinite(outs["0"][:1500 * 2]))
    for want in ("1", "2", "4"):
        assert np.array_equal(outs["0"], outs[want], equal_nan=True), (want, float(np.nanmax(np.abs(outs["0"] - outs[want]))))


def test_shape_builds_and_the_compact_cell_table_give_the_generic_bits(oracle_lib):
    """Round-4 advisor (low): the CPU emulation ran the Lee-Richards tile kernel only in its generic build (SHAPE 0) over
    the dense cell table.  Here the builds with a compile-time tile shape - 6 x 20 (coils), 3 x 100 (100 slices), 3 x 20
    and 4 x 20 with three mask words and the cover filter (protein density) - run where launch_lr2_main would pick them,
    and the cells are looked up in the COMPACT table (a bit per cell, occupied cells before each 32-cell word, first atoms
    of the occupied cells: cell_rank, lr2_pre_b / b2), built the way k_sort_struct's stage F leaves it, including a cell
    count that is a multiple of 32 (the sentinel word).  Every variant must reproduce the generic build's areas bit for
    bit (the device-side twins: test_shape_builds_give_the_generic_builds_bits,
    test_cell_sort_in_one_kernel_equals_the_general_pipeline in tests/test_gpu_parity.py)."""
    L = emu._load()
    g = load_golden("1ubq")
    coil = tools.coil(1500, 77)
    cases = [("coil, 6 x 20 (shape 1)", 1, coil[0], coil[1], 20, 0, 0, 0),
             ("coil, 3 x 100 (shape 2)", 2, coil[0][:600], coil[1][:600], 100, 0, 0, 128),
             ("1ubq, 3 x 20, three mask words (shape 3)", 3, g["xyz"], g["radii"], 20, 3, 96, 0),
             ("1ubq, 4 x 20, three mask words (shape 4)", 4, g["xyz"], g["radii"], 20, 4, 96, 0)]
    try:
        for name, shape, x, r, ns, ta, cap, pool in cases:
            L.emu_set_lr2(1, ta, 0)
            emu.set_lr2_opts(False, False)
            base, _, _, st0 = run_batch(True, x, r, resolution=ns, cap_idx=cap, pool=pool)
            assert L.emu_last_lr2_variant() == 0
            assert close(base, oracle_lib.lee_richards(x.reshape(-1, 3), r, 1.4, ns)), name
            for shapes, compact in ((True, False), (False, True), (True, True)):
                emu.set_lr2_opts(shapes, compact)
                got, _, _, st = run_batch(True, x, r, resolution=ns, cap_idx=cap, pool=pool)
                assert L.emu_last_lr2_variant() == (shape if shapes else 0) | (16 if compact else 0), (name, shapes, compact)
                assert np.array_equal(got, base), (name, shapes, compact)
                assert st["TA"] == st0["TA"]
        # a grid whose cell count is a multiple of 32: the entry behind the last cell opens a table word of its own
        emu.set_lr2_opts(False, False)
        L.emu_set_lr2(1, 0, 0)
        found = False
        for n in range(60, 400, 7):
            x, r = tools.coil(n, 500 + n)
            a, _, _, st = run_batch(True, x, r)
            if st["cells"] % 32 == 0:
                emu.set_lr2_opts(True, True)
                b, *_ = run_batch(True, x, r)
                emu.set_lr2_opts(False, False)
                assert np.array_equal(a, b)
                found = True
                break
        assert found, "no coil with a cell count that is a multiple of 32 among the sizes tried"
    finally:
        emu.set_lr2_opts(False, False)
        L.emu_set_lr2(1, 0, 0)
