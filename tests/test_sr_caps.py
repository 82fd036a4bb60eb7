"""Shrake-Rupley, third arrangement (freesasa_amd/csrc/sr_caps.h): the table of cap masks and the phases that use it,
on the CPU.  The arrangement claims bit-exactness BY CONSTRUCTION: a point is taken as covered without the reference's
test only if it lies inside the neighbor's cap by more than every rounding involved.  Tested here: that claim for the
table itself (against the cap condition in extended precision on random directions and cap sizes), and the emulated
phases against the second arrangement and the oracle on inputs chosen to reach every branch (all test-point counts
around the 128-point limit, huge coordinates, coincident and nested spheres, a list that overflows)."""
import ctypes as C

import numpy as np
import pytest

import tools
from conftest import load_golden
import emu
from emu import run_batch

import freesasa_amd as fa


def _lib():
    L = emu._load()
    L.emu_sr_captab.argtypes = [C.POINTER(C.c_double), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint32)]
    L.emu_sr_cap_cell.argtypes = [C.c_float, C.c_float, C.c_float, C.c_int]
    L.emu_sr_cap_level.argtypes = [C.c_float, C.c_int]
    L.emu_sr_caps_counts.argtypes = [C.POINTER(C.c_longlong), C.c_int]
    return L


def _table(L, unit, n, l):
    np_ = unit.shape[0]
    out = np.zeros(6 * n * n * l * 8, dtype=np.uint32)
    got = L.emu_sr_captab(np.ascontiguousarray(unit).ctypes.data_as(C.POINTER(C.c_double)), np_, n, l,
                          out.ctypes.data_as(C.POINTER(C.c_uint32)))
    return out.reshape(-1, 2, 4) if got else None


def _bits(words, np_):
    return ((words[..., None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(*words.shape[:-1], -1)[..., :np_].astype(bool)


@pytest.mark.parametrize("n_points,N,Lv,samples", [(100, 16, 32, 20000), (20, 8, 16, 20000), (128, 12, 24, 20000), (1, 4, 4, 20000), (33, 16, 32, 20000),
                                                     (100, 3, 4, 200000)])     # the last: ~900 caps per table entry
def test_table_masks_are_conservative(n_points, N, Lv, samples):
    """For random cap directions v and cosines g: every DEF point satisfies u.v >= g + 1e-4 (the share of the table's margin
    that is NOT spent on the lookup's own rounding), every point outside DEF and BAND satisfies u.v < g - 1e-4."""
    L = _lib()
    U = fa.test_points(n_points).reshape(-1, 3)
    T = _table(L, U, N, Lv)
    assert T is not None and T.shape[0] == 6 * N * N * Lv
    rng = np.random.default_rng(n_points * 1000 + N)
    v = rng.normal(size=(samples, 3)) * rng.uniform(0.5, 8.0, size=(samples, 1))
    v[:2000] = np.round(v[:2000])                       # directions on cell borders and face diagonals
    v = v[np.abs(v).max(axis=1) > 0]
    g = rng.uniform(-1.0, 1.0, size=len(v))
    g[:3000] = np.round(g[:3000] * Lv / 2) * 2 / Lv     # ... and cosines on interval borders
    vf = v.astype(np.float32)
    cell = np.fromiter((L.emu_sr_cap_cell(float(a), float(b), float(c), N) for a, b, c in vf), dtype=np.int64, count=len(vf))
    lev = np.fromiter((L.emu_sr_cap_level(float(x), Lv) for x in g.astype(np.float32)), dtype=np.int64, count=len(g))
    E = T[cell * Lv + lev]
    DEF, BAND = _bits(E[:, 0], n_points), _bits(E[:, 1], n_points)
    assert not (DEF & BAND).any()
    cosv = (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.longdouble) @ U.T.astype(np.longdouble)
    gg = g.astype(np.longdouble)[:, None]
    assert np.all(cosv[DEF] >= np.broadcast_to(gg + 1e-4, cosv.shape)[DEF])
    OUT = ~(DEF | BAND)
    assert np.all(cosv[OUT] < np.broadcast_to(gg - 1e-4, cosv.shape)[OUT])
    # the masks are worth having: most of a mid-sized cap's points are decided by the table
    if n_points >= 100 and N >= 8:
        mid = np.abs(g) < 0.5
        assert BAND[mid].sum() < 0.2 * n_points * mid.sum()


def test_points_that_are_not_unit_vectors_or_too_many_get_no_table():
    L = _lib()
    U = fa.test_points(100).reshape(-1, 3)
    assert _table(L, U * (1 + 1e-9), 8, 8) is None
    assert _table(L, fa.test_points(129).reshape(-1, 3), 8, 8) is None
    assert _table(L, fa.test_points(128).reshape(-1, 3), 8, 8) is not None


def _both(L, xyz, r, npts, probe=1.4, tables=((16, 32), (4, 4)), **kw):
    """counts and areas of the second arrangement and of the third with several table resolutions: identical"""
    up = fa.test_points(npts)
    L.emu_set_sr_caps(0, 0)
    try:
        s0, c0, _, st0 = run_batch(False, xyz, r, probe=probe, resolution=npts, unit_pts=up, **kw)
        cnt = (C.c_longlong * 4)()
        for n, l in tables:
            L.emu_set_sr_caps(n, l)
            L.emu_sr_caps_counts(cnt, 1)
            s1, c1, _, st1 = run_batch(False, xyz, r, probe=probe, resolution=npts, unit_pts=up, **kw)
            L.emu_sr_caps_counts(cnt, 1)
            assert np.array_equal(c1, c0) and np.array_equal(s1, s0), (npts, n, l)
    finally:
        L.emu_set_sr_caps(16, 32)
    return c0, list(cnt), st0


@pytest.mark.parametrize("npts", [1, 2, 7, 20, 33, 64, 100, 127, 128, 129, 200])
def test_third_arrangement_equals_second_for_every_point_count(npts):
    L = _lib()
    g = load_golden("1ubq")
    xyz, r = np.asarray(g["xyz"]).reshape(-1, 3), np.asarray(g["radii"])
    _, cnt, _ = _both(L, xyz, r, npts)
    if npts <= 128:
        assert cnt[0] > 0, "the third arrangement did not run"
    else:
        assert cnt[0] == 0, "more than 128 points must stay with the second arrangement"


def test_third_arrangement_against_the_oracle_and_its_work(oracle_lib):
    L = _lib()
    g = load_golden("1ubq")
    xyz, r = np.asarray(g["xyz"]).reshape(-1, 3), np.asarray(g["radii"])
    L.emu_set_sr_caps(16, 32)
    cnt = (C.c_longlong * 4)()
    L.emu_sr_caps_counts(cnt, 1)
    s, c, _, st = run_batch(False, xyz, r, resolution=100, unit_pts=fa.test_points(100))
    L.emu_sr_caps_counts(cnt, 1)
    ws, wc = oracle_lib.shrake_rupley(xyz, r)
    assert np.array_equal(s, ws) and np.array_equal(c, wc)
    assert np.array_equal(c, g["sr100_counts"]) and np.array_equal(s, g["sr100"])      # the REAL reference's counts and areas (tests/golden)
    n = len(r)
    assert cnt[0] > 30 * n                      # every neighbor record looked up
    assert cnt[1] < 20 * n and cnt[2] == 0      # ... and a handful of (neighbor, point) pairs per atom asked the reference's way (it asks ~800)


def test_inputs_that_reach_the_guarded_branches():
    L = _lib()
    g = load_golden("1ubq")
    xyz, r = np.asarray(g["xyz"]).reshape(-1, 3), np.asarray(g["radii"])
    for shift in (1e3, 1e6, 1e9):
        _both(L, xyz + shift, r, 100)
    # beyond ~1e11 A the bound on the reference's own rounding exceeds the margin: every pair asks for all its points, the list
    # overflows and the threads ask for themselves
    _, cnt, _ = _both(L, xyz[:200] + 1e12, r[:200], 100, tables=((16, 32),))
    assert cnt[2] > 0
    rng = np.random.default_rng(5)
    rr = rng.uniform(0.1, 10, len(r))                      # nested spheres both ways, long lists (second and third launch)
    for probe in (0.0, 1.4, 5.0):
        _both(L, xyz, rr, 100, probe=probe)
    x2 = np.concatenate([xyz[:50], xyz[:50] + 1e-9, xyz[:50] + 1e-3, xyz[:50]])   # (nearly) coincident centres
    r2 = np.concatenate([r[:50], r[:50], r[:50] * 1.001, r[:50] * 0.5])
    _both(L, x2, r2, 100)
    for scale in (1e12, 1e20):                            # absurd but finite: the reciprocal of R_i |v| underflows, the bound refuses
        _both(L, xyz[:40] * scale, r[:40] * scale, 100, probe=0.0, tables=((16, 32),))
    xyz_c, r_c, _ = tools.coil_batch(1, 1500, seed0=77)
    _both(L, xyz_c.reshape(-1, 3), r_c, 100, tables=((16, 32), (2, 2)))


# ----------------------------------------------------------------------------------------------------------- on the device
def _gpu_sr(xyz, r, offsets, npts, env, probe=1.4):
    """one batch through a context of its own; FREESASA_AMD_SR_CAPS is read when a context first sees a set of test points"""
    import os
    import torch
    old = os.environ.get("FREESASA_AMD_SR_CAPS")
    if env is None:
        os.environ.pop("FREESASA_AMD_SR_CAPS", None)
    else:
        os.environ["FREESASA_AMD_SR_CAPS"] = env
    try:
        dev = torch.device("cuda:0")
        dx, dr = torch.from_numpy(np.ascontiguousarray(xyz, dtype=np.float64).reshape(-1)).to(dev), torch.from_numpy(np.ascontiguousarray(r, dtype=np.float64)).to(dev)
        out = torch.empty(len(r), dtype=torch.float64, device=dev)
        cnt = torch.empty(len(r), dtype=torch.int32, device=dev)
        ctx = fa.GpuContext(0)
        try:
            ctx.shrake_rupley(dx.data_ptr(), dr.data_ptr(), offsets, out.data_ptr(), cnt.data_ptr(), probe=probe, n_points=npts)
            st = ctx.stats()
        finally:
            ctx.close()
        return out.cpu().numpy(), cnt.cpu().numpy(), st
    finally:
        if old is None:
            os.environ.pop("FREESASA_AMD_SR_CAPS", None)
        else:
            os.environ["FREESASA_AMD_SR_CAPS"] = old


@pytest.mark.gpu
def test_device_third_arrangement_equals_second_and_the_reference(reference_lib):
    """the shipped arrangement (table 16 x 32), a coarse table (more points asked the reference's way) and the second
    arrangement on the device: identical counts and areas; the reference's own on a sample"""
    xyz, r, offs = tools.coil_batch(40, 3000, seed0=4000)
    parts = [tools.globule(4000, 900 + k) for k in range(10)]
    xyz = np.concatenate([xyz.reshape(-1, 3)] + [p[0].reshape(-1, 3) for p in parts])
    r = np.concatenate([r] + [p[1] for p in parts])
    offs = np.concatenate([offs, offs[-1] + 4000 * np.arange(1, 11)])
    for npts in (100, 20, 128, 129, 1):
        s2, c2, _ = _gpu_sr(xyz, r, offs, npts, "0")
        s3, c3, st = _gpu_sr(xyz, r, offs, npts, None)
        s4, c4, _ = _gpu_sr(xyz, r, offs, npts, "3,4")
        assert np.array_equal(c2, c3) and np.array_equal(s2, s3) and np.array_equal(c2, c4) and np.array_equal(s2, s4), npts
    for k in (0, 17, 45):                                    # the REAL reference on three of the structures, 100 points
        sl = slice(int(offs[k]), int(offs[k + 1]))
        s3, c3, _ = _gpu_sr(xyz[sl], r[sl], [0, sl.stop - sl.start], 100, None)
        want = reference_lib.calc_coord(xyz[sl], r[sl], alg=fa.SHRAKE_RUPLEY, n_points=100)[0]
        assert np.array_equal(s3, np.asarray(want))


@pytest.mark.gpu
def test_device_third_arrangement_on_hostile_inputs():
    g = load_golden("1ubq")
    xyz, r = np.asarray(g["xyz"]).reshape(-1, 3), np.asarray(g["radii"])
    rng = np.random.default_rng(11)
    cases = [(xyz + 1e6, r, 1.4), (xyz + 1e12, r, 1.4), (xyz, rng.uniform(0.1, 10, len(r)), 0.0), (xyz, rng.uniform(0.1, 10, len(r)), 5.0),
             (np.concatenate([xyz[:80], xyz[:80] + 1e-9, xyz[:80] + 1e-3, xyz[:80]]), np.concatenate([r[:80], r[:80], r[:80] * 1.001, r[:80] * 0.5]), 1.4),
             (xyz[:40] * 1e12, r[:40] * 1e12, 0.0), (xyz[:40] * 1e20, r[:40] * 1e20, 0.0)]
    for x, rad, probe in cases:
        s2, c2, _ = _gpu_sr(x, rad, [0, len(rad)], 100, "0", probe)
        s3, c3, _ = _gpu_sr(x, rad, [0, len(rad)], 100, None, probe)
        assert np.array_equal(c2, c3) and np.array_equal(s2, s3)
