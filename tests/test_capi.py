"""CPU tests of the C boundary: the library loads, exports every symbol the headers declare,
keeps the reference's struct layouts, validates parameters like the reference, and FAILS
LOUDLY (never falls back to a CPU path) when no HIP device is present."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT

import freesasa_amd as fa


@pytest.fixture(scope="module")
def L():
    fa.build()
    return fa.lib()


def _declared_symbols(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = set(re.findall(r"\b(freesasa_[a-z_0-9]+)\s*\(", txt))
    names |= set(re.findall(r"extern\s+const\s+\w+\s+(\w+)\s*;", txt))
    return names


def test_exports_every_declared_symbol(L):
    out = subprocess.run(["nm", "-D", "--defined-only", fa.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {line.split()[-1] for line in out.splitlines() if line.strip()}
    declared = _declared_symbols("freesasa_amd.h") | _declared_symbols("freesasa_gpu.h") | _declared_symbols("freesasa_ingest.h")
    assert {"freesasa_calc_coord", "freesasa_calc_structure", "freesasa_result_free",
            "freesasa_default_parameters", "FREESASA_DEF_NUMBER_THREADS", "freesasa_lee_richards",
            "freesasa_shrake_rupley", "freesasa_gpu_lr_batch_dev", "freesasa_gpu_calc_batch",
            "freesasa_gpu_trajectory", "freesasa_gpu_segment_sums_dev", "freesasa_ingest_pdb_files",
            "freesasa_ingest_pdb_texts", "freesasa_ingest_free", "freesasa_ingest_select", "freesasa_gpu_sweep_files",
            "freesasa_gpu_residue_areas_dev", "freesasa_gpu_class_sums_dev"} <= declared
    missing = declared - exported
    assert not missing, f"declared in include/ but not exported: {sorted(missing)}"


def test_struct_layouts_match_the_reference():
    # src/freesasa.h:232-238 -> 32 bytes; :267-272 -> 56 bytes (ctypes-verified in SURVEY §8a)
    assert C.sizeof(fa.Parameters) == 32 and fa.Parameters.probe_radius.offset == 8
    assert C.sizeof(fa.Result) == 56 and fa.Result.parameters.offset == 24
    assert C.sizeof(fa.CoordT) == 16


def test_default_parameters(L):
    p = fa.Parameters.in_dll(L, "freesasa_default_parameters")
    assert (p.alg, p.probe_radius, p.shrake_rupley_n_points, p.lee_richards_n_slices, p.n_threads) == \
           (fa.LEE_RICHARDS, 1.4, 100, 20, 2)
    assert C.c_int.in_dll(L, "FREESASA_DEF_NUMBER_THREADS").value == 2


def _seam(L, fn, n, **kw):
    xyz = np.zeros(3 * max(n, 1))
    xyz[::3] = np.arange(max(n, 1)) * 3.0
    r = np.ones(max(n, 1))
    sasa = np.full(max(n, 1), -7.0)
    c = fa.CoordT(n, 1, xyz.ctypes.data_as(C.POINTER(C.c_double)))
    p = fa.Parameters(kw.get("alg", 0), 1.4, kw.get("n_points", 100), kw.get("n_slices", 20), kw.get("n_threads", 1))
    dp = C.POINTER(C.c_double)
    return getattr(L, fn)(sasa.ctypes.data_as(dp), C.byref(c), r.ctypes.data_as(dp), C.byref(p)), sasa


def test_seam_validation_matches_reference(L):
    """src/sasa_lr.c:177-187, src/sasa_sr.c:188-194 — checked before any device work."""
    L.freesasa_set_verbosity(fa.V_SILENT)
    try:
        assert _seam(L, "freesasa_lee_richards", 4, n_threads=17)[0] == fa.FAIL
        assert _seam(L, "freesasa_shrake_rupley", 4, n_threads=17)[0] == fa.FAIL
        assert _seam(L, "freesasa_lee_richards", 4, n_slices=0)[0] == fa.FAIL
        assert _seam(L, "freesasa_lee_richards", 4, n_slices=-1)[0] == fa.FAIL
        assert _seam(L, "freesasa_shrake_rupley", 4, n_points=0)[0] == fa.FAIL
        ret, sasa = _seam(L, "freesasa_lee_richards", 0)
        assert ret == fa.WARN and np.all(sasa == -7.0)      # n == 0: WARN, sasa untouched
        ret, sasa = _seam(L, "freesasa_shrake_rupley", 0)
        assert ret == fa.WARN and np.all(sasa == -7.0)
    finally:
        L.freesasa_set_verbosity(fa.V_NORMAL)


def test_verbosity_and_error_stream(L, tmp_path):
    assert L.freesasa_set_verbosity(fa.V_NOWARNINGS) == fa.SUCCESS
    assert L.freesasa_get_verbosity() == fa.V_NOWARNINGS
    assert L.freesasa_set_verbosity(fa.V_NORMAL) == fa.SUCCESS
    libc = C.CDLL(None)
    libc.fopen.restype = C.c_void_p
    libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
    libc.fclose.argtypes = [C.c_void_p]
    L.freesasa_set_err_out.argtypes = [C.c_void_p]
    L.freesasa_get_err_out.restype = C.c_void_p
    path = tmp_path / "err.log"
    fp = libc.fopen(str(path).encode(), b"w")
    L.freesasa_set_err_out(fp)
    assert L.freesasa_get_err_out() == fp
    assert _seam(L, "freesasa_lee_richards", 4, n_threads=1000)[0] == fa.FAIL
    libc.fflush(None)
    libc.fopen.restype = C.c_void_p
    stderr_fp = C.c_void_p.in_dll(libc, "stderr")
    L.freesasa_set_err_out(stderr_fp)
    libc.fclose(fp)
    txt = path.read_text()
    assert "error: L&R does not support more than 16 threads" in txt and txt.startswith("freesasa:")


def test_test_points_are_the_references(oracle_lib):
    for n in (1, 20, 100, 5000):
        assert np.array_equal(fa.test_points(n), oracle_lib.test_points(n))


def test_calc_structure_without_reference_structure_module(L):
    L.freesasa_set_verbosity(fa.V_SILENT)
    try:
        assert not L.freesasa_calc_structure(C.c_void_p(1), None)
    finally:
        L.freesasa_set_verbosity(fa.V_NORMAL)


@pytest.mark.skipif(fa.device_count() > 0, reason="a HIP device is present")
def test_no_gpu_means_loud_failure_not_a_cpu_path(L):
    L.freesasa_set_verbosity(fa.V_SILENT)
    try:
        with pytest.raises(RuntimeError):
            fa.calc_coord(np.zeros((2, 3)), np.ones(2))
        with pytest.raises(RuntimeError, match="no HIP device"):
            fa.calc_batch(np.zeros((2, 3)), np.ones(2), [0, 2])
        with pytest.raises(RuntimeError):
            fa.GpuContext(0)
        with pytest.raises(RuntimeError, match="no HIP device"):
            fa.trajectory(np.zeros((2, 3, 3)), np.ones(3))
    finally:
        L.freesasa_set_verbosity(fa.V_NORMAL)


def test_product_never_references_the_oracle():
    """The oracle and the emulation are test infrastructure: nothing under freesasa_amd/,
    include/ or the product Makefile targets may mention them."""
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "freesasa_amd")):
        for f in files:
            if f.endswith((".c", ".h", ".hip", ".py")):
                txt = open(os.path.join(base, f)).read()
                if re.search(r"#include\s+\"[^\"]*oracle|import\s+oracle|from\s+oracle|sasa_oracle|libsasa_emu", txt):
                    bad.append(f)
    assert not bad
    syms = subprocess.run(["nm", "-D", fa.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle_" not in syms and "emu_run_batch" not in syms


def test_hot_kernels_do_not_spill(L):
    """The tile kernels sit at their register caps (the L&R kernel of lr2_kernels.h: 128 VGPRs for 4 waves per SIMD):
    a spill to scratch costs HBM traffic and time, and an innocent edit elsewhere in a shared function can cause
    one.  The build keeps the compiler's resource report (Makefile)."""
    path = os.path.join(ROOT, "freesasa_amd", "lib", "kernel_resources.txt")
    if not os.path.exists(path):
        pytest.skip("library was built without the resource report")
    txt = open(path).read()
    blocks = re.findall(r"Function Name: (\S+).*?VGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+)", txt, flags=re.S)
    seen = {name: (int(v), int(sc)) for name, v, sc in blocks}
    main = [n for n in seen if n.startswith("_Z10k_lr2_tileILi4ELi0ELi4E")]
    assert main, "main L&R kernel not found in the report"
    assert seen[main[0]][0] <= 128 and seen[main[0]][1] == 0, seen[main[0]]
    # the cell sort of a structure in one 1024-thread workgroup sits at the same limit: round 6 saw an innocent change of how
    # one counter is ADDRESSED put 125 registers into scratch (0.28 -> 0.50 ms per 1e7 atoms) with every test green
    sort = [n for n in seen if "k_sort_struct" in n]
    assert sort and seen[sort[0]][0] <= 128 and seen[sort[0]][1] == 0, seen[sort[0]] if sort else None
    for n, (v, sc) in seen.items():
        if "k_lr2_tileILi" in n and "ELi0ELi4E" in n:          # every main-launch variant
            assert v <= 128 and sc == 0, (n, v, sc)
        if "k_lr_tileILi64ELb0ELi0ELi4ELb1" in n:
            assert sc == 0, n
        if "k_sr_tileILi128ELb0ELi0ELb1" in n:
            # Shrake-Rupley, third arrangement (sr_caps.h; what runs for up to 128 test points): capped for FIVE waves per SIMD -
            # measured in round 6 (4 / 5 / 6 / 7 waves: 3.48 / 3.24 / 3.48 / 4.6 ms on the PDB entries): no register in scratch
            assert v <= 96 and sc == 0, (n, v, sc)
        if "k_sr_tileILi256ELb0ELi0ELb0" in n:
            # Shrake-Rupley is latency-bound (67 % of its issue slots used), so its registers are CAPPED for seven waves
            # per SIMD (gpu_kernels.hip, SR_WPE): measured on the MI355X in round 5, seven waves with a few spilled
            # dwords beat six by 3 %, and the uncapped build (92 registers, five waves) loses 12 %.  What this guards is
            # the cap itself and that the spill stays a handful of dwords.
            assert v <= 72 and sc <= 96, (n, v, sc)
    # the first-generation L&R kernel (resolutions above 256 slices, last-resort launch) is capped at 96 VGPRs for
    # 5 waves per SIMD and may keep a few bytes in scratch (measured: no slower than the spill-free 4-wave build)
    old = [n for n in seen if n.startswith("_Z9k_lr_tileILi64ELb0ELi0ELi5ELb0")]
    assert old and seen[old[0]][0] <= 96 and seen[old[0]][1] <= 16, seen[old[0]] if old else None
