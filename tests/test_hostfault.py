"""Host-side failure injection (round-5 review, item 6): the n-th allocation / thread creation of the library's OWN
host code fails, n = 1, 2, ... walked through the loaders, the cache reader, the selection parser, the C API and - on
the GPU - every driver.  The reference does the same to itself by interposing malloc / realloc / strdup for its test
process (tests/tools.c:10-48, tests/test_freesasa.c:475-514, tests/test_nb.c:29-44); here the hook is library-local
(freesasa_host_test_fail_after, csrc/hostfault.h) because the test process is Python.

The bar, per call: a failure value with a message or a clean success - never an abort, a std::terminate from a C++
exception crossing the C boundary, a hang or a leak that breaks the next call - and the call after the walk gives the
un-faulted result bit for bit."""
import ctypes as C
import os

import numpy as np
import pytest

import freesasa_amd as fa
import tools
from freesasa_amd import ingest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PDB = os.path.join(ROOT, "tests", "golden", "pdb")
CIF = os.path.join(ROOT, "tests", "golden", "cif")
FILES = [os.path.join(PDB, "1ubq.pdb"), os.path.join(CIF, "3bkr.cif"), os.path.join(PDB, "empty.pdb"),
         os.path.join(PDB, "1a0q.pdb"), os.path.join(PDB, "does_not_exist.pdb"), os.path.join(CIF, "1ubq.cif"),
         os.path.join(PDB, "icode.pdb"), os.path.join(PDB, "3bzd_trimmed.pdb")]


def walk(call, limit=100000):
    """call() -> (ok, message).  Arms the hook with n = 1, 2, ... (steps grow once n is large) until a call makes
    fewer than n allocations.  Returns (faults that fired, calls that failed)."""
    fired = failed = 0
    n = 1
    while n <= limit:
        fa.host_test_fail_after(n)
        try:
            ok, msg = call()
        finally:
            left = fa.host_test_fail_after(0)
        if left > 0:                      # never fired: the walk is over, and that call had nothing injected
            assert ok, msg
            return fired, failed
        fired += 1
        if not ok:
            failed += 1
            assert msg, f"failure without a message at n = {n}"
        n += 1 if n < 48 else max(1, n // 6)
    raise AssertionError("the walk did not end")


def _same_batch(a, b):
    for name in ("xyz", "radii", "atom_class", "atom_backbone", "offsets", "res_first", "res_offsets", "res_ref", "status",
                 "atom_name_raw", "res_name_raw", "res_number_raw", "res_chain_raw"):
        assert getattr(a, name).tobytes() == getattr(b, name).tobytes(), name


# ------------------------------------------------------------------------------------------------ CPU: the C sources

def test_hook_counts_down_and_reports_what_is_left():
    assert fa.host_test_fail_after(0) == 0
    fa.host_test_fail_after(5)
    assert fa.host_test_fail_after(0) == 5      # nothing allocated in between


@pytest.mark.parametrize("threads", [1, 3])
def test_loader_survives_every_allocation_failing(threads):
    want = ingest.load_files(FILES, n_threads=threads)

    def call():
        try:
            got = ingest.load_files(FILES, n_threads=threads)
        except RuntimeError as e:
            return False, str(e)
        # a batch that could be built may carry per-file ENOMEM; every file that is OK holds the right atoms
        for k in range(got.n_structs):
            if got.status[k] == ingest.OK:
                assert got.offsets[k + 1] - got.offsets[k] == want.offsets[k + 1] - want.offsets[k]
            else:
                assert got.status[k] in (want.status[k], ingest.ENOMEM)
        return True, ""

    fired, failed = walk(call)
    assert fired >= 6 and failed >= 3, (fired, failed)
    _same_batch(ingest.load_files(FILES, n_threads=threads), want)


def test_cache_file_paths_survive_every_allocation_failing(tmp_path):
    b = ingest.load_files(FILES, n_threads=2)
    f = tmp_path / "b.fsab"

    def save():
        try:
            b.save(f)
        except RuntimeError as e:
            return False, str(e)
        return True, ""
    fired, failed = walk(save)
    assert fired >= 1 and failed >= 1
    b.save(f)

    for threads in (1, 3):
        def load():
            try:
                _same_batch(ingest.load_cache(f, n_threads=threads), b)
            except RuntimeError as e:
                return False, str(e)
            return True, ""
        fired, failed = walk(load)
        assert fired >= 2 and failed >= 2, (threads, fired, failed)

    def partial():
        try:
            c = ingest.Cache(f)
        except RuntimeError as e:
            return False, str(e)
        try:
            xyz, r, cls = c.read_atoms(10, b.n_atoms - 7)
            assert xyz.tobytes() == b.xyz[10:b.n_atoms - 7].tobytes() and r.tobytes() == b.radii[10:b.n_atoms - 7].tobytes()
        except RuntimeError as e:
            return False, str(e)
        finally:
            c.close()
        return True, ""
    fired, failed = walk(partial)
    assert fired >= 3 and failed >= 3, (fired, failed)
    assert partial() == (True, "")


def test_selection_parser_survives_every_allocation_failing():
    b = ingest.load_files(FILES[:1])
    cmd = "s, (resn ala+arg and not name ca) or (resi 10-20 and symbol n+o) or chain A"
    name, want, warned = b.select(0, cmd)

    def call():
        try:
            _, got, _ = b.select(0, cmd)
        except ValueError as e:
            return False, str(e)
        assert np.array_equal(got, want)
        return True, ""
    fired, failed = walk(call)
    assert fired >= 5 and failed == fired        # every node and mask of the parser is needed
    assert call() == (True, "")


def test_result_allocation_failure_is_the_reference_s_null_with_a_message(tmp_path):
    """ref: tests/test_freesasa.c:475-514 (freesasa_calc_coord with a failing malloc returns NULL).  On a box without
    a GPU the un-faulted call fails too (no CPU path) - with another message; the injected ones must say 'Out of memory'."""
    L = fa.lib()
    libc = C.CDLL(None)
    libc.fopen.restype = C.c_void_p
    libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
    libc.fclose.argtypes = [C.c_void_p]
    L.freesasa_set_err_out.argtypes = [C.c_void_p]
    L.freesasa_get_err_out.restype = C.c_void_p
    xyz, r = tools.globule(40, 3)
    old = L.freesasa_get_err_out()
    for n in (1, 2):
        path = tmp_path / f"err{n}.txt"
        fp = libc.fopen(str(path).encode(), b"w")
        L.freesasa_set_err_out(fp)
        fa.host_test_fail_after(n)
        try:
            with pytest.raises(RuntimeError):
                fa.calc_coord(xyz, r)
        finally:
            assert fa.host_test_fail_after(0) == 0
            libc.fclose(fp)
            if old:
                L.freesasa_set_err_out(old)
        assert "error: Out of memory" in path.read_text()
    if not old:   # back to stderr: the hook takes any non-NULL stream, so hand it the process's own
        libc.fdopen.restype = C.c_void_p
        L.freesasa_set_err_out(libc.fdopen(2, b"w"))


# ------------------------------------------------------------------------------------------------ GPU: the drivers

gpu = pytest.mark.gpu


def _driver_walk(call, baseline):
    want = baseline()
    fired, failed = walk(call)
    got = baseline()
    for a, b in zip(want, got):
        assert np.asarray(a).tobytes() == np.asarray(b).tobytes()
    return fired, failed


@gpu
@pytest.mark.parametrize("parser", ["host", "device"])
def test_file_sweep_on_a_device_list_survives_host_faults(tmp_path, parser):
    files = FILES * 3
    opt = ingest.PARSE_ON_DEVICE if parser == "device" else 0     # (the device-side parser: staging buffers, file tables, the host fallback's batch)

    def run():
        t, c, a, s = fa.sweep_files(files, n_threads=4, batch_atoms=1500, devices=[0, 0, 0], ingest_options=opt)
        return t, c, a, s

    def call():
        try:
            run()
        except RuntimeError as e:
            return False, str(e)
        return True, ""
    fired, failed = _driver_walk(call, run)
    assert fired >= 30 and failed >= 10, (fired, failed)


@gpu
def test_resumable_file_sweep_survives_host_faults(tmp_path):
    files = FILES * 2
    ok, t0, c0, a0, s0 = fa.sweep_files_resumable(files, tmp_path / "ref.done", batch_atoms=1500, devices=[0, 0])
    assert ok
    k = [0]

    def call():
        k[0] += 1
        try:
            fa.sweep_files_resumable(files, tmp_path / f"w{k[0]}.done", n_threads=2, batch_atoms=1500, devices=[0, 0])
        except RuntimeError as e:
            return False, str(e)
        return True, ""
    fired, failed = walk(call)
    assert fired >= 30 and failed >= 10, (fired, failed)
    ok, t1, c1, a1, s1 = fa.sweep_files_resumable(files, tmp_path / "again.done", batch_atoms=1500, devices=[0, 0])
    assert ok and t1.tobytes() == t0.tobytes() and c1.tobytes() == c0.tobytes()


@gpu
def test_cache_sweep_survives_host_faults(tmp_path):
    b = ingest.load_files(FILES * 3, n_threads=2)
    f = tmp_path / "sweep.fsab"
    b.save(f)

    def run():
        return fa.sweep_cache(f, batch_atoms=1500, devices=[0, 0], lanes_per_device=2)

    def call():
        try:
            run()
        except RuntimeError as e:
            return False, str(e)
        return True, ""
    fired, failed = _driver_walk(call, run)
    assert fired >= 10 and failed >= 5, (fired, failed)


@gpu
def test_trajectory_file_survives_host_faults(tmp_path):
    xyz, r = tools.globule(700, 11)
    rng = np.random.default_rng(5)
    frames = (xyz[None] + rng.normal(0, 0.3, (12, 700, 3))).astype(np.float64)
    fpath = tmp_path / "frames.f64"
    frames.tofile(fpath)
    k = [0]

    def run(tag):
        ok, n = fa.trajectory_file(fpath, r, tmp_path / f"{tag}.tot", tmp_path / f"{tag}.sasa", tmp_path / f"{tag}.done",
                                   frames_per_batch=2, devices=[0, 0, 0])
        assert ok and n == 12
        return (tmp_path / f"{tag}.tot").read_bytes(), (tmp_path / f"{tag}.sasa").read_bytes()

    want = run("ref")

    def call():
        k[0] += 1
        try:
            run(f"w{k[0]}")
        except RuntimeError as e:
            return False, str(e)
        return True, ""
    fired, failed = walk(call)
    assert fired >= 10 and failed >= 5, (fired, failed)
    assert run("after") == want
    # a run that a fault interrupted resumes to the same bytes
    fa.host_test_fail_after(25)
    try:
        fa.trajectory_file(fpath, r, tmp_path / "r.tot", tmp_path / "r.sasa", tmp_path / "r.done", frames_per_batch=2, devices=[0, 0, 0])
    except RuntimeError:
        pass
    finally:
        fa.host_test_fail_after(0)
    assert run("r") == want


@gpu
def test_in_memory_batches_survive_host_faults():
    bx, br, offs = tools.coil_batch(12, 400, seed0=3)

    def run_p():
        s, _, t = fa.calc_batch_pipelined(bx, br, offs, lanes=3, chunk_atoms=900)
        return s, t

    def run_d():
        s, _, t = fa.calc_batch_devices(bx, br, offs, [0, 0, 0])
        return s, t

    def run_t():
        frames = np.stack([np.asarray(bx).reshape(-1, 3)[:400] + 0.01 * k for k in range(6)])
        t, s = fa.trajectory(frames, br[:400], frames_per_batch=2, devices=[0, 0])
        return t, s

    for run, need in ((run_p, 6), (run_d, 6), (run_t, 6)):
        def call():
            try:
                run()
            except RuntimeError as e:
                return False, str(e)
            return True, ""
        fired, failed = _driver_walk(call, run)
        assert fired >= need and failed >= 3, (run.__name__, fired, failed)


@gpu
def test_device_pointer_entries_survive_host_faults():
    import torch
    dev = torch.device("cuda:0")
    bx, br, offs = tools.coil_batch(6, 500, seed0=9)
    d_xyz, d_r = torch.from_numpy(bx).to(dev), torch.from_numpy(br).to(dev)
    d_out = torch.empty(len(br), dtype=torch.float64, device=dev)
    ctx = fa.GpuContext(0)
    ctx.lee_richards(d_xyz.data_ptr(), d_r.data_ptr(), offs, d_out.data_ptr())
    want = d_out.cpu().numpy().copy()
    offs2 = np.concatenate([offs[:-1], [offs[-1] - 3, offs[-1]]])      # other offsets: the tables are rebuilt (host vectors)

    def call():
        try:
            ctx.lee_richards(d_xyz.data_ptr(), d_r.data_ptr(), offs2, d_out.data_ptr())
            ctx.lee_richards_async(d_xyz.data_ptr(), d_r.data_ptr(), offs, d_out.data_ptr())
            ctx.wait()
        except RuntimeError as e:
            return False, str(e)
        finally:
            offs2[-2] -= 1      # never the same table twice
        return True, ""
    fired, failed = walk(call)
    assert fired >= 3 and failed >= 3, (fired, failed)
    ctx.lee_richards(d_xyz.data_ptr(), d_r.data_ptr(), offs, d_out.data_ptr())
    assert np.array_equal(d_out.cpu().numpy(), want)
    ctx.close()
