"""The drivers of BASELINE configs[3] / configs[4] over a LIST of devices (row E2 of the round-4 review), and the
version-2 binary cache behind the cache sweep.

What replaces what: the reference processes one file per run (src/main.cc:763-779) and spreads one structure over
<= 16 pthreads (src/sasa_lr.c:219-253); here batches of files / shards of frames are dealt to the workers of several
devices from one shared list, with ONE done-list.  On the one-GPU box of the driver the device lists [0, 0, 0] and
[0] * 8 run the several-device code (one worker / lane set, context and stream per entry) on device 0; the
distinct-device variants run wherever two GPUs are visible.  The bar everywhere: the bits (bytes of the result files)
of the single-device drivers."""
import os
import signal
import subprocess
import sys
import time

import numpy as np
import pytest

import tools
from freesasa_amd import ingest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PDB = os.path.join(ROOT, "tests", "golden", "pdb")
CIF = os.path.join(ROOT, "tests", "golden", "cif")


def fixture(name):
    return os.path.join(CIF if name.endswith(".cif") else PDB, name)


NAMES = ["1ubq.pdb", "empty.pdb", "3bkr.cif", "1a0q.pdb", "does_not_exist.pdb", "5dx9.pdb", "icode.pdb", "1ubq.cif", "3bzd_trimmed.pdb"]


# ---------------------------------------------------------------------------------------------- CPU: cache v2, helpers

def _big_batch(copies=240):
    text = open(fixture("1ubq.pdb"), "rb").read()
    other = open(fixture("3bkr.pdb"), "rb").read()
    return ingest.load_pdb_texts([text if k % 3 else other for k in range(copies)], n_threads=2)


def test_usable_cpus_is_the_cgroup_aware_count():
    n = ingest.usable_cpus()
    assert 1 <= n <= (os.cpu_count() or 1)
    assert n == tools.usable_cpus()           # the Python twin bench.py divides among its ranks


def test_device_to_numa_node_mapping_on_a_made_up_sysfs_tree(tmp_path):
    """Round-5 review, item 7b: a device's lanes and page-locked staging run on the socket the GPU hangs off
    (engine_internal.h, DeviceNodeScope).  The mapping - PCI address -> numa_node -> the node's cpulist - on a sysfs tree
    made up here: an 8-GPU, two-socket box as /sys shows it."""
    import ctypes as C
    import freesasa_amd as fa
    L = fa.lib()
    L.freesasa_gpu_test_node_cpus.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_int), C.c_int]
    root = tmp_path / "sys"
    nodes = {0: "0-63,128-191", 1: "64-127,192-255\n"}
    for k, text in nodes.items():
        d = root / "devices" / "system" / "node" / f"node{k}"
        d.mkdir(parents=True)
        (d / "cpulist").write_text(text)
    gpus = {"0000:05:00.0": 0, "0000:26:00.0": 0, "0000:85:00.0": 1, "0000:c6:00.0": 1, "0000:e5:00.0": -1}
    for addr, node in gpus.items():
        d = root / "bus" / "pci" / "devices" / addr
        d.mkdir(parents=True)
        (d / "numa_node").write_text(f"{node}\n")

    def cpus(addr, cap=1024):
        buf = (C.c_int * cap)()
        n = L.freesasa_gpu_test_node_cpus(str(root).encode(), addr.encode(), buf, cap)
        return n, list(buf[:max(0, min(n, cap))])

    n, got = cpus("0000:05:00.0")
    assert n == 128 and got == list(range(0, 64)) + list(range(128, 192))
    n, got = cpus("0000:C6:00.0")                        # (HIP prints the address in upper case on some versions)
    assert n == 128 and got[0] == 64 and got[-1] == 255
    assert cpus("0000:e5:00.0") == (0, [])               # no node named: nothing is bound
    assert cpus("0000:ff:00.0")[0] == -1                 # not in the tree
    n, got = cpus("0000:26:00.0", cap=4)                 # a short buffer still learns the count
    assert n == 128 and got == [0, 1, 2, 3]
    (root / "devices" / "system" / "node" / "node1" / "cpulist").write_text("64-,3\n")
    assert cpus("0000:85:00.0")[0] == -1                 # a list it cannot read is not half-used


def test_cache_v2_parallel_load_equals_the_serial_one(tmp_path):
    """Version 2 of the cache file: every array checksummed in 1 MiB pieces, read and verified by several threads."""
    b = _big_batch()
    assert 24 * b.n_atoms > 3 << 20           # the coordinates span several pieces
    f = tmp_path / "big.fsab"
    b.save(f)
    one, four = ingest.load_cache(f, n_threads=1), ingest.load_cache(f, n_threads=4)
    for name in ("xyz", "radii", "atom_class", "atom_backbone", "offsets", "res_first", "res_offsets", "res_ref", "status"):
        x = getattr(b, name)
        assert x.tobytes() == getattr(one, name).tobytes() == getattr(four, name).tobytes(), name
    # one flipped bit in the middle of the coordinates: refused by every reader count
    raw = bytearray(f.read_bytes())
    pos = 128 + 2 * ((8 * (b.n_structs + 1) + 15) & ~15) + ((4 * b.n_structs + 15) & ~15) + (5 << 19)
    raw[pos] ^= 1
    g = tmp_path / "flipped.fsab"
    g.write_bytes(bytes(raw))
    for nt in (1, 3):
        with pytest.raises(RuntimeError, match=f"code {ingest.EFORMAT}"):
            ingest.load_cache(g, n_threads=nt)


def test_cache_partial_reader_returns_verified_runs_of_atoms(tmp_path):
    """freesasa_ingest_cache_read_atoms: what a sweep reads of a cache - coordinates, radii, classes of a run of atoms -
    equals the batch's slices for runs inside a piece, across piece borders and over whole pieces; a damaged piece
    fails the runs that touch it and ONLY those."""
    b = _big_batch()
    f = tmp_path / "big.fsab"
    b.save(f)
    c = ingest.Cache(f)
    assert (c.n_structs, c.n_atoms) == (b.n_structs, b.n_atoms)
    assert np.array_equal(c.offsets, b.offsets) and np.array_equal(c.status, b.status)
    per_piece = (1 << 20) // 24               # atoms in one piece of the coordinates (not a whole number: borders cut atoms)
    rng = np.random.default_rng(11)
    runs = [(0, 10), (0, b.n_atoms), (per_piece - 3, per_piece + 5), (per_piece, 2 * per_piece + 1), (b.n_atoms - 7, b.n_atoms), (5, 5)]
    runs += [tuple(sorted(rng.integers(0, b.n_atoms, 2).tolist())) for _ in range(12)]
    for a0, a1 in runs:
        xyz, r, cls = c.read_atoms(a0, a1)
        assert np.array_equal(xyz, b.xyz.reshape(-1, 3)[a0:a1]) and np.array_equal(r, b.radii[a0:a1]) and np.array_equal(cls, b.atom_class[a0:a1])
    with pytest.raises(RuntimeError):
        c.read_atoms(0, b.n_atoms + 1)
    c.close()
    raw = bytearray(f.read_bytes())
    xyz_start = 128 + 2 * ((8 * (b.n_structs + 1) + 15) & ~15) + ((4 * b.n_structs + 15) & ~15)
    raw[xyz_start + (1 << 20) + 100] ^= 0x10   # second piece of the coordinates
    g = tmp_path / "damaged.fsab"
    g.write_bytes(bytes(raw))
    d = ingest.Cache(g)                        # header, table, offsets are intact
    xyz, _, _ = d.read_atoms(0, per_piece - 1)                          # first piece only: fine
    assert np.array_equal(xyz, b.xyz.reshape(-1, 3)[:per_piece - 1])
    d.read_atoms(3 * per_piece, 3 * per_piece + 50)                     # fourth piece: fine
    with pytest.raises(RuntimeError, match=f"code {ingest.EFORMAT}"):
        d.read_atoms(per_piece + 10, per_piece + 20)                    # inside the damaged piece
    with pytest.raises(RuntimeError, match=f"code {ingest.EFORMAT}"):
        d.read_atoms(0, 2 * per_piece)                                  # a run that crosses it
    d.close()


# ---------------------------------------------------------------------------------------------- GPU: the drivers

@pytest.fixture(scope="module")
def fa():
    import freesasa_amd
    return freesasa_amd


def _device_lists(fa, distinct):
    nd = fa.device_count()
    if distinct:
        if nd < 2:
            pytest.skip("needs two HIP devices")
        return [list(range(nd)), [nd - 1, 0], list(range(nd)) * 2]
    return [[0, 0, 0], [0] * 8]


@pytest.mark.gpu
@pytest.mark.parametrize("distinct", [False, True])
def test_file_sweep_over_a_device_list_equals_the_single_device_sweep(fa, tmp_path, distinct):
    paths = [fixture(n) for n in NAMES] * 4
    for alg, res in ((fa.LEE_RICHARDS, 20), (fa.SHRAKE_RUPLEY, 100)):
        want = fa.sweep_files(paths, alg, resolution=res, batch_atoms=3000, n_threads=2)
        for devs in _device_lists(fa, distinct):
            got = fa.sweep_files(paths, alg, resolution=res, batch_atoms=3000, n_threads=4, devices=devs)
            for g, w in zip(got, want):
                assert np.array_equal(g, w), devs
    # one batch only, more devices than batches
    got = fa.sweep_files(paths[:3], batch_atoms=0, devices=_device_lists(fa, distinct)[-1])
    assert np.array_equal(got[0], fa.sweep_files(paths[:3])[0])
    with pytest.raises(RuntimeError, match="out of range"):
        fa.sweep_files(paths[:3], devices=[0, 99])
    with pytest.raises(RuntimeError, match="device list"):
        fa.sweep_files(paths[:3], devices=[])


@pytest.mark.gpu
@pytest.mark.parametrize("distinct", [False, True])
def test_resumable_sweep_moves_between_device_lists(fa, tmp_path, distinct):
    """One done-list for all devices: a sweep stopped after a few batches on a device list is finished on ONE device,
    and the other way round; results and the result file equal the uninterrupted single-device sweep's."""
    paths = [fixture(n) for n in NAMES] * 4
    want = fa.sweep_files(paths, batch_atoms=3000, n_threads=2)
    ref_done = tmp_path / "ref.done"
    assert fa.sweep_files_resumable(paths, ref_done, batch_atoms=3000, n_threads=2)[0]
    for k, devs in enumerate(_device_lists(fa, distinct)):
        done = tmp_path / f"sweep{k}.done"
        complete, *_ = fa.sweep_files_resumable(paths, done, batch_atoms=3000, n_threads=4, max_new_batches=5, devices=devs)
        assert not complete and len(done.read_text().splitlines()) == 1 + 5
        complete, totals, cls, atoms, status = fa.sweep_files_resumable(paths, done, batch_atoms=3000, n_threads=2, max_new_batches=3)  # one device
        assert not complete and len(done.read_text().splitlines()) == 1 + 8
        complete, totals, cls, atoms, status = fa.sweep_files_resumable(paths, done, batch_atoms=3000, n_threads=4, devices=devs)
        assert complete
        for g, w in zip((totals, cls, atoms, status), want):
            assert np.array_equal(g, w), devs
        assert (tmp_path / f"sweep{k}.done.bin").read_bytes() == (tmp_path / "ref.done.bin").read_bytes()
        assert sorted(done.read_text().splitlines()[1:]) == sorted(ref_done.read_text().splitlines()[1:])


@pytest.mark.gpu
@pytest.mark.parametrize("distinct", [False, True])
def test_cache_sweep_equals_the_file_sweep(fa, tmp_path, distinct):
    """freesasa_gpu_sweep_cache_devices: the sweep of a saved batch reads only coordinates, radii and classes, verified
    piece by piece into page-locked staging by a few lanes per device, and gives the file sweep's arrays."""
    paths = [fixture(n) for n in NAMES] * 6
    b = ingest.load_pdb_files(paths)
    f = tmp_path / "sweep.fsab"
    b.save(f)
    for alg, res in ((fa.LEE_RICHARDS, 20), (fa.SHRAKE_RUPLEY, 100)):
        want = fa.sweep_files(paths, alg, resolution=res)
        for devs in [[0]] + _device_lists(fa, distinct):
            for batch_atoms in (0, 4000):
                got = fa.sweep_cache(f, alg, resolution=res, batch_atoms=batch_atoms, devices=devs)
                for g, w in zip(got, want):
                    assert np.array_equal(g, w), (devs, batch_atoms)
    got = fa.sweep_cache(f, class_sums=False, devices=[0], lanes_per_device=1)           # one lane, totals only
    assert got[1] is None and np.array_equal(got[0], fa.sweep_files(paths)[0])
    raw = bytearray(f.read_bytes())
    raw[128 + 2 * ((8 * (b.n_structs + 1) + 15) & ~15) + ((4 * b.n_structs + 15) & ~15) + 4000] ^= 2      # inside the coordinates
    bad = tmp_path / "bad.fsab"
    bad.write_bytes(bytes(raw))
    with pytest.raises(RuntimeError, match="checksum"):
        fa.sweep_cache(bad, devices=[0, 0])
    with pytest.raises(RuntimeError):
        fa.sweep_cache(tmp_path / "missing.fsab")


@pytest.mark.gpu
@pytest.mark.parametrize("distinct", [False, True])
def test_trajectory_over_a_device_list_is_byte_identical(fa, tmp_path, distinct):
    n, nf = 2500, 29
    base, r = tools.coil(n, 37)
    frames = np.stack([tools.jitter(base, 900 + f, 0.4) for f in range(nf)])
    want_tot, want_sasa = fa.trajectory(frames, r, frames_per_batch=3)
    f64, f32 = tmp_path / "frames.f64", tmp_path / "frames.f32"
    frames.tofile(f64)
    frames.astype(np.float32).tofile(f32)
    assert fa.trajectory_file(f64, r, tmp_path / "t0.bin", tmp_path / "s0.bin", frames_per_batch=3)[0]
    assert fa.trajectory_file(f32, r, tmp_path / "t0f.bin", tmp_path / "s0f.bin", f32=True, frames_per_batch=3)[0]
    for k, devs in enumerate(_device_lists(fa, distinct)):
        tot, sasa = fa.trajectory(frames, r, frames_per_batch=3, devices=devs)
        assert np.array_equal(tot, want_tot) and np.array_equal(sasa, want_sasa), devs
        for alg, res in ((fa.SHRAKE_RUPLEY, 100),):
            a, b_ = fa.trajectory(frames[:7], r, alg=alg, resolution=res, frames_per_batch=2), fa.trajectory(frames[:7], r, alg=alg, resolution=res, frames_per_batch=2, devices=devs)
            assert np.array_equal(a[0], b_[0]) and np.array_equal(a[1], b_[1])
        done, got = fa.trajectory_file(f64, r, tmp_path / f"t{k}.bin", tmp_path / f"s{k}.bin", frames_per_batch=3, devices=devs)
        assert done and got == nf
        assert (tmp_path / f"t{k}.bin").read_bytes() == (tmp_path / "t0.bin").read_bytes()
        assert (tmp_path / f"s{k}.bin").read_bytes() == (tmp_path / "s0.bin").read_bytes()
        assert fa.trajectory_file(f32, r, tmp_path / f"tf{k}.bin", tmp_path / f"sf{k}.bin", f32=True, frames_per_batch=3, devices=devs)[0]
        assert (tmp_path / f"sf{k}.bin").read_bytes() == (tmp_path / "s0f.bin").read_bytes()
        # interrupted on the device list, continued on one device, finished on the list: one done-list
        d = tmp_path / f"d{k}.txt"
        assert not fa.trajectory_file(f64, r, tmp_path / f"u{k}.bin", tmp_path / f"v{k}.bin", d, frames_per_batch=3, max_new_shards=4, devices=devs)[0]
        assert not fa.trajectory_file(f64, r, tmp_path / f"u{k}.bin", tmp_path / f"v{k}.bin", d, frames_per_batch=3, max_new_shards=2)[0]
        assert fa.trajectory_file(f64, r, tmp_path / f"u{k}.bin", tmp_path / f"v{k}.bin", d, frames_per_batch=3, devices=devs)[0]
        assert len(d.read_text().splitlines()) == 1 + 10
        assert (tmp_path / f"u{k}.bin").read_bytes() == (tmp_path / "t0.bin").read_bytes()
        assert (tmp_path / f"v{k}.bin").read_bytes() == (tmp_path / "s0.bin").read_bytes()
    with pytest.raises(RuntimeError, match="out of range"):
        fa.trajectory(frames[:2], r, devices=[0, 64])


@pytest.mark.gpu
def test_multi_device_trajectory_killed_and_resumed(fa, tmp_path):
    """The SIGKILL test of the single-device driver on a device list: a child running the trajectory on [0] * 4 is shot
    once the done-list shows a few shards; the run is finished on another list and on one device, and the files equal an
    uninterrupted single-device run's byte for byte."""
    n, nf = 2000, 240
    base, r = tools.coil(n, 41)
    frames = np.stack([tools.jitter(base, 100 + f % 24, 0.4) for f in range(nf)])
    big = tmp_path / "big.f64"
    frames.tofile(big)
    np.save(tmp_path / "radii.npy", r)
    child = ("import sys, numpy as np; sys.path.insert(0, %r); import freesasa_amd as fa; "
             "fa.trajectory_file(%r, np.load(%r), %r, %r, %r, frames_per_batch=2, devices=[0, 0, 0, 0])"
             % (ROOT, str(big), str(tmp_path / "radii.npy"), str(tmp_path / "t.bin"), str(tmp_path / "s.bin"), str(tmp_path / "d.txt")))
    proc = subprocess.Popen([sys.executable, "-c", child])
    t0 = time.time()
    while time.time() - t0 < 120 and proc.poll() is None:
        if (tmp_path / "d.txt").exists() and len((tmp_path / "d.txt").read_text().splitlines()) > 8:
            break
        time.sleep(0.002)
    killed = proc.poll() is None
    if killed:
        proc.send_signal(signal.SIGKILL)
    proc.wait()
    listed = len((tmp_path / "d.txt").read_text().splitlines()) - 1
    fa.trajectory_file(big, r, tmp_path / "t.bin", tmp_path / "s.bin", tmp_path / "d.txt", frames_per_batch=2, max_new_shards=7, devices=[0, 0])
    assert fa.trajectory_file(big, r, tmp_path / "t.bin", tmp_path / "s.bin", tmp_path / "d.txt", frames_per_batch=2)[0]
    assert fa.trajectory_file(big, r, tmp_path / "t1.bin", tmp_path / "s1.bin", frames_per_batch=2)[0]
    assert (tmp_path / "t.bin").read_bytes() == (tmp_path / "t1.bin").read_bytes()
    assert (tmp_path / "s.bin").read_bytes() == (tmp_path / "s1.bin").read_bytes()
    assert killed and 0 < listed < 120, (killed, listed)
