"""Batched PDB ingestion (include/freesasa_ingest.h) against vectors minted from
the reference library (tests/golden/make_ingest_golden.py): for every PDB file of the reference's
test suite and a set of corner-case inputs, under ten option sets, the loader must hold exactly
what freesasa_structure_from_pdb() holds — same atoms, bit-identical coordinates and radii, same
classes, same residue boundaries and labels — and fail where the reference fails."""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import ROOT

from freesasa_amd import ingest

PDB = os.path.join(ROOT, "tests", "golden", "pdb")
CIF = os.path.join(ROOT, "tests", "golden", "cif")


def fixture(name):
    return os.path.join(CIF if name.endswith(".cif") else PDB, name)
with open(os.path.join(ROOT, "tests", "golden", "ingest.json")) as fh:
    GOLD = json.load(fh)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def labels_digest(b, lo=0, hi=None):
    hi = b.n_residues if hi is None else hi
    return hashlib.sha256("\n".join(f"{b.res_name[k]}|{b.res_number[k]}|{b.res_chain[k]}" for k in range(lo, hi)).encode()).hexdigest()


@pytest.mark.parametrize("name", sorted(GOLD))
def test_matches_the_reference_reader(name):
    checked = 0
    for opt, exp in GOLD[name].items():
        if exp.get("crash"):        # the reference itself aborts on this input (double free)
            continue
        b = ingest.load_pdb_files([fixture(name)], options=int(opt))
        if exp.get("fail"):
            assert b.status[0] != ingest.OK and b.n_atoms == 0 and b.offsets.tolist() == [0, 0], (name, opt)
        else:
            assert b.status[0] == ingest.OK, (name, opt)
            assert (b.n_atoms, b.n_residues) == (exp["n_atoms"], exp["n_residues"]), (name, opt)
            assert sha(b.xyz) == exp["xyz"] and sha(b.radii) == exp["radii"], (name, opt)
            assert sha(b.atom_class) == exp["classes"] and sha(b.res_first) == exp["res_first"], (name, opt)
            assert labels_digest(b) == exp["labels"], (name, opt)
            assert sha(b.atom_backbone) == exp["backbone"], (name, opt)
            assert sha((b.res_ref >= 0).astype(np.uint8)) == exp["has_reference"], (name, opt)
        checked += 1
    assert checked >= 9


def test_mmcif_and_pdb_of_the_same_entry_agree():
    """ref: tests/test-cli.in:229-236 (the CLI gives the same totals for 1ubq.pdb and 1ubq.cif)."""
    b = ingest.load_pdb_files([fixture("1ubq.pdb"), fixture("1ubq.cif"), fixture("3bkr.pdb"), fixture("3bkr.cif")], n_threads=2)
    assert b.status.tolist() == [0, 0, 0, 0]
    for k in (0, 2):
        a, c = slice(b.offsets[k], b.offsets[k + 1]), slice(b.offsets[k + 1], b.offsets[k + 2])
        assert np.array_equal(b.xyz[a], b.xyz[c]) and np.array_equal(b.radii[a], b.radii[c])
        assert np.array_equal(b.atom_class[a], b.atom_class[c])
        assert np.array_equal(np.diff(b.res_first[b.res_offsets[k]:b.res_offsets[k + 1] + 1]),
                              np.diff(b.res_first[b.res_offsets[k + 1]:b.res_offsets[k + 2] + 1]))


def test_reference_atom_counts_of_its_cli_tests():
    """ref: tests/test-cli.in:143,156,159 (602 / 660 with HETATM / 1231 with hydrogens)."""
    f = [os.path.join(PDB, "1ubq.pdb")]
    assert ingest.load_pdb_files(f).n_atoms == 602
    assert ingest.load_pdb_files(f, options=ingest.INCLUDE_HETATM).n_atoms == 660
    assert ingest.load_pdb_files([os.path.join(PDB, "1d3z.pdb")], options=ingest.INCLUDE_HYDROGEN).n_atoms == 1231


def test_batch_layout_failures_and_thread_count_independence():
    names = ["1ubq.pdb", "empty.pdb", "3bkr.pdb", "does_not_exist.pdb", "syn_short_line.pdb", "icode.pdb", "1a0q.pdb"]
    paths = [os.path.join(PDB, n) for n in names]
    one = ingest.load_pdb_files(paths, n_threads=1)
    many = ingest.load_pdb_files(paths, n_threads=5)
    assert one.status.tolist() == [ingest.OK, ingest.EEMPTY, ingest.OK, ingest.EIO, ingest.EFORMAT, ingest.OK, ingest.OK]
    for f in ("xyz", "radii", "atom_class", "offsets", "res_first", "res_offsets", "status"):
        assert np.array_equal(getattr(one, f), getattr(many, f)), f
    assert one.res_name == many.res_name and one.res_number == many.res_number and one.res_chain == many.res_chain
    # every structure's slice is what the file gives alone; failed inputs are empty structures
    for k, n in enumerate(names):
        lo, hi = one.offsets[k], one.offsets[k + 1]
        rl, rh = one.res_offsets[k], one.res_offsets[k + 1]
        if one.status[k] != ingest.OK:
            assert lo == hi and rl == rh
            continue
        exp = GOLD[n]["0"]
        assert hi - lo == exp["n_atoms"] and rh - rl == exp["n_residues"]
        assert sha(one.xyz[lo:hi]) == exp["xyz"] and sha(one.radii[lo:hi]) == exp["radii"]
        assert sha(np.append(one.res_first[rl:rh] - lo, hi - lo)) == exp["res_first"]
        assert labels_digest(one, rl, rh) == exp["labels"]
    assert one.offsets[-1] == one.n_atoms == len(one.radii) and one.res_first[-1] == one.n_atoms
    # texts in memory == files on disk
    texts = [open(p, "rb").read() if os.path.exists(p) else b"" for p in paths]
    mem = ingest.load_pdb_texts(texts, n_threads=3)
    assert np.array_equal(mem.xyz, one.xyz) and np.array_equal(mem.radii, one.radii)
    assert mem.status.tolist() == [ingest.OK, ingest.EEMPTY, ingest.OK, ingest.EEMPTY, ingest.EFORMAT, ingest.OK, ingest.OK]
    # no inputs at all
    empty = ingest.load_pdb_files([])
    assert empty.n_structs == 0 and empty.n_atoms == 0 and empty.offsets.tolist() == [0]


def test_coordinate_fast_path_equals_strtod_on_random_fields():
    """The loader converts plain decimals exactly (integer / power of ten); anything else goes to
    strtod.  Python's float() is correctly rounded like glibc's strtod, so both must agree bit for
    bit on whatever fits the three 8-character coordinate fields."""
    rng = np.random.default_rng(11)
    fields = []
    for _ in range(4000):
        kind = rng.integers(0, 6)
        if kind == 0:
            v = f"{rng.uniform(-99, 999):7.3f}"
        elif kind == 1:
            v = f"{rng.uniform(-9, 99):7.4f}"[:7]
        elif kind == 2:
            v = f"{rng.integers(-999999, 9999999):7d}"
        elif kind == 3:
            v = f"{rng.uniform(0, 9):7.1e}"[:7]
        elif kind == 4:
            v = f"{rng.uniform(0, 1):7.5f}"[:7]
        else:
            v = f"{rng.integers(0, 9999)}.".rjust(7)
        v = " " + v                       # fields that run together are one number for sscanf too
        assert len(v) == 8
        fields.append(v)
    lines = []
    for k in range(0, len(fields) - 2, 3):
        sec = "".join(fields[k:k + 3])
        assert len(sec) == 24
        lines.append(f"ATOM  {k % 99999:5d}  CA  ALA A{k % 9999:4d}    {sec}  1.00  0.00           C  ")
    b = ingest.load_pdb_texts(["\n".join(lines) + "\n"])
    assert b.status[0] == ingest.OK and b.n_atoms == len(lines)
    want = np.array([[float(fields[k + i]) for i in range(3)] for k in range(0, len(fields) - 2, 3)])
    assert np.array_equal(b.xyz, want)


def test_concurrent_loads_share_the_arena_cache_safely():
    from concurrent.futures import ThreadPoolExecutor
    paths = [fixture(n) for n in ("1ubq.pdb", "3bkr.pdb", "1a0q.pdb", "1ubq.cif", "5dx9.pdb")] * 6
    ref = ingest.load_pdb_files(paths, n_threads=1)
    with ThreadPoolExecutor(max_workers=6) as ex:
        outs = list(ex.map(lambda t: ingest.load_pdb_files(paths, n_threads=1 + t % 4), range(12)))
    for o in outs:
        assert np.array_equal(o.xyz, ref.xyz) and np.array_equal(o.radii, ref.radii)
        assert np.array_equal(o.res_first, ref.res_first) and np.array_equal(o.offsets, ref.offsets)


def test_unsupported_options_are_refused():
    for opt in (1 << 3, 1 << 4, 1 << 9):       # SEPARATE_MODELS, SEPARATE_CHAINS, unknown bit
        with pytest.raises(RuntimeError):
            ingest.load_pdb_files([os.path.join(PDB, "1ubq.pdb")], options=opt)


def test_classifier_spot_values():
    """ProtOr radii of the paper's atom types as the reference config lists them
    (ref: share/protor.config:22-44) and the lookup rules (names are trimmed, no ANY residue)."""
    assert ingest.protor_radius("ALA", " CA ") == (1.88, ingest.APOLAR)
    assert ingest.protor_radius("ALA", "C") == (1.61, ingest.APOLAR)
    assert ingest.protor_radius("ALA", " O  ") == (1.42, ingest.POLAR)
    assert ingest.protor_radius("ALA", " N  ") == (1.64, ingest.POLAR)
    assert ingest.protor_radius("SER", " OG ") == (1.46, ingest.POLAR)
    assert ingest.protor_radius("CYS", " SG ") == (1.77, ingest.POLAR)
    assert ingest.protor_radius("PHE", " CZ ") == (1.76, ingest.APOLAR)
    assert ingest.protor_radius(" DA", " P  ") == (1.8, ingest.POLAR)
    assert ingest.protor_radius("  A", " C1'")[0] == 1.88
    for res, atom in (("XXX", " CA "), ("ANY", " CA "), ("ALA", " XX "), ("ALA", ""), ("", " CA "), ("ALAX", " CA ")):
        assert ingest.protor_radius(res, atom) == (-1.0, ingest.UNKNOWN)
    assert ingest.guess_radius(" C") == 1.7 and ingest.guess_radius("C") == 1.7     # right-justified like "%2s"
    assert ingest.guess_radius("FE") > 0 and ingest.guess_radius("Fe") == -1.0 and ingest.guess_radius(" Q") == -1.0


def test_residue_sums_helper():
    b = ingest.load_pdb_files([os.path.join(PDB, "icode.pdb")])
    v = np.arange(1.0, b.n_atoms + 1)
    got = b.residue_sums(v)
    assert len(got) == b.n_residues and got.sum() == v.sum()
    assert got[0] == v[b.res_first[0]:b.res_first[1]].sum()


@pytest.mark.gpu
def test_pdb_to_sasa_end_to_end_matches_reference_totals():
    """PDB file -> loader -> GPU batch -> totals, classes and per-residue sums, against the
    reference's own published numbers for 1UBQ (ref: tests/test_freesasa.c:155-178,
    tests/data/seq.reference)."""
    import freesasa_amd as fa
    from conftest import read_seq_reference
    b = ingest.load_pdb_files([os.path.join(PDB, "1ubq.pdb"), os.path.join(PDB, "3bzd_trimmed.pdb"),
                               os.path.join(PDB, "1d3z.pdb"), fixture("1ubq.cif")])
    lr, _, lr_tot = fa.calc_batch(b.xyz, b.radii, b.offsets, fa.LEE_RICHARDS, resolution=20)
    sr, _, sr_tot = fa.calc_batch(b.xyz, b.radii, b.offsets, fa.SHRAKE_RUPLEY, resolution=100)
    ubq = slice(b.offsets[0], b.offsets[1])
    polar = b.atom_class[ubq] == ingest.POLAR
    assert abs(lr_tot[0] - 4804.055641) < 1e-5 * 4804.055641
    assert abs(lr[ubq][polar].sum() - 2504.217302) < 1e-5 * 2504.217302
    assert abs(lr[ubq][~polar].sum() - 2299.838339) < 1e-5 * 2299.838339
    assert abs(sr_tot[0] - 4834.716265) < 1e-5 * 4834.716265
    assert abs(sr[ubq][polar].sum() - 2515.821238) < 1e-5 * 2515.821238
    assert abs(sr_tot[1] - 16133.867124) < 1e-5 * 16133.867124          # 3BZD, tests/test_freesasa.c:305-327
    assert abs(sr_tot[2] - 5000.340175) < 1e-5 * 5000.340175            # 1D3Z model 1, :441-451
    assert lr_tot[3] == lr_tot[0] and sr_tot[3] == sr_tot[0]             # 1ubq.cif == 1ubq.pdb (tests/test-cli.in:229-236)
    # the same aggregates on the device (N2): per-structure class sums and per-residue sums
    import torch
    dev = torch.device("cuda:0")
    d_sr = torch.from_numpy(sr).to(dev)
    d_cls = torch.from_numpy(b.atom_class).to(dev)
    d_cs = torch.empty(3 * b.n_structs, dtype=torch.float64, device=dev)
    d_res = torch.empty(b.n_residues, dtype=torch.float64, device=dev)
    ctx = fa.GpuContext(0)
    ctx.class_sums(d_sr.data_ptr(), d_cls.data_ptr(), b.offsets, d_cs.data_ptr())
    ctx.segment_sums(d_sr.data_ptr(), b.res_first, d_res.data_ptr())
    cs = d_cs.cpu().numpy().reshape(-1, 3)
    assert abs(cs[0, 1] - 2515.821238) < 1e-5 * 2515.821238 and abs(cs[0, 0] - 2318.895027) < 1e-5 * 2318.895027
    assert np.allclose(cs.sum(1), sr_tot, rtol=1e-12) and cs[0, 2] == 0
    assert np.allclose(d_res.cpu().numpy(), b.residue_sums(sr), rtol=1e-13, atol=1e-12)
    ctx.close()
    per_res = b.residue_sums(sr)[b.res_offsets[0]:b.res_offsets[1]]
    ref = read_seq_reference()
    assert len(per_res) == len(ref) == 76
    for k, (chain, number, name, area) in enumerate(ref):
        assert (b.res_chain[k], b.res_number[k].strip(), b.res_name[k]) == (chain, number, name)
        assert abs(per_res[k] - area) <= 0.005 + 1e-9                   # the file prints 2 decimals


def read_rsa(name):
    """Rows of a reference --format=rsa file: (residue, chain, number, [abs, rel] x (all, side, main, apolar, polar))
    and the TOTAL line; rel is None where the file says N/A."""
    rows, total = [], None
    with open(os.path.join(ROOT, "tests", "golden", name)) as fh:
        for line in fh:
            if line.startswith("RES "):
                res, chain, number = line[4:7], line[8:11].strip(), line[11:15].strip()
                f = line[16:].split()
                vals = [(float(f[2 * k]), None if f[2 * k + 1] == "N/A" else float(f[2 * k + 1])) for k in range(5)]
                rows.append((res, chain, number, vals))
            elif line.startswith("TOTAL"):
                total = [float(v) for v in line.split()[1:]]
    return rows, total


@pytest.mark.gpu
@pytest.mark.parametrize("pdb, rsa, alg", [("1ubq.pdb", "1ubq.sr100.rsa", "sr"), ("1ubq.pdb", "1ubq.lr20.rsa", "lr"),
                                           ("3bkr.pdb", "3bkr.sr100.rsa", "sr")])
def test_relative_sasa_matches_the_references_rsa_output(pdb, rsa, alg):
    """SURVEY §8(f) N4 (RSA half): per-residue absolute and relative areas computed on the device
    from a PDB file equal the reference CLI's --format=rsa output (fixtures generated by running
    the reference, tests/golden/make_golden.py) to the precision it prints: %.2f abs, %.1f rel, N/A."""
    import torch
    import freesasa_amd as fa
    b = ingest.load_pdb_files([os.path.join(PDB, pdb)])
    if alg == "sr":
        sasa, _, tot = fa.calc_batch(b.xyz, b.radii, b.offsets, fa.SHRAKE_RUPLEY, resolution=100)
    else:
        sasa, _, tot = fa.calc_batch(b.xyz, b.radii, b.offsets, fa.LEE_RICHARDS, resolution=20)
    dev = torch.device("cuda:0")
    d_sasa = torch.from_numpy(sasa).to(dev)
    d_cls, d_bb = torch.from_numpy(b.atom_class).to(dev), torch.from_numpy(b.atom_backbone).to(dev)
    d_abs = torch.empty(6 * b.n_residues, dtype=torch.float64, device=dev)
    d_rel = torch.empty(5 * b.n_residues, dtype=torch.float64, device=dev)
    ctx = fa.GpuContext(0)
    ctx.residue_areas(d_sasa.data_ptr(), d_cls.data_ptr(), d_bb.data_ptr(), b.res_first, d_abs.data_ptr(),
                      res_ref=b.res_ref, ref_table=ingest.residue_reference_table(), d_rel=d_rel.data_ptr())
    ctx.close()
    A, R = d_abs.cpu().numpy().reshape(-1, 6), d_rel.cpu().numpy().reshape(-1, 5)
    rows, total = read_rsa(rsa)
    assert len(rows) == b.n_residues
    # file columns: all, side, main, apolar, polar  <-  device: total, main, side, polar, apolar
    cols = [0, 2, 1, 4, 3]
    for r, (res, chain, number, vals) in enumerate(rows):
        assert (b.res_name[r], b.res_chain[r], b.res_number[r].strip()) == (res.strip(), chain, number)
        for k, (a, rel) in enumerate(vals):
            assert abs(A[r, cols[k]] - a) <= 0.005 + 1e-9, (r, k)
            if rel is None:
                assert not np.isfinite(R[r, cols[k]]), (r, k)
            else:
                assert abs(R[r, cols[k]] - rel) <= 0.05 + 1e-9, (r, k)
    sums = A.sum(0)
    for k, want in enumerate(total):                                   # TOTAL line, one decimal
        assert abs(sums[cols[k]] - want) <= 0.05 + 1e-6
    assert abs(sums[0] - tot[0]) < 1e-9 * tot[0] and np.all(A[:, 5] == 0)


@pytest.mark.gpu
def test_sweep_entry_point_equals_load_then_batch():
    """freesasa_gpu_sweep_files (loader thread || GPU, several batches) gives the totals and class sums
    of the two-step path, keeps the input order and reports failed inputs without failing the sweep."""
    import freesasa_amd as fa
    names = ["1ubq.pdb", "empty.pdb", "3bkr.cif", "1a0q.pdb", "does_not_exist.pdb", "5dx9.pdb", "icode.pdb", "1ubq.cif", "3bzd_trimmed.pdb"]
    paths = [fixture(n) for n in names] * 3
    b = ingest.load_pdb_files(paths)
    ok = b.status == 0
    for alg, res in ((fa.LEE_RICHARDS, 20), (fa.SHRAKE_RUPLEY, 100)):
        keep = np.nonzero(ok)[0]
        offs = np.concatenate([[0], np.cumsum(np.diff(b.offsets)[keep])]).astype(np.int64)
        sasa, _, tot = fa.calc_batch(b.xyz, b.radii, offs, alg, resolution=res)
        for batch_atoms in (0, 3000):                      # one batch / many small batches
            totals, cls, atoms, status = fa.sweep_files(paths, alg, resolution=res, batch_atoms=batch_atoms, n_threads=3)
            assert np.array_equal(status, b.status) and np.array_equal(atoms, np.diff(b.offsets))
            assert np.array_equal(totals[ok], tot) and np.all(totals[~ok] == 0)
            assert np.allclose(cls.sum(1), totals, rtol=1e-12, atol=1e-9)
            k = names.index("1ubq.pdb")
            polar = b.atom_class[b.offsets[k]:b.offsets[k + 1]] == ingest.POLAR
            assert abs(cls[k, 1] - sasa[:len(polar)][polar].sum()) < 1e-9
    assert abs(fa.sweep_files([fixture("1ubq.pdb")])[0][0] - 4804.055641) < 1e-5 * 4804.055641


@pytest.mark.gpu
def test_resumable_sweep_equals_the_uninterrupted_one(tmp_path):
    """freesasa_gpu_sweep_files_resumable: a sweep stopped after every batch and restarted through its done-list
    (results of finished batches come from <done>.bin, only the others are computed) gives exactly the arrays of
    the plain sweep; a finished sweep called again computes nothing; other parameters are refused."""
    import freesasa_amd as fa
    names = ["1ubq.pdb", "empty.pdb", "3bkr.cif", "1a0q.pdb", "does_not_exist.pdb", "5dx9.pdb", "icode.pdb", "1ubq.cif", "3bzd_trimmed.pdb"]
    paths = [fixture(n) for n in names] * 2
    want = fa.sweep_files(paths, batch_atoms=3000, n_threads=2)
    done = tmp_path / "sweep.done"
    calls = 0
    while True:
        complete, totals, cls, atoms, status = fa.sweep_files_resumable(paths, done, batch_atoms=3000, n_threads=2, max_new_batches=1)
        calls += 1
        assert calls < 40
        if complete:
            break
    assert calls > 3
    for got, exp in zip((totals, cls, atoms, status), want):
        assert np.array_equal(got, exp)
    n_lines = len(done.read_text().splitlines())
    complete, totals2, *_ = fa.sweep_files_resumable(paths, done, batch_atoms=3000, n_threads=2)
    assert complete and np.array_equal(totals2, want[0]) and len(done.read_text().splitlines()) == n_lines
    with pytest.raises(RuntimeError):
        fa.sweep_files_resumable(paths, done, batch_atoms=3000, resolution=21)
    # ... and so is the same sweep over a file that changed since (the done-list names sizes and modification times)
    import shutil
    local = tmp_path / "copy_1ubq.pdb"
    shutil.copy(fixture("1ubq.pdb"), local)
    paths2, done2 = [str(local), fixture("3bkr.cif")], tmp_path / "sweep2.done"
    assert fa.sweep_files_resumable(paths2, done2, batch_atoms=3000, n_threads=2)[0]
    with open(local, "a") as fh:
        fh.write("REMARK changed\n")
    with pytest.raises(RuntimeError, match="changed"):
        fa.sweep_files_resumable(paths2, done2, batch_atoms=3000, n_threads=2)


def test_mmcif_row_scanner_equals_the_byte_at_a_time_tokenizer():
    """The SSE2 row scanner of the mmCIF reader (whitespace bit masks, freesasa_amd/csrc/ingest.c
    cif_row_next) against its scalar twin (tests/emu/libingest_scalar.so, the same source built with
    -DFREESASA_INGEST_NO_SIMD) on randomized _atom_site loops: ragged padding, tabs and CRLF, rows
    broken over lines, comments, quoted values with blanks, text fields, keywords in value position,
    loops that end at every offset modulo the 64-byte block."""
    import ctypes as C
    from freesasa_amd.ingest import _CBatch, Batch
    path = os.path.join(ROOT, "tests", "emu", "libingest_scalar.so")
    if not os.path.exists(path):
        pytest.skip("scalar twin not built (make emu)")
    S = C.CDLL(path)
    S.freesasa_ingest_pdb_texts.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_int, C.c_int, C.c_int, C.POINTER(_CBatch)]
    S.freesasa_ingest_free.argtypes = [C.POINTER(_CBatch)]
    S.freesasa_ingest_free.restype = None

    def scalar(texts, options):
        raw = [t.encode() for t in texts]
        arr = (C.c_char_p * len(raw))(*raw)
        lens = (C.c_size_t * len(raw))(*[len(t) for t in raw])
        cb = _CBatch()
        assert S.freesasa_ingest_pdb_texts(arr, lens, len(raw), options, 1, C.byref(cb)) == 0
        try:
            return Batch(cb)
        finally:
            S.freesasa_ingest_free(C.byref(cb))

    rng = np.random.default_rng(11)
    cols = ["group_PDB", "id", "type_symbol", "label_atom_id", "label_alt_id", "label_comp_id", "label_asym_id",
            "label_seq_id", "pdbx_PDB_ins_code", "Cartn_x", "Cartn_y", "Cartn_z", "occupancy", "B_iso_or_equiv",
            "auth_seq_id", "auth_comp_id", "auth_asym_id", "auth_atom_id", "pdbx_PDB_model_num"]
    res = ["ALA", "LEU", "SER", "ASP", "GLY", "DA", "HOH", "loop", "data", "XYZ"]
    atoms = ["N", "CA", "C", "O", "CB", "\"C1'\"", "'O P'", "OXT", "\"N A\"", "H", "_x", "CG"]

    def ws():
        k = rng.integers(0, 12)
        return [" ", "  ", "   ", "\t", " \t ", "      ", " ", "  ", "\n", "\r\n", " \n ", "  # note\n"][k]

    def make_text():
        order = list(rng.permutation(len(cols)))
        out = ["data_fuzz\n", "#\n" if rng.random() < 0.5 else "", "_cell.length_a 10.0\n", "loop_\n"]
        out += ["_atom_site.%s%s" % (cols[k], "\r\n" if rng.random() < 0.1 else "\n") for k in order]
        n = int(rng.integers(1, 60))
        for i in range(n):
            rn, an = res[rng.integers(len(res))], atoms[rng.integers(len(atoms))]
            sym = "H" if an == "H" else an.strip("\"'")[0].upper() if an[0] != "_" else "C"
            seq = str(int(rng.integers(-5, 400)))
            vals = {"group_PDB": "ATOM" if rng.random() < 0.85 else "HETATM", "id": str(i + 1), "type_symbol": sym,
                    "label_atom_id": an, "label_alt_id": [".", ".", ".", "A", "B"][rng.integers(5)], "label_comp_id": rn,
                    "label_asym_id": "A", "label_seq_id": seq, "pdbx_PDB_ins_code": ["?", "?", "?", "A"][rng.integers(4)],
                    "Cartn_x": "%.3f" % rng.uniform(-99, 99), "Cartn_y": "%.*f" % (int(rng.integers(0, 6)), rng.uniform(-999, 999)),
                    "Cartn_z": ["%.3f" % rng.uniform(-9, 9), "1.5e1", "-.25", "+3."][rng.integers(4)],
                    "occupancy": "1.00", "B_iso_or_equiv": "%.2f" % rng.uniform(0, 99), "auth_seq_id": seq, "auth_comp_id": rn,
                    "auth_asym_id": ["A", "B", "AA", "'C'"][rng.integers(4)], "auth_atom_id": an,
                    "pdbx_PDB_model_num": str(1 + (i * 3 // (n + 1)) if rng.random() < 0.9 else 1)}
            row = "".join(vals[cols[k]] + ws() for k in order)
            out.append(row if row.endswith("\n") or rng.random() < 0.3 else row + "\n")
        tail = ["", "#\n", "loop_\n_x.a\n_x.b\n1 2\n3 4\n", "_after.tag value\n", "\n;text\nfield\n;\n", "data_next\n"][rng.integers(6)]
        out.append(tail)
        out.append(" " * int(rng.integers(0, 70)))
        return "".join(out)

    texts = [make_text() for _ in range(400)]
    for options in (0, ingest.INCLUDE_HETATM | ingest.INCLUDE_HYDROGEN, ingest.JOIN_MODELS):
        a = ingest.load_pdb_texts(texts, options=options, n_threads=3)
        b = scalar(texts, options)
        assert a.n_atoms == b.n_atoms and a.n_atoms > 1000
        for f in ("xyz", "radii", "atom_class", "atom_backbone", "atom_name_raw", "atom_symbol_raw", "offsets", "res_first",
                  "res_offsets", "res_ref", "res_name_raw", "res_number_raw", "res_chain_raw", "status"):
            assert np.array_equal(getattr(a, f), getattr(b, f)), (options, f)


BATCH_ARRAYS = ("xyz", "radii", "atom_class", "atom_backbone", "atom_name_raw", "atom_symbol_raw", "offsets", "res_first",
                "res_offsets", "res_ref", "res_name_raw", "res_number_raw", "res_chain_raw", "status")


def test_binary_cache_round_trip_and_refusals(tmp_path):
    """SURVEY 8(f) N1's binary cache: a saved batch comes back array for array (failed inputs and an empty batch
    included), and a file that is not exactly what save() wrote - truncated, extended, one flipped byte anywhere,
    another version, another byte order, inconsistent offsets behind a valid checksum - is refused, not half-loaded."""
    paths = [fixture("1ubq.pdb"), fixture("empty.pdb"), fixture("1ubq.cif"), "/nonexistent/x.pdb", fixture("3bkr.pdb"), fixture("1d3z.pdb")]
    b = ingest.load_pdb_files(paths, options=ingest.INCLUDE_HETATM, n_threads=2)
    f = tmp_path / "batch.fsab"
    b.save(f)
    back = ingest.load_cache(f)
    assert (back.n_structs, back.n_atoms, back.n_residues) == (b.n_structs, b.n_atoms, b.n_residues)
    for name in BATCH_ARRAYS:
        x, y = getattr(b, name), getattr(back, name)
        assert x.dtype == y.dtype and x.shape == y.shape and x.tobytes() == y.tobytes(), name
    assert back.select(0, "s, resn ala and name ca")[1].sum() == b.select(0, "s, resn ala and name ca")[1].sum() > 0
    assert not [p for p in os.listdir(tmp_path) if ".tmp" in p]            # written under a temporary name, renamed

    empty = ingest.load_pdb_files([])
    empty.save(tmp_path / "empty.fsab")
    e2 = ingest.load_cache(tmp_path / "empty.fsab")
    assert e2.n_structs == 0 and e2.n_atoms == 0 and e2.offsets.tolist() == [0]

    raw = f.read_bytes()
    assert len(raw) % 16 == 0 and raw[:8] == b"FSASABAT"

    def refused(data, code):
        g = tmp_path / "bad.fsab"
        g.write_bytes(data)
        with pytest.raises(RuntimeError, match=f"code {code}"):
            ingest.load_cache(g)

    refused(raw[:-16], ingest.EFORMAT)                       # truncated
    refused(raw[:100], ingest.EFORMAT)                       # not even a header
    refused(raw + b"\0" * 16, ingest.EFORMAT)                # extended
    refused(b"", ingest.EFORMAT)
    rng = np.random.default_rng(5)
    for pos in [8, 12, 16, 24, 40, 48, 130] + rng.integers(128, len(raw), 12).tolist():
        flipped = bytearray(raw); flipped[pos] ^= 0x40
        refused(bytes(flipped), ingest.EFORMAT)              # version, byte-order mark, counts, checksum, payload
    v1 = bytearray(raw); v1[8:12] = (1).to_bytes(4, "little")    # a file of format version 1 (rounds 2-4): told apart from damage
    refused(bytes(v1), ingest.EVERSION)
    with pytest.raises(RuntimeError, match=f"code {ingest.EIO}"):
        ingest.load_cache(tmp_path / "missing.fsab")
    with pytest.raises(RuntimeError, match=f"code {ingest.EIO}"):
        b.save(tmp_path / "no_such_dir" / "x.fsab")
    # a batch whose offsets do not add up is not written at all
    broken = ingest.load_pdb_files([fixture("1ubq.pdb")])
    broken.offsets[1] -= 1
    with pytest.raises(RuntimeError, match=f"code {ingest.EFORMAT}"):
        broken.save(tmp_path / "broken.fsab")
    assert not (tmp_path / "broken.fsab").exists()


_M64 = (1 << 64) - 1


def _mix_step(h, w):
    h = ((h ^ w) * 0x9E3779B97F4A7C15) & _M64
    return h ^ (h >> 29)


def _piece_checksum(piece):
    """ingest_cache.c, mix_piece: four interleaved multiply-xorshift lanes, 32 bytes a step, the tail zero-padded."""
    h = [0x243F6A8885A308D3, 0x13198A2E03707344, 0xA4093822299F31D0, 0x082EFA98EC4E6C89]
    for i in range(0, len(piece), 32):
        blk = piece[i:i + 32].ljust(32, b"\0")
        for q in range(4):
            h[q] = _mix_step(h[q], int.from_bytes(blk[8 * q:8 * q + 8], "little"))
    r = _mix_step(_mix_step(_mix_step(h[0], h[1]), h[2]), h[3])
    return ((r ^ len(piece)) * 0xC2B2AE3D27D4EB4F) & _M64


def _cache_table(payload_sections, S, N, R):
    """The piece table (one checksum per 1 MiB piece of every section) and the header's checksum over it (mix_table)."""
    table = [_piece_checksum(sec[o:o + (1 << 20)]) for sec in payload_sections for o in range(0, len(sec), 1 << 20)]
    h = _mix_step(_mix_step(_mix_step(0x452821E638D01377, S & 0xffffffff), N & _M64), R & _M64)
    for t in table:
        h = _mix_step(h, t)
    return table, ((h ^ len(table)) * 0xC2B2AE3D27D4EB4F) & _M64


def test_binary_cache_refuses_crafted_indices_behind_a_valid_checksum(tmp_path):
    """A cache file whose index arrays were rewritten AND whose checksum was recomputed to match: offsets that point
    outside their arrays (monotone, right end value — the validator used to index res_first with them), a residue
    reference row or an atom class outside its table.  Refused by load() and by save(), no out-of-bounds read
    (`make asan-test` runs this under AddressSanitizer)."""
    import struct
    b = ingest.load_pdb_files([fixture("1ubq.pdb"), fixture("3bkr.pdb")])
    f = tmp_path / "good.fsab"
    b.save(f)
    raw = bytearray(f.read_bytes())
    S, N, R = b.n_structs, b.n_atoms, b.n_residues
    lens = [8 * (S + 1), 8 * (S + 1), 4 * S, 24 * N, 8 * N, N, N, 4 * N, 2 * N, 8 * (R + 1), 2 * R, 4 * R, 6 * R, 4 * R]
    starts, pos = [], 128
    for ln in lens:
        starts.append(pos)
        pos += (ln + 15) & ~15
    n_pieces = sum((ln + (1 << 20) - 1) >> 20 for ln in lens)
    assert pos + ((8 * n_pieces + 15) & ~15) == len(raw)                 # the piece table follows the payload

    def crafted(section, offset, fmt, value):
        data = bytearray(raw)
        struct.pack_into("<" + fmt, data, starts[section] + offset, value)
        secs = [bytes(data[st:st + ln]) for st, ln in zip(starts, lens)]
        table, head_sum = _cache_table(secs, S, N, R)
        struct.pack_into("<%dQ" % len(table), data, pos, *table)
        struct.pack_into("<Q", data, 48, head_sum)      # header: magic 8, version + byte-order mark 8, counts 8 + 16, payload 8, then the checksum
        return bytes(data)

    assert ingest.load_cache(_write(tmp_path / "same.fsab", crafted(3, 0, "d", float(np.asarray(b.xyz).reshape(-1)[0])))).n_atoms == N   # the recipe itself is right
    for section, offset, fmt, value in [(1, 8, "q", 1 << 20),       # res_offsets = {0, 1 << 20, R}
                                        (0, 8, "q", 1 << 40),       # offsets beyond the atoms
                                        (1, 8, "q", -5),
                                        (10, 0, "h", 30000),        # res_ref beyond the reference table
                                        (10, 2, "h", -2),
                                        (5, 3, "B", 7),             # atom class
                                        (2, 0, "i", 99)]:           # status
        g = _write(tmp_path / "crafted.fsab", crafted(section, offset, fmt, value))
        with pytest.raises(RuntimeError, match=f"code {ingest.EFORMAT}"):
            ingest.load_cache(g)
    bad = ingest.load_pdb_files([fixture("1ubq.pdb"), fixture("3bkr.pdb")])
    bad.res_offsets[1] = 1 << 20
    with pytest.raises(RuntimeError, match=f"code {ingest.EFORMAT}"):
        bad.save(tmp_path / "bad.fsab")
    bad = ingest.load_pdb_files([fixture("1ubq.pdb")])
    bad.res_ref[0] = 20000
    with pytest.raises(RuntimeError, match=f"code {ingest.EFORMAT}"):
        bad.save(tmp_path / "bad.fsab")


def _write(path, data):
    path.write_bytes(data)
    return path


@pytest.mark.gpu
def test_sweep_from_the_binary_cache_equals_the_sweep_from_the_files(tmp_path):
    import freesasa_amd as fa
    names = ["1ubq.pdb", "3bkr.pdb", "1ubq.cif", "2jo4.pdb"] if os.path.exists(fixture("2jo4.pdb")) else ["1ubq.pdb", "3bkr.pdb", "1ubq.cif"]
    b = ingest.load_pdb_files([fixture(n) for n in names])
    b.save(tmp_path / "sweep.fsab")
    c = ingest.load_cache(tmp_path / "sweep.fsab")
    want, _, wtot = fa.calc_batch(b.xyz, b.radii, b.offsets, fa.LEE_RICHARDS, 1.4, 20)
    got, _, gtot = fa.calc_batch(c.xyz, c.radii, c.offsets, fa.LEE_RICHARDS, 1.4, 20)
    assert np.array_equal(want, got) and np.array_equal(wtot, gtot)
    assert np.array_equal(b.residue_sums(want), c.residue_sums(got))


def test_batches_in_reused_blocks_equal_fresh_ones(tmp_path):
    """Round 5: a batch's arrays lie in one block, and the blocks of freed batches serve the next batches (a sweep builds
    a ~40 MB batch every few milliseconds: fresh from malloc that was an mmap, ten thousand page faults and a munmap
    each time).  Batches of changing sizes, from the parser and from the binary cache, built and freed in a row and by
    several threads at once, must equal the first ones built - whatever block they landed in."""
    import threading
    names = [fixture(n) for n in ("1a0q.pdb", "1d3z.pdb", "1ubq.pdb", "2jo4.pdb", "3bkr.pdb", "1ubq.cif", "5dx9.cif")]
    big = [n for n in names for _ in range(12)]           # ~2e5 atoms: above the size from which blocks are kept
    sets = [big, names[:2], big[:50], names, big[:17], names[:1]]
    keys = ("xyz", "radii", "offsets", "atom_name_raw", "res_number_raw", "status", "res_first", "atom_class")
    fresh = []
    for s_ in sets:                                       # (the first pass takes every block from malloc)
        b = ingest.load_pdb_files(s_, n_threads=3)
        fresh.append({k: np.array(getattr(b, k)).copy() for k in keys})

    def same(b, ref):
        return all(np.array_equal(np.array(getattr(b, k)), ref[k]) for k in keys)
    for rep in range(3):                                  # now out of kept blocks, in another order
        for k in (0, 2, 1, 4, 3, 5):
            assert same(ingest.load_pdb_files(sets[k], n_threads=1 + (rep + k) % 4), fresh[k]), (rep, k)
    cache = str(tmp_path / "big.bin")
    ingest.load_pdb_files(big, n_threads=2).save(cache)
    bad = []

    def hammer(seed):
        for k in np.random.default_rng(seed).permutation(len(sets)):
            if not same(ingest.load_pdb_files(sets[k], n_threads=2), fresh[k]):
                bad.append((seed, int(k)))
            if not same(ingest.load_cache(cache, n_threads=2), fresh[0]):
                bad.append((seed, "cache"))
    th = [threading.Thread(target=hammer, args=(s_,)) for s_ in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not bad, bad
    # the kept blocks go back to the allocator on request (round-5 advisor: a long-lived process kept up to 3 GiB) ...
    ingest.load_pdb_files(big, n_threads=2)
    released = ingest.trim(0)
    assert released >= 1 << 20 and ingest.trim(0) == 0
    # ... and batches built afterwards are the same again
    assert same(ingest.load_pdb_files(big, n_threads=2), fresh[0])


def test_coordinate_columns_that_run_together_are_read_like_sscanf_reads_them():
    """The reference reads the coordinates with sscanf("%lf%lf%lf") on columns 31-54 (src/pdb.c:176-197): numbers are
    whitespace-delimited, so a field that neither begins with a blank nor with its sign continues the number before
    it.  The loader's fixed-column conversion (coords_8_3, round 5) may only take a section whose three fields are
    delimited; everything else must come out as sscanf's tokens do - emulated here with strtod's grammar."""
    import re
    num = re.compile(r"[ \t\n\v\f\r]*([+-]?(?:\d+\.?\d*|\.\d+)(?:[eE][+-]?\d+)?)")
    rng = np.random.default_rng(23)

    def field():
        kind = rng.integers(0, 7)
        if kind == 0: return f"{rng.uniform(-999, 9999):8.3f}"[:8]        # the standard form, sometimes without a blank in front
        if kind == 1: return f"{rng.uniform(1000, 9999):8.3f}"              # fills its field: runs into its neighbor
        if kind == 2: return f"{rng.uniform(-999, -100):8.3f}"              # begins with its sign
        if kind == 3: return f"{rng.uniform(-9, 99):8.2f}"                  # '.' in another column
        if kind == 4: return f"{rng.integers(0, 999):4d}.000"
        if kind == 5: return f"{rng.uniform(0, 99):8.3f}".replace(" ", "0", 1) if rng.integers(0, 2) else "  -0.000"
        return f"{rng.uniform(0, 9):8.4f}"
    lines, want, ok = [], [], []
    for k in range(3000):
        sec = field() + field() + field()
        assert len(sec) == 24
        vals, pos = [], 0
        for _ in range(3):
            m = num.match(sec, pos)
            if not m:
                break
            vals.append(float(m.group(1)))
            pos = m.end()
        lines.append(f"ATOM  {k % 99999:5d}  CA  ALA A{k % 9999:4d}    {sec}  1.00  0.00           C  ")
        ok.append(len(vals) == 3)
        want.append(vals if len(vals) == 3 else [0, 0, 0])
    assert sum(ok) > 2500 and any(not q for q in ok) or all(ok)
    b = ingest.load_pdb_texts([ln + "\n" for ln in lines])
    for k in range(len(lines)):
        if ok[k]:
            assert b.status[k] == ingest.OK, (k, lines[k])
            got = b.xyz[b.offsets[k]]
            assert np.array_equal(got, np.array(want[k])) and np.array_equal(np.signbit(got), np.signbit(np.array(want[k]))), (lines[k], got, want[k])
        else:
            assert b.status[k] != ingest.OK, (k, lines[k])


def test_mmcif_rows_by_template_equal_the_byte_at_a_time_tokenizer():
    """Round 5: rows of an _atom_site loop whose tokens start in the same columns as the row before are taken by template
    (cif_tpl_match: three bit masks of the row instead of 21 trips through the token scanner).  Aligned loops as the wwPDB
    writes them - and everything that must throw a row back to the scanner: a value that outgrows its column, a quoted
    value with a blank inside that shows the starts of two plain ones, quotes / '#' / ';' / '_' at a token start, keywords
    in value position (loop_, data_x, save_), comment lines and blank lines between rows, tabs, CRLF, rows split over two
    lines, a second loop right behind, the text ending inside the last row's stride - against the scalar twin of the same
    source (-DFREESASA_INGEST_NO_SIMD: no templates, no bit masks)."""
    import ctypes as C
    from freesasa_amd.ingest import _CBatch, Batch
    path = os.path.join(ROOT, "tests", "emu", "libingest_scalar.so")
    if not os.path.exists(path):
        pytest.skip("scalar twin not built (make emu)")
    S = C.CDLL(path)
    S.freesasa_ingest_pdb_texts.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_int, C.c_int, C.c_int, C.POINTER(_CBatch)]
    S.freesasa_ingest_free.argtypes = [C.POINTER(_CBatch)]
    S.freesasa_ingest_free.restype = None

    def scalar(texts, options):
        raw = [t.encode() for t in texts]
        arr = (C.c_char_p * len(raw))(*raw)
        lens = (C.c_size_t * len(raw))(*[len(t) for t in raw])
        cb = _CBatch()
        assert S.freesasa_ingest_pdb_texts(arr, lens, len(raw), options, 1, C.byref(cb)) == 0
        try:
            return Batch(cb)
        finally:
            S.freesasa_ingest_free(C.byref(cb))

    rng = np.random.default_rng(29)
    cols = ["group_PDB", "id", "type_symbol", "label_atom_id", "label_alt_id", "label_comp_id", "label_asym_id", "label_entity_id",
            "label_seq_id", "pdbx_PDB_ins_code", "Cartn_x", "Cartn_y", "Cartn_z", "occupancy", "B_iso_or_equiv", "pdbx_formal_charge",
            "auth_seq_id", "auth_comp_id", "auth_asym_id", "auth_atom_id", "pdbx_PDB_model_num"]
    widths = {"group_PDB": 6, "id": 5, "type_symbol": 2, "label_atom_id": 4, "label_alt_id": 1, "label_comp_id": 3, "label_asym_id": 2,
              "label_entity_id": 1, "label_seq_id": 4, "pdbx_PDB_ins_code": 1, "Cartn_x": 7, "Cartn_y": 7, "Cartn_z": 7, "occupancy": 4,
              "B_iso_or_equiv": 6, "pdbx_formal_charge": 1, "auth_seq_id": 4, "auth_comp_id": 3, "auth_asym_id": 2, "auth_atom_id": 4,
              "pdbx_PDB_model_num": 1}
    res = ["ALA", "LEU", "SER", "ASP", "GLY", "DA", "HOH", "MSE"]
    atoms = ["N", "CA", "C", "O", "CB", "CG", "OXT", "SE", "H", "OD1"]
    odd = ["\"C1'\"", "'O P'", "\"N A\"", "_x", "#c", ";t", "loop_", "data_x", "save_", "'C'", "stop_", "AB_DE"]

    def make_text():
        order = list(range(len(cols))) if rng.random() < 0.6 else list(rng.permutation(len(cols)))
        out = ["data_fuzz\n", "#\n", "_cell.length_a 10.0\n", "loop_\n"]
        out += ["_atom_site.%s\n" % cols[k] for k in order]
        n = int(rng.integers(3, 90))
        eol = "\r\n" if rng.random() < 0.15 else "\n"
        sep = "\t" if rng.random() < 0.1 else " "
        for i in range(n):
            rn, an = res[rng.integers(len(res))], atoms[rng.integers(len(atoms))]
            sym = an[0] if an != "SE" else "SE"
            seq = str(int(rng.integers(1, 400)))
            vals = {"group_PDB": "ATOM" if rng.random() < 0.9 else "HETATM", "id": str(i + 1), "type_symbol": sym, "label_atom_id": an,
                    "label_alt_id": [".", ".", ".", "A", "B"][rng.integers(5)], "label_comp_id": rn, "label_asym_id": "A",
                    "label_entity_id": "1", "label_seq_id": seq, "pdbx_PDB_ins_code": ["?", "?", "?", "A"][rng.integers(4)],
                    "Cartn_x": "%.3f" % rng.uniform(-99, 99), "Cartn_y": "%.3f" % rng.uniform(-99, 99), "Cartn_z": "%.3f" % rng.uniform(-99, 99),
                    "occupancy": "1.00", "B_iso_or_equiv": "%.2f" % rng.uniform(0, 99), "pdbx_formal_charge": "?", "auth_seq_id": seq,
                    "auth_comp_id": rn, "auth_asym_id": ["A", "B", "AA"][rng.integers(3)], "auth_atom_id": an,
                    "pdbx_PDB_model_num": "1" if rng.random() < 0.95 else "2"}
            u = rng.random()
            if u < 0.04:
                vals["Cartn_y"] = "%.5f" % rng.uniform(-9999, 9999)                    # outgrows its column
            elif u < 0.10:
                vals[["auth_atom_id", "label_atom_id", "auth_asym_id", "label_comp_id", "B_iso_or_equiv"][rng.integers(5)]] = odd[rng.integers(len(odd))]
            row = sep.join(vals[cols[k]].ljust(widths[cols[k]]) for k in order) + sep
            u = rng.random()
            if u < 0.03:
                cut_at = row.find(sep, len(row) // 2)
                row = row[:cut_at] + eol + row[cut_at + 1:]                             # a row over two lines
            out.append(row + eol)
            u = rng.random()
            if u < 0.03: out.append("# a comment between rows" + eol)
            elif u < 0.05: out.append(eol)
        tail = ["", "#\n", "loop_\n_x.a\n_x.b\n1 2\n3 4\n", "_after.tag value\n", "\n;text\nfield\n;\n", "data_next\n", "loop_\n_atom_site.group_PDB\nATOM\n"][rng.integers(7)]
        out.append(tail)
        text = "".join(out)
        if rng.random() < 0.3:
            text = text[:len(text) - int(rng.integers(0, 150))]                          # ends somewhere in the last rows
        else:
            text += " " * int(rng.integers(0, 200))
        return text

    texts = [make_text() for _ in range(500)]
    for options in (0, ingest.INCLUDE_HETATM | ingest.INCLUDE_HYDROGEN, ingest.JOIN_MODELS):
        a = ingest.load_pdb_texts(texts, options=options, n_threads=3)
        b = scalar(texts, options)
        assert a.n_atoms == b.n_atoms and a.n_atoms > 5000
        for f in BATCH_ARRAYS:
            assert np.array_equal(getattr(a, f), getattr(b, f)), (options, f)


def test_cif_locate_finds_the_atom_site_loop_or_hands_the_file_to_the_host_parser():
    """What the device-side parser (csrc/gpu_parse.hip) asks of the host per file: PDB or mmCIF, and for mmCIF in its everyday
    form - one data block whose _atom_site category is one loop, one tag per header line - where the rows begin and which of
    the loop's columns are the twelve wanted ones.  Everything else (pair form, a second block first, an incomplete loop)
    must come back as 'host'.  The rows are NOT looked at here: oddities inside them are the device's to refuse."""
    import ctypes as C
    L = ingest._proto()
    L.freesasa_ingest_cif_locate.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_byte), C.POINTER(C.c_size_t)]

    def locate(text):
        ncol, slot, row0 = C.c_int(0), (C.c_byte * 12)(), C.c_size_t(0)
        kind = L.freesasa_ingest_cif_locate(text, len(text), C.byref(ncol), slot, C.byref(row0))
        return kind, ncol.value, list(slot), row0.value

    for name in ("1ubq.pdb", "empty.pdb", "syn_crlf.pdb"):
        assert locate(open(fixture(name), "rb").read())[0] == 0, name
    cols = ["group_PDB", "auth_asym_id", "auth_seq_id", "pdbx_PDB_ins_code", "auth_comp_id", "auth_atom_id", "label_alt_id", "type_symbol",
            "Cartn_x", "Cartn_y", "Cartn_z", "pdbx_PDB_model_num"]
    for name in ("1ubq.cif", "3bkr.cif", "5dx9.cif", "7cma-assembly1.cif", "syn_basic.cif", "syn_reordered_columns.cif", "syn_two_blocks_textfield.cif"):
        text = open(fixture(name), "rb").read()
        kind, ncol, slot, row0 = locate(text)
        assert kind == 1, name
        # the header as the file spells it: tag k of the loop is on line k behind "loop_"
        head = text[:row0].decode().split("\n")
        start = max(i for i, l in enumerate(head) if l.strip().lower() == "loop_")
        tags = [l.strip() for l in head[start + 1:] if l.strip().startswith("_")]
        assert len(tags) == ncol, name
        for k, c in enumerate(cols):
            assert tags[slot[k]].lower() == ("_atom_site." + c).lower(), (name, c)
        assert text[row0 - 1:row0] == b"\n" and not text[row0:row0 + 1] in (b"_", b"#"), name     # the first row's line
    for name in ("syn_pair_form.cif", "syn_pair_form_incomplete.cif", "syn_missing_column.cif", "syn_no_atoms.cif"):
        assert locate(open(fixture(name), "rb").read())[0] == 2, name
    # a text field that talks about loop_ and _atom_site.group_PDB before the real loop is stepped over
    text = b"data_X\n_struct.title\n;loop_\n_atom_site.group_PDB\n;\n" + open(fixture("syn_basic.cif"), "rb").read().split(b"\n", 1)[1]
    kind, ncol, slot, row0 = locate(text)
    assert kind == 1 and text[row0:row0 + 4] == b"ATOM"
