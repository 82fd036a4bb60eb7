"""The PDB / mmCIF parser ON THE DEVICE (freesasa_amd/csrc/gpu_parse.hip; round-5 review, item 2): host threads read bytes,
kernels find the lines, filter the records (ATOM / HETATM, hydrogens, first model, alternate locations), convert the
coordinates, classify the atoms (ProtOr radii and classes, element fallback) - what the reference's readers do for one
file on one core (src/structure.c:644-722, src/pdb.c:176-283, src/cif.cc:113-240, src/classifier.c:781-796,1002-1017).

The bar is the reference-minted vectors of the host loader (tests/golden/ingest.json, made by make_ingest_golden.py from
the REAL reference library): for every fixture file and every option set, what the device keeps must have the
reference's coordinate / radius / class digests - or the device must refuse the file (which then goes to the host
parser), and the files of the everyday forms must NOT be refused.  The sweep with the device parser must give the host
sweep's results bit for bit, refusals included."""
import hashlib
import json
import os

import numpy as np
import pytest

from freesasa_amd import ingest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PDB = os.path.join(ROOT, "tests", "golden", "pdb")
CIF = os.path.join(ROOT, "tests", "golden", "cif")
with open(os.path.join(ROOT, "tests", "golden", "ingest.json")) as fh:
    GOLD = json.load(fh)


def fixture(name):
    return os.path.join(CIF if name.endswith(".cif") else PDB, name)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def fa():
    import freesasa_amd
    assert freesasa_amd.device_count() > 0
    return freesasa_amd


# files the device is EXPECTED to parse itself under the default options (everything of an everyday form)
MUST_PARSE = {"1ubq.pdb", "1a0q.pdb", "3bkr.pdb", "3bzd_trimmed.pdb", "5dx9.pdb", "1d3z.pdb", "2jo4.pdb", "3gnn.pdb", "icode.pdb",
              "alt_model_twochain.pdb", "1ubq.cif", "3bkr.cif", "5dx9.cif", "7cma-assembly1.cif",  # (syn_crlf.pdb holds exponents and run-together columns, syn_basic.cif holds -2.5e0 and +3. in a HETATM row: strtod's, the host's)
              "syn_altloc_icode_chain.pdb", "syn_altloc_icode_chains.cif", "syn_models_out_of_order.cif"}  # (syn_reordered_columns.cif: a row over three lines)


@pytest.mark.parametrize("name", sorted(GOLD))
def test_device_parser_against_the_reference_minted_vectors(fa, name):
    refused, parsed = 0, 0
    for opt, exp in GOLD[name].items():
        if exp.get("crash"):
            continue
        xyz, r, cls, offs, status, host = fa.parse_files_dev([fixture(name)], ingest_options=int(opt))
        if host[0]:
            refused += 1
            assert offs[1] == 0
            continue
        parsed += 1
        if exp.get("fail"):
            assert status[0] != ingest.OK and offs[1] == 0, (name, opt, status[0])
        else:
            assert status[0] == ingest.OK, (name, opt, status[0])
            assert offs[1] == exp["n_atoms"], (name, opt)
            assert sha(xyz) == exp["xyz"] and sha(r) == exp["radii"] and sha(cls) == exp["classes"], (name, opt)
    if name in MUST_PARSE:
        # (RADIUS_FROM_OCCUPANCY, option 256, is the host's: one refusal allowed)
        assert refused <= 1 and parsed >= 8, (name, refused, parsed)


def test_device_parser_equals_the_host_parser_file_by_file_in_one_batch(fa):
    """All fixtures in ONE batch (lines of different files side by side in the text), several option sets: per file the
    device's atoms are the host loader's, or the file is refused."""
    names = sorted(GOLD) + ["does_not_exist.pdb"]
    paths = [fixture(n) for n in names]
    for opt in (0, ingest.INCLUDE_HETATM | ingest.INCLUDE_HYDROGEN, ingest.JOIN_MODELS, ingest.SKIP_UNKNOWN, ingest.HALT_AT_UNKNOWN):
        want = ingest.load_files(paths, options=opt, n_threads=4)
        xyz, r, cls, offs, status, host = fa.parse_files_dev(paths, ingest_options=opt, n_threads=3)
        n_dev = 0
        for k, n in enumerate(names):
            if host[k]:
                assert offs[k + 1] == offs[k]
                continue
            n_dev += 1
            lo, hi = int(want.offsets[k]), int(want.offsets[k + 1])
            a, b = int(offs[k]), int(offs[k + 1])
            assert status[k] == want.status[k], (n, opt, status[k], want.status[k])
            assert b - a == hi - lo, (n, opt)
            assert np.array_equal(xyz[a:b], want.xyz[lo:hi]) and np.array_equal(r[a:b], want.radii[lo:hi]), (n, opt)
            assert np.array_equal(cls[a:b], want.atom_class[lo:hi]), (n, opt)
        assert n_dev >= 25, (opt, n_dev)                              # (the made-up fixtures are mostly oddities: a third is refused)
        assert host[names.index("does_not_exist.pdb")] == 1           # an unreadable file is the host's to report


def test_sweep_with_the_device_parser_is_the_host_sweep_bit_for_bit(fa, tmp_path):
    names = sorted(GOLD) * 3 + ["does_not_exist.pdb"]
    paths = [fixture(n) for n in names]
    fa.sweep_parse_stats()
    for alg, res in ((fa.LEE_RICHARDS, 20), (fa.SHRAKE_RUPLEY, 100)):
        want = fa.sweep_files(paths, alg=alg, resolution=res, n_threads=4, batch_atoms=4000, devices=[0, 0])
        got = fa.sweep_files(paths, alg=alg, resolution=res, n_threads=4, batch_atoms=4000, devices=[0, 0],
                             ingest_options=ingest.PARSE_ON_DEVICE)
        for a, b, what in zip(want, got, ("totals", "class sums", "atoms", "status")):
            assert np.array_equal(a, b), what
    dev, host = fa.sweep_parse_stats()
    assert dev > 2 * host > 0, (dev, host)        # most files were parsed on the device; the odd ones (a third of these fixtures) were handed over and counted
    # resumable form: the done-list of a device-parsed sweep is the host-parsed sweep's (who parses is not part of a run's name)
    ok, t1, c1, a1, s1 = fa.sweep_files_resumable(paths, tmp_path / "d.txt", batch_atoms=4000, ingest_options=ingest.PARSE_ON_DEVICE, max_new_batches=2)
    assert not ok
    ok, t2, c2, a2, s2 = fa.sweep_files_resumable(paths, tmp_path / "d.txt", batch_atoms=4000)
    assert ok and np.array_equal(t2, want[0] if False else fa.sweep_files(paths, batch_atoms=4000)[0])


def test_a_large_batch_of_real_entries(fa):
    """A few hundred copies of the reference's PDB entries and mmCIF files in one batch: ~7e5 lines, files of every size next
    to each other; every file parsed on the device, every atom the host loader's."""
    names = ["1a0q.pdb", "3bkr.cif", "1ubq.pdb", "5dx9.cif", "3gnn.pdb", "1ubq.cif", "2jo4.pdb", "5dx9.pdb", "7cma-assembly1.cif"]
    paths = [fixture(n) for n in names] * 40
    want = ingest.load_files(paths, n_threads=8)
    xyz, r, cls, offs, status, host = fa.parse_files_dev(paths, n_threads=8)
    assert not host.any() and np.array_equal(status, want.status)
    assert np.array_equal(offs, want.offsets)
    assert np.array_equal(xyz, want.xyz) and np.array_equal(r, want.radii) and np.array_equal(cls, want.atom_class)
