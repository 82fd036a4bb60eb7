"""GPU plumbing test of BASELINE configs[0]: the REFERENCE'S OWN CLI and library (structure/PDB
parsing, ProtOr classifier, result tree, log/PDB/sequence writers — all the reference's code,
compiled where it lies by `make -C oracle dropin`) running on top of the MI355X seam
(libfreesasa_amd_seam.a in place of nb.o, sasa_lr.o, sasa_sr.o).  The binary is a prebuilt,
git-ignored artefact (oracle/_ref/freesasa_dropin); the test is skipped where it was not built.
Assertions are the reference's own CLI checks (tests/test-cli.in:141-165, 214-215, 296-306)."""
import os
import re
import subprocess

import pytest

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu
CLI = os.path.join(ROOT, "oracle", "_ref", "freesasa_dropin")
PDB = os.path.join(GOLDEN, "1ubq.pdb")


def run(args, stdin=None, check=True):
    p = subprocess.run([CLI] + args, stdin=stdin, capture_output=True, text=True, timeout=120)
    if check:
        assert p.returncode == 0, p.stderr[-2000:]
    return p


@pytest.fixture(scope="module", autouse=True)
def _need_cli():
    if not os.path.exists(CLI):
        pytest.skip("oracle/_ref/freesasa_dropin not built (needs /root/reference at build time)")


def test_default_run_is_lee_richards_4804():
    out = run([PDB]).stdout
    assert re.search(r"atoms\s+: 602", out) and re.search(r"algorithm\s+: Lee & Richards", out)
    assert re.search(r"Total\s+:\s+4804.06", out)      # tests/test_freesasa.c:175 via the CLI
    assert re.search(r"Apolar\s+:\s+2299.84", out) and re.search(r"Polar\s+:\s+2504.22", out)


def test_shrake_rupley_totals():
    out = run(["-S", PDB]).stdout
    assert re.search(r"Total\s+:\s+4834.72", out)      # tests/test-cli.in:148-150
    assert re.search(r"Polar\s+:\s+2515.82", out) and re.search(r"Apolar\s+:\s+2318.90", out)


def test_per_atom_pdb_output_is_byte_identical_to_the_references_golden_file():
    with open(PDB) as fh:
        out = run(["-S", "--format=pdb"], stdin=fh).stdout
    got = "".join(l + "\n" for l in out.splitlines() if "REMARK" not in l)
    assert got == open(os.path.join(GOLDEN, "1ubq.B.pdb")).read()     # tests/test-cli.in:302-303


def test_per_residue_sequence_output_matches_reference_file():
    with open(PDB) as fh:
        out = run(["-S", "-R"], stdin=fh).stdout
    assert out == open(os.path.join(GOLDEN, "seq.reference")).read()   # tests/test-cli.in:298-299


def test_cli_rejects_what_the_reference_rejects():
    for bad in (["-S", "-n", "0"], ["-S", "-n", "-1"], ["-S", "-t", "1000"], ["-L", "-n", "0"]):
        with open(PDB) as fh:
            assert run(bad, stdin=fh, check=False).returncode != 0      # tests/test-cli.in:162-164
    with open(PDB) as fh:
        assert run(["-S", "-t", "16"], stdin=fh).returncode == 0
