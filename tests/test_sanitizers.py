"""The host C sources (PDB / mmCIF readers, selection language, C-API shims, test-point generator) under
AddressSanitizer + UndefinedBehaviorSanitizer: `make asan-test` builds them instrumented (tests/emu/
libfreesasa_amd_asan.so, with the ordinary engine object) and runs the CPU suites that drive them — the reference
vectors of the readers and the selection language, the C-API layout and error-path tests.  SURVEY.md section 5."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_sources_under_asan_and_ubsan():
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan):
        pytest.skip("this gcc has no libasan")
    res = subprocess.run(["make", "-C", ROOT, "asan-test"], capture_output=True, text=True, timeout=900)
    tail = (res.stdout + res.stderr)[-3000:]
    assert res.returncode == 0, tail
    assert "passed" in tail and "ERROR: AddressSanitizer" not in tail and "runtime error" not in tail, tail
