import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


def _hip_device_count():
    """HIP devices the product library sees (0 in the GPU-less build container, or when it does not load)."""
    try:
        import freesasa_amd
        return int(freesasa_amd.device_count())
    except Exception:
        return 0


def pytest_collection_modifyitems(config, items):
    """`pytest` without -m: the gpu-marked tests are skipped, not failed, where there is no HIP device."""
    if _hip_device_count() > 0:
        return
    skip = pytest.mark.skip(reason="no HIP device (the gpu-marked tests run on the MI355X box)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle_lib():
    import oracle
    oracle.build()
    return oracle.Oracle()


@pytest.fixture(scope="session")
def reference_lib():
    import oracle
    if not oracle.Reference.available():
        pytest.skip("oracle/_ref/libfreesasa_ref.so not built (needs /root/reference)")
    ref = oracle.Reference()
    ref.lib.freesasa_set_verbosity(2)
    return ref


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def read_bfactor_pdb(path):
    """xyz, radius (occupancy column) and SASA (B-factor column) of the reference's golden
    per-atom file tests/data/1ubq.B.pdb (fixed PDB columns)."""
    xyz, rad, sasa = [], [], []
    with open(path) as fh:
        for line in fh:
            if line.startswith("ATOM"):
                xyz.append([float(line[30:38]), float(line[38:46]), float(line[46:54])])
                rad.append(float(line[54:60]))
                sasa.append(float(line[60:66]))
    return np.array(xyz), np.array(rad), np.array(sasa)


def read_seq_reference():
    """(chain, residue number, residue name, area) per line of the reference's per-residue S&R
    output for 1UBQ (tests/data/seq.reference, tests/test-cli.in:298-299)."""
    rows = []
    with open(os.path.join(GOLDEN, "seq.reference")) as fh:
        for line in fh:
            if line.startswith("SEQ"):
                head, area = line.split(":")
                _, chain, number, name = head.split()
                rows.append((chain, number, name, float(area)))
    return rows
