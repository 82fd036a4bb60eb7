"""The reference's selection language on loaded batches (include/freesasa_ingest.h)
against freesasa_select_area() of the reference library: 945 vectors minted by
tests/golden/make_select_golden.py — every command of the reference's tests/test_selection.c and
more, on seven structures.  The reference was handed seeded random per-atom weights as "SASA", so
an equal area (bit for bit) means an equal atom set."""
import json
import os

import numpy as np
import pytest

from conftest import ROOT

from freesasa_amd import ingest

with open(os.path.join(ROOT, "tests", "golden", "select.json")) as fh:
    GOLD = json.load(fh)


def _load(g):
    path = os.path.join(ROOT, "tests", "golden", "cif" if g["file"].endswith(".cif") else "pdb", g["file"])
    b = ingest.load_pdb_files([path], options=g["options"])
    assert b.n_atoms == g["n_atoms"]
    return b, np.random.default_rng(g["seed"]).uniform(0.0, 100.0, b.n_atoms)


@pytest.mark.parametrize("k", range(len(GOLD)), ids=[f"{g['file']}-{g['options']}" for g in GOLD])
def test_selections_match_the_reference(k):
    g = GOLD[k]
    b, w = _load(g)
    n_fail = n_warn = 0
    for r in g["selections"]:
        try:
            name, mask, warned = b.select(0, r["command"])
        except ValueError:
            assert r["rc"] == -1, r["command"]                 # FREESASA_FAIL: syntax error
            n_fail += 1
            continue
        assert (-2 if warned else 0) == r["rc"], r["command"]
        n_warn += warned
        assert name == r["name"], r["command"]
        area = 0.0
        for j in np.nonzero(mask)[0]:                          # sequential, like src/selection.c:717-720
            area += w[j]
        assert area == float.fromhex(r["area"]), r["command"]
    assert n_fail > 20 and n_warn > 5


def test_selection_of_a_later_structure_in_a_batch():
    paths = [os.path.join(ROOT, "tests", "golden", "pdb", n) for n in ("icode.pdb", "1ubq.pdb", "alt_model_twochain.pdb")]
    b = ingest.load_pdb_files(paths)
    alone = ingest.load_pdb_files(paths[1:2])
    for cmd in ("s, resn lys and not name n+ca+c+o", "s, resi 10-20 or symbol s", "s, chain A-B"):
        for k, ref in ((1, alone),):
            name, mask, _ = b.select(k, cmd)
            _, want, _ = ref.select(0, cmd)
            assert np.array_equal(mask, want) and len(mask) == b.offsets[k + 1] - b.offsets[k]
    name, mask, _ = b.select(2, "two, chain B")
    assert name == "two" and 0 < mask.sum() < len(mask)
    with pytest.raises(ValueError):
        b.select(0, "no comma here")


@pytest.mark.gpu
def test_selection_areas_on_the_device():
    """Selection area = masked sum of per-atom SASA: the class-sum kernel with the mask as class."""
    import torch
    import freesasa_amd as fa
    b = ingest.load_pdb_files([os.path.join(ROOT, "tests", "golden", "pdb", "1ubq.pdb")])
    sasa, _, tot = fa.calc_batch(b.xyz, b.radii, b.offsets, fa.SHRAKE_RUPLEY, resolution=100)
    dev = torch.device("cuda:0")
    d_sasa = torch.from_numpy(sasa).to(dev)
    ctx = fa.GpuContext(0)
    for cmd in ("bb, name n+ca+c+o", "hyd, resn ala+val+leu+ile+met+phe+trp+pro", "r, resi 1-38 and not symbol c"):
        name, mask, _ = b.select(0, cmd)
        d_mask = torch.from_numpy(mask).to(dev)
        d_out = torch.empty(3, dtype=torch.float64, device=dev)
        ctx.class_sums(d_sasa.data_ptr(), d_mask.data_ptr(), b.offsets, d_out.data_ptr())
        out = d_out.cpu().numpy()
        assert abs(out[1] - sasa[mask == 1].sum()) < 1e-9 and abs(out[0] + out[1] - tot[0]) < 1e-9
    ctx.close()
