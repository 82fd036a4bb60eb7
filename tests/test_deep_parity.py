"""A 1e6-atom slice of tools/deep_parity.py inside the suite the driver runs (round-5 review, hygiene 8b): per-atom
Lee-Richards 20 / 100 slices and Shrake-Rupley 100 points of 100 synthetic structures x 10 000 atoms - random coils
(the headline workload's kind) and lattice globules (protein density) - from the GPU engine against the REAL reference
library (oracle/_ref/libfreesasa_ref.so: the reference's own sources compiled where they lie; the plain-C oracle where
that artefact is missing), computed on the box's host cores.  Bars: LR_TOL = 1e-8 A^2 per atom (north_star: 1e-4), and
no atom's S&R area different at all.  What the sampled checks of test_gpu_parity.py (5 of 1000 structures) cannot
show, this does: every atom of the slice is compared."""
import concurrent.futures as cf

import numpy as np
import pytest

import tools

pytestmark = pytest.mark.gpu

LR_TOL = 1e-8
N_COILS, N_GLOBULES, ATOMS = 60, 40, 10_000


def _checker():
    import oracle
    if oracle.Reference.available():
        ref = oracle.Reference()
        ref.lib.freesasa_set_verbosity(2)
        lr = lambda x, r, ns: ref.calc_coord(x, r, oracle.LEE_RICHARDS, 1.4, n_slices=ns, n_threads=1)[0]
        sr = lambda x, r: ref.calc_coord(x, r, oracle.SHRAKE_RUPLEY, 1.4, n_points=100, n_threads=1)[0]
        return lr, sr, "reference"
    orc = oracle.Oracle()
    return (lambda x, r, ns: orc.lee_richards(x, r, 1.4, ns)), (lambda x, r: orc.shrake_rupley(x, r, 1.4, 100)[0]), "port"


def test_one_million_atoms_against_the_reference():
    import freesasa_amd as fa
    assert fa.device_count() > 0
    parts = [tools.coil(ATOMS, 9000 + k) for k in range(N_COILS)] + [tools.globule(ATOMS, 1700 + k) for k in range(N_GLOBULES)]
    xyz = np.concatenate([p[0] for p in parts]); r = np.concatenate([p[1] for p in parts])
    offs = np.arange(len(parts) + 1, dtype=np.int64) * ATOMS
    lr20, _, _ = fa.calc_batch(xyz, r, offs, fa.LEE_RICHARDS, 1.4, 20)
    lr100, _, _ = fa.calc_batch(xyz, r, offs, fa.LEE_RICHARDS, 1.4, 100)
    sr100, _, _ = fa.calc_batch(xyz, r, offs, fa.SHRAKE_RUPLEY, 1.4, 100)
    lr, sr, kind = _checker()
    every4 = list(range(0, len(parts), 4))           # L&R 100 costs the CPU five times L&R 20: a quarter of the structures

    def job(task):
        what, k = task
        x, rr = parts[k]
        sl = slice(int(offs[k]), int(offs[k + 1]))
        if what == "lr20":
            return what, float(np.max(np.abs(lr20[sl] - lr(x, rr, 20))))
        if what == "lr100":
            return what, float(np.max(np.abs(lr100[sl] - lr(x, rr, 100))))
        return what, int(np.count_nonzero(sr100[sl] != sr(x, rr)))

    tasks = [("lr100", k) for k in every4] + [("lr20", k) for k in range(len(parts))] + [("sr", k) for k in range(len(parts))]
    worst = {"lr20": 0.0, "lr100": 0.0, "sr": 0}
    with cf.ThreadPoolExecutor(max_workers=max(1, tools.usable_cpus())) as pool:   # (the checker's C code runs outside the GIL)
        for what, v in pool.map(job, tasks):
            worst[what] = max(worst[what], v) if what != "sr" else worst[what] + v
    print(f"\ndeep parity [{kind}]: {len(parts) * ATOMS} atoms: L&R-20 max|dSASA| {worst['lr20']:.3e} A^2, "
          f"L&R-100 ({len(every4) * ATOMS} atoms) {worst['lr100']:.3e} A^2, S&R-100 atoms with a different area: {worst['sr']}")
    assert worst["lr20"] < LR_TOL and worst["lr100"] < LR_TOL, worst
    assert worst["sr"] == 0
