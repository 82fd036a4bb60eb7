"""DEV: the giant-radius case of tests/test_adversarial.py under different launch shapes / sizes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import oracle
import freesasa_amd as fa
ref = oracle.Reference() if oracle.Reference.available() else None
orc = oracle.Oracle()
def case(n, r_small=0.1, box=(60, 31, 31)):
    rng = np.random.default_rng(3)
    xyz = np.vstack([rng.uniform([0, 0, 0], box, size=(n, 3)), [[260.0, 15.0, 15.0]]])
    r = np.append(np.full(n, r_small), 30.0)
    return xyz, r
for n in (2000, 8000, 16000, 20000):
    xyz, r = case(n)
    want = orc.lee_richards(xyz, r, 1.4, 20)
    for env in ({}, {"FREESASA_AMD_COVER": "0"}, {"FREESASA_AMD_LR2": "3,0,-1,0"}, {"FREESASA_AMD_LR2": "6,0,-1,0"}, {"FREESASA_AMD_LR1": "1"}):
        for k in ("FREESASA_AMD_COVER", "FREESASA_AMD_LR2", "FREESASA_AMD_LR1"): os.environ.pop(k, None)
        os.environ.update(env); os.environ["FREESASA_AMD_SHOW_SHAPE"] = "1"
        fa.lib().freesasa_gpu_release_pool()
        got, _, _ = fa.calc_batch(xyz, r, np.array([0, len(r)], dtype=np.int64), fa.LEE_RICHARDS, 1.4, 20)
        d = np.abs(got - want)
        print(f"n={n} env={env} max diff {np.nanmax(d):.3g} bad {int((d > 1e-8).sum())} nan {int(np.isnan(got).sum())}", flush=True)
