import sys, os, glob, time, ctypes as C
sys.path.insert(0, '.')
import numpy as np
import freesasa_amd as fa
from freesasa_amd import ingest
paths = [p for p in sorted(glob.glob('tests/golden/pdb/*.pdb')) if os.path.getsize(p) > 10000] * 130
L = ingest._proto()
for nt in (16, 32, 64):
    for rep in range(3):
        arr = (C.c_char_p * len(paths))(*[p.encode() for p in paths]); cb = ingest._CBatch()
        t0=time.perf_counter(); L.freesasa_ingest_pdb_files(arr, len(paths), 0, nt, C.byref(cb)); t1=time.perf_counter()
        b = ingest.Batch(cb); t2=time.perf_counter(); L.freesasa_ingest_free(C.byref(cb))
    print('threads %d: C loader %.1f ms (%.0f M atoms/s), numpy copy %.1f ms, atoms %d' % (nt, 1e3*(t1-t0), b.n_atoms/(t1-t0)/1e6, 1e3*(t2-t1), b.n_atoms))
fa.calc_batch(b.xyz[:3000], b.radii[:3000], [0,3000])
for rep in range(3):
    t0=time.perf_counter(); fa.calc_batch(b.xyz, b.radii, b.offsets, fa.LEE_RICHARDS, resolution=20); t1=time.perf_counter()
print('calc_batch (H2D + kernels + D2H from pageable numpy): %.1f ms -> %.0f M atoms/s' % (1e3*(t1-t0), b.n_atoms/(t1-t0)/1e6))
