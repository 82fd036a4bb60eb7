"""Kernel-time ablation of the S&R kernel (100 globules x 10k atoms and one 200k-atom globule)."""
import os, subprocess, sys
code = r'''
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import freesasa_amd as fa, tools
dev = torch.device('cuda:0')
ctx = fa.GpuContext(0, timing=True)
for label, parts in (("100x10k", [tools.globule(10000, 500 + k) for k in range(100)]), ("1x200k", [tools.globule(200000, 77)])):
    xyz = np.concatenate([p[0] for p in parts]); r = np.concatenate([p[1] for p in parts])
    offs = np.concatenate([[0], np.cumsum([len(p[1]) for p in parts])]).astype(np.int64)
    dx, dr = torch.from_numpy(xyz).to(dev), torch.from_numpy(r).to(dev)
    out = torch.empty(len(r), dtype=torch.float64, device=dev)
    for i in range(4): ctx.shrake_rupley(dx.data_ptr(), dr.data_ptr(), offs, out.data_ptr())
    st = ctx.stats(); print(label, 'kernel_ms %.3f prep_ms %.3f lds %d B %d TA %d' % (st['ms_kernel'], st['ms_prep'], st['lds_bytes'], st['block_threads'], st['tile_atoms']))
'''
for lib in sys.argv[1:]:
    env = dict(os.environ, FREESASA_AMD_LIB=lib)
    out = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True).stdout.strip().splitlines()
    print(lib.split('/')[-1], ' | '.join(out[-2:]))
