"""L&R launch shapes on protein-like density (100 globules x 10k atoms): FREESASA_AMD_CFG sweep."""
import os, subprocess, sys
code = r'''
import sys, numpy as np, torch
sys.path.insert(0, '.')
import freesasa_amd as fa, tools
parts = [tools.globule(10000, 500 + k) for k in range(100)]
xyz = np.concatenate([p[0] for p in parts]); r = np.concatenate([p[1] for p in parts])
offs = np.arange(101, dtype=np.int64) * 10000
dev = torch.device('cuda:0')
dx, dr = torch.from_numpy(xyz).to(dev), torch.from_numpy(r).to(dev)
out = torch.empty(len(r), dtype=torch.float64, device=dev)
ctx = fa.GpuContext(0, timing=True)
for i in range(4): ctx.lee_richards(dx.data_ptr(), dr.data_ptr(), offs, out.data_ptr())
st = ctx.stats(); print('kernel_ms %.3f fallback %d lds %d B %d TA %d' % (st['ms_kernel'], st['fallback_tiles'], st['lds_bytes'], st['block_threads'], st['tile_atoms']))
'''
for cfg in sys.argv[1:]:
    env = dict(os.environ)
    if cfg != 'default': env['FREESASA_AMD_CFG'] = cfg
    out = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True).stdout.strip().splitlines()
    print(cfg, out[-1] if out else 'FAILED')
