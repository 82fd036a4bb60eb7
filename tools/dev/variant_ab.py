"""A/B of library variants (tools/build_variant.sh) on protein-like density: L&R-20 and S&R-100 kernel times."""
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
res = []
for alg in ('lr', 'sr'):
    for i in range(5): (ctx.lee_richards if alg == 'lr' else ctx.shrake_rupley)(dx.data_ptr(), dr.data_ptr(), offs, out.data_ptr())
    res.append('%s %.3f' % (alg, ctx.stats()['ms_kernel']))
print(' '.join(res))
'''
for lib in sys.argv[1:]:
    env = dict(os.environ, FREESASA_AMD_LIB=lib)
    out = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True).stdout.strip().splitlines()
    print(os.path.basename(lib), out[-1] if out else 'FAILED')
