# minimal perf check: L&R headline (300 structs), S&R 200k globule, L&R on protein-like globules
export PYTHONUNBUFFERED=1
fmt() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('%-34s value %.4g  ms/step %.3f kernel_ms %.3f prep_ms %.3f fallback %d lds %d B %d TA %d' % (d['metric'], d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['prep_ms'], d['config']['fallback_tiles'], d['config']['lds_bytes_per_block'], d['config']['block_threads'], d['config']['tile_atoms']))
"; }
python bench.py --steps 3 --warmup 1 --structs 300 --no-cpu-baseline 2>&1 | fmt
python bench.py --workload globule_sr --steps 20 --warmup 3 2>&1 | fmt
python - <<'PY' 2>&1 | tail -3
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import freesasa_amd as fa, tools
# protein-like packing: 100 globules x 10k atoms, L&R 20
parts = [tools.globule(10000, 500 + k) for k in range(100)]
xyz = np.concatenate([p[0] for p in parts]); r = np.concatenate([p[1] for p in parts])
offs = np.arange(101, dtype=np.int64) * 10000
dev = torch.device('cuda:0')
dx, dr = torch.from_numpy(xyz).to(dev), torch.from_numpy(r).to(dev)
out = torch.empty(len(r), dtype=torch.float64, device=dev)
ctx = fa.GpuContext(0, timing=True)
for alg in ('lr', 'sr'):
    for i in range(4):
        t0 = time.perf_counter()
        if alg == 'lr': ctx.lee_richards(dx.data_ptr(), dr.data_ptr(), offs, out.data_ptr())
        else: ctx.shrake_rupley(dx.data_ptr(), dr.data_ptr(), offs, out.data_ptr())
        dt = time.perf_counter() - t0
    st = ctx.stats()
    print('globule100x10k %s: %.4g atoms/s  kernel_ms %.3f prep_ms %.3f fallback %d lds %d maxnn %d' % (alg, len(r)/dt, st['ms_kernel'], st['ms_prep'], st['fallback_tiles'], st['lds_bytes'], st['max_neighbors']))
PY
python - <<'PY' 2>&1 | tail -4
import sys, time, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import freesasa_amd as fa
g = np.load('tests/golden/1ubq.npz')
for alg, name in ((fa.LEE_RICHARDS, 'L&R-20'), (fa.SHRAKE_RUPLEY, 'S&R-100')):
    fa.calc_coord(g['xyz'], g['radii'], alg)
    t0 = time.perf_counter()
    for _ in range(50): fa.calc_coord(g['xyz'], g['radii'], alg)
    print('1UBQ freesasa_calc_coord %s: %.0f us per call (host arrays in, host result out)' % (name, (time.perf_counter() - t0) / 50 * 1e6))
PY
