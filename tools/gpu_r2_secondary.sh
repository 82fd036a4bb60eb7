# round 2: secondary workloads (one line each)
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out
fmt() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); c=d.get('config',{}); r=d.get('roofline',{})
        print('%-46s value %.4g  ms/step %.3f kernel_ms %s prep_ms %s fallback %s lds %s TA %s' % (d['metric'], d['value'], d['ms_per_step'], r.get('kernel_ms'), r.get('prep_ms'), c.get('fallback_tiles'), c.get('lds_bytes_per_block'), c.get('tile_atoms')))
"; }
: > gpurun_out/secondary_bench.jsonl
for args in "--workload globule_sr --steps 20 --warmup 3" "--slices 100 --structs 200 --no-cpu-baseline --no-end-to-end" "--workload sweep_lr --no-cpu-baseline" "--workload traj_lr --steps 8"; do
  timeout 400 python bench.py $args 2>/dev/null | tail -1 | tee -a gpurun_out/secondary_bench.jsonl | fmt
done
python - <<'PY' 2>&1 | tail -4
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import freesasa_amd as fa, tools
parts = [tools.globule(10000, 500 + k) for k in range(100)]
xyz = np.concatenate([p[0] for p in parts]); r = np.concatenate([p[1] for p in parts])
offs = np.arange(101, dtype=np.int64) * 10000
dev = torch.device('cuda:0')
dx, dr = torch.from_numpy(xyz).to(dev), torch.from_numpy(r).to(dev)
out = torch.empty(len(r), dtype=torch.float64, device=dev)
ctx = fa.GpuContext(0, timing=True)
for alg in ('lr', 'sr'):
    best = 1e9
    for i in range(5):
        t0 = time.perf_counter()
        if alg == 'lr': ctx.lee_richards(dx.data_ptr(), dr.data_ptr(), offs, out.data_ptr())
        else: ctx.shrake_rupley(dx.data_ptr(), dr.data_ptr(), offs, out.data_ptr())
        best = min(best, time.perf_counter() - t0)
    st = ctx.stats()
    print('globule100x10k %s: %.4g atoms/s  kernel_ms %.3f prep_ms %.3f fallback %d lds %d TA %d maxnn %d' % (alg, len(r)/best, st['ms_kernel'], st['ms_prep'], st['fallback_tiles'], st['lds_bytes'], st['tile_atoms'], st['max_neighbors']))
g = np.load('tests/golden/1ubq.npz')
gx, gr = g['xyz'], g['radii']   # (an NpzFile reads the archive member again on every access)
for alg, name in ((fa.LEE_RICHARDS, 'L&R-20'), (fa.SHRAKE_RUPLEY, 'S&R-100')):
    fa.calc_coord(gx, gr, alg)
    t0 = time.perf_counter()
    for _ in range(50): fa.calc_coord(gx, gr, alg)
    print('1UBQ freesasa_calc_coord %s: %.0f us per call' % (name, (time.perf_counter() - t0) / 50 * 1e6))
PY
