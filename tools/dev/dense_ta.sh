# DEV: what the chooser picks on the L&R workloads, and the kernel time (FREESASA_AMD_LR2="TA,pool,ds,refill" forces a shape)
export PYTHONUNBUFFERED=1
export FREESASA_AMD_SHOW_SHAPE=1
for wl in pdb_lr coil_lr; do
  python bench.py --workload $wl --steps 8 --warmup 4 --sustain-seconds 0 --no-cpu-baseline --no-end-to-end --no-secondary --no-neighbors --no-live-counters --no-drivers 2> /tmp/err.txt | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$wl kernel_ms %.4f step %.4f' % (d['roofline']['kernel_ms'], d['ms_per_step']))"
  grep "lr2 shape" /tmp/err.txt | cut -c1-60 | uniq -c | tail -4
done
python tools/gpu_shapes.py g100 "0,0,-1,0" > /tmp/out.txt 2> /tmp/err.txt; echo "globules: $(grep kernel_ms /tmp/out.txt /tmp/err.txt | tail -1 | cut -c1-140)"
grep "lr2 shape" /tmp/err.txt /tmp/out.txt | cut -c1-80 | uniq -c | tail -4
