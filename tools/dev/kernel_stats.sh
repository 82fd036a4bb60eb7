export PYTHONUNBUFFERED=1
R=$PWD; export TMPDIR=/tmp; cd /tmp
BENCH="python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-end-to-end --no-secondary --no-neighbors --no-live-counters --no-drivers --sustain-seconds 0"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ks -o trace -- $BENCH > /dev/null 2>&1
cut -d, -f1-6 $R/gpurun_out/ks/*/trace_kernel_stats.csv 2>/dev/null | head -12 || find $R/gpurun_out/ks -name "*stats*"
