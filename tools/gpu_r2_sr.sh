# round 2: S&R evidence — kernel stats and SQ counters of k_sr_tile on the one-structure 200k-atom workload
export PYTHONUNBUFFERED=1
REPO=$(pwd); TAG=${1:-r02}
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
BENCH="python $REPO/bench.py --workload globule_sr --steps 20 --warmup 3"
(timeout 300 $BENCH) > $REPO/gpurun_out/sr_bench_$TAG.json 2> $REPO/gpurun_out/sr_bench_$TAG.err
(timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_sr_$TAG -o trace -- $BENCH) > $REPO/gpurun_out/sr_trace.log 2>&1
PM="python $REPO/bench.py --workload globule_sr --steps 2 --warmup 1"
(timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU --output-format csv -d $REPO/gpurun_out/prof_sr_$TAG -o pmc1 -- $PM) > $REPO/gpurun_out/sr_pmc1.log 2>&1
(timeout 300 rocprofv3 --kernel-trace --pmc VALUBusy VALUUtilization SALUBusy --output-format csv -d $REPO/gpurun_out/prof_sr_$TAG -o pmc2 -- $PM) > $REPO/gpurun_out/sr_pmc2.log 2>&1
(timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $REPO/gpurun_out/prof_sr_$TAG -o pmc3 -- $PM) > $REPO/gpurun_out/sr_pmc3.log 2>&1
(timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $REPO/gpurun_out/prof_sr_$TAG -o pmc4 -- $PM) > $REPO/gpurun_out/sr_pmc4.log 2>&1
cd $REPO
cat gpurun_out/sr_bench_$TAG.json | cut -c1-600
python tools/pmc_summary.py gpurun_out/prof_sr_$TAG all | tee gpurun_out/sr_pmc_$TAG.txt | grep "sr_tile" | cut -c1-500
cat gpurun_out/prof_sr_$TAG/trace_kernel_stats.csv | head -12
