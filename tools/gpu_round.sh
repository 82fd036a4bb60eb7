# Full GPU session for a round: tests, smoke, bench, rocprof kernel trace + PMC passes.
# Usage: bash tools/gpu_round.sh <tag>     (outputs under gpurun_out/, summaries to copy into profiles/)
TAG=${1:-r01}
set -x
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
REPO=$(pwd)
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()") > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
(timeout 900 python -m pytest tests -m gpu -q --durations=5) > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
(timeout 600 python bench.py) > gpurun_out/bench_$TAG.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench_$TAG.log
export TMPDIR=/tmp
cd /tmp
BENCH="python $REPO/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-end-to-end"
(timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_$TAG -o trace -- $BENCH) > $REPO/gpurun_out/rocprof_trace.log 2>&1
PM="python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-end-to-end"
(timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU --output-format csv -d $REPO/gpurun_out/prof_$TAG -o pmc1 -- $PM) > $REPO/gpurun_out/rocprof_pmc1.log 2>&1
(timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VALU --output-format csv -d $REPO/gpurun_out/prof_$TAG -o pmc2 -- $PM) > $REPO/gpurun_out/rocprof_pmc2.log 2>&1
(timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $REPO/gpurun_out/prof_$TAG -o pmc3 -- $PM) > $REPO/gpurun_out/rocprof_pmc3.log 2>&1
(timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $REPO/gpurun_out/prof_$TAG -o pmc4 -- $PM) > $REPO/gpurun_out/rocprof_pmc4.log 2>&1
cd $REPO
python tools/pmc_summary.py gpurun_out/prof_$TAG all | tee gpurun_out/pmc_$TAG.txt
cat gpurun_out/prof_$TAG/trace_kernel_stats.csv
tail -2 gpurun_out/smoke.log; tail -8 gpurun_out/pytest_gpu.log; tail -2 gpurun_out/bench_$TAG.log | cut -c1-3000
