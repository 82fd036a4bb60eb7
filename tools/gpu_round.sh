# one GPU session: tests, bench, rocprof.  Usage: bash tools/gpu_round.sh [tag]
TAG=${1:-r01}
set -x
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
cd $GRAFT_REPO_ROOT 2>/dev/null || true
REPO=$(pwd)
(timeout 900 python -m pytest tests -m gpu -q --durations=5) > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
(timeout 400 python bench.py --steps 5 --warmup 1) > gpurun_out/bench.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench.log
export TMPDIR=/tmp
cd /tmp
(timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_$TAG -o trace -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline) > $REPO/gpurun_out/rocprof_trace.log 2>&1
(timeout 120 rocprofv3 -L) > $REPO/gpurun_out/counters_list.txt 2>&1
PM="python $REPO/bench.py --steps 1 --warmup 0 --structs 200 --no-cpu-baseline"
(timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU --output-format csv -d $REPO/gpurun_out/prof_$TAG -o pmc1 -- $PM) > $REPO/gpurun_out/rocprof_pmc1.log 2>&1
(timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d $REPO/gpurun_out/prof_$TAG -o pmc2 -- $PM) > $REPO/gpurun_out/rocprof_pmc2.log 2>&1
(timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $REPO/gpurun_out/prof_$TAG -o pmc3 -- $PM) > $REPO/gpurun_out/rocprof_pmc3.log 2>&1
(timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $REPO/gpurun_out/prof_$TAG -o pmc4 -- $PM) > $REPO/gpurun_out/rocprof_pmc4.log 2>&1
cd $REPO
find gpurun_out/prof_$TAG -name "*.csv" | head -50
tail -5 gpurun_out/pytest_gpu.log; tail -2 gpurun_out/bench.log | cut -c1-600
