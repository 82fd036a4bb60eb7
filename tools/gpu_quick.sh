# quick GPU iteration: tests (optional), bench, two PMC passes on a 200-structure batch
TAG=${1:-q}
TESTS=${2:-1}
set -x
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
REPO=$(pwd)
if [ "$TESTS" = "1" ]; then
(timeout 900 python -m pytest tests -m gpu -q -x) > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
fi
(timeout 400 python bench.py --steps 5 --warmup 1 --no-cpu-baseline) > gpurun_out/bench_$TAG.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench_$TAG.log
export TMPDIR=/tmp
cd /tmp
PM="python $REPO/bench.py --steps 1 --warmup 0 --structs 200 --no-cpu-baseline"
(timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU --output-format csv -d $REPO/gpurun_out/prof_$TAG -o pmc1 -- $PM) > $REPO/gpurun_out/rocprof_pmc1.log 2>&1
(timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VALU --output-format csv -d $REPO/gpurun_out/prof_$TAG -o pmc2 -- $PM) > $REPO/gpurun_out/rocprof_pmc2.log 2>&1
cd $REPO
python tools/pmc_summary.py gpurun_out/prof_$TAG | tee gpurun_out/pmc_$TAG.txt
tail -4 gpurun_out/pytest_gpu.log; tail -2 gpurun_out/bench_$TAG.log | cut -c1-2500
