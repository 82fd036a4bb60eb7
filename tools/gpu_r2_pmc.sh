# DEV: SQ counters of the L&R kernels on a 300-structure batch: bash tools/gpu_r2_pmc.sh "<spec> <spec> ..."
export PYTHONUNBUFFERED=1
REPO=$(pwd)
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
CMD="python $REPO/tools/gpu_r2_sweep.py ${STRUCTS:-300} $1"
(timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU --output-format csv -d $REPO/gpurun_out/prof_dev -o pmc1 -- $CMD) > $REPO/gpurun_out/rocprof_pmc1.log 2>&1
(timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VALU --output-format csv -d $REPO/gpurun_out/prof_dev -o pmc2 -- $CMD) > $REPO/gpurun_out/rocprof_pmc2.log 2>&1
cd $REPO
tail -3 gpurun_out/rocprof_pmc1.log
python tools/pmc_summary.py gpurun_out/prof_dev | tee gpurun_out/pmc_dev.txt
