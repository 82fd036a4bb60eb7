set -x
export PYTHONUNBUFFERED=1
python bench.py --workload globule_sr --steps 20 --warmup 3 2>&1 | tail -1 | cut -c1-1500
python bench.py --slices 100 --structs 200 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-900
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --output-format csv -d /root/repo/gpurun_out/prof_sr -o pmc1 -- python /root/repo/bench.py --workload globule_sr --steps 2 --warmup 1 > /dev/null 2>&1
cd /root/repo; python tools/pmc_summary.py gpurun_out/prof_sr
