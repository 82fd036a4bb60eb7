# rocprofv3 kernel stats of the secondary workloads (S&R 200k-atom globule, L&R-100 coils) -> gpurun_out/
TAG=${1:-r01}
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONUNBUFFERED=1
REPO=$(pwd)
cd /tmp
(timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_${TAG}_sr -o trace -- python $REPO/bench.py --workload globule_sr --steps 20 --warmup 3) > $REPO/gpurun_out/rocprof_sr.log 2>&1
(timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_${TAG}_lr100 -o trace -- python $REPO/bench.py --slices 100 --structs 200 --steps 3 --warmup 1 --no-cpu-baseline) > $REPO/gpurun_out/rocprof_lr100.log 2>&1
cd $REPO
head -6 gpurun_out/prof_${TAG}_sr/trace_kernel_stats.csv; head -5 gpurun_out/prof_${TAG}_lr100/trace_kernel_stats.csv
