# kernel-level timing of the launches of one headline-shaped step (300 structures)
export TMPDIR=/tmp; R=$(pwd); cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/t1 -o t -- python $R/bench.py --steps 4 --warmup 1 --structs 300 --no-cpu-baseline > $R/gpurun_out/t1.log 2>&1
grep -E "k_lr_tile|k_scatter|k_count" $R/gpurun_out/t1/t_kernel_stats.csv | cut -d, -f1-4 | cut -c1-120
grep '^{' $R/gpurun_out/t1.log | cut -c1-120
