# First GPU session of a round: does everything still work on the device?  smoke, the gpu-marked tests, one bench line.
#   bash tools/gpu_check.sh r05a
TAG=${1:-check}
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
O=$PWD/gpurun_out
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()") > $O/${TAG}_smoke.log 2>&1; echo "smoke rc=$?" >> $O/${TAG}_smoke.log
(timeout 1500 python -m pytest tests -m gpu -q -x -s --durations=12) > $O/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/${TAG}_pytest_gpu.log
(timeout 900 python bench.py) > $O/${TAG}_bench.out 2> $O/${TAG}_bench.err; echo "bench rc=$?" >> $O/${TAG}_bench.err
grep '^{' $O/${TAG}_bench.out | tail -1 > $O/${TAG}_bench.json
tail -2 $O/${TAG}_smoke.log; tail -25 $O/${TAG}_pytest_gpu.log; tail -5 $O/${TAG}_bench.err; cut -c1-600 $O/${TAG}_bench.json
