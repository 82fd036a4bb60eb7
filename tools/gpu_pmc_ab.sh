# VALU / SALU / LDS instruction counts and wave cycles of the L&R kernel per environment variant:
#   bash tools/gpu_pmc_ab.sh <structs-spec e.g. g100 or 300> "VAR=val" "VAR=val2" ...
export PYTHONUNBUFFERED=1
REPO=$(pwd)
GEOM=$1; shift
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
i=0
for envs in "$@"; do
  i=$((i+1))
  rm -rf $REPO/gpurun_out/pmcab_$i
  env $envs timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $REPO/gpurun_out/pmcab_$i -o pmc1 -- python $REPO/tools/gpu_shapes.py $GEOM "0,0,-1,0" > $REPO/gpurun_out/pmcab_$i.log 2>&1
  echo "== $envs: $(grep kernel_ms $REPO/gpurun_out/pmcab_$i.log)"
  python $REPO/tools/pmc_summary.py $REPO/gpurun_out/pmcab_$i | grep "lr2_tile<[234]"
done
