# DEV: wave-level VALU instructions of the L&R kernel, cumulative by phase (variant builds -DLR2_STOP_AFTER=k)
# bash tools/gpu_r2_ablate.sh "<spec>" lib1 lib2 ...
export PYTHONUNBUFFERED=1
REPO=$(pwd)
SPEC=$1; shift
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
for lib in "$@"; do
  tag=$(basename $lib .so)
  FREESASA_AMD_LIB=$REPO/$lib timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $REPO/gpurun_out/abl_$tag -o pmc1 -- python $REPO/tools/gpu_shapes.py ${STRUCTS:-300} "$SPEC" > $REPO/gpurun_out/abl_$tag.log 2>&1
  echo "== $tag: $(grep kernel_ms $REPO/gpurun_out/abl_$tag.log)"
  python $REPO/tools/pmc_summary.py $REPO/gpurun_out/abl_$tag | grep lr2
done
