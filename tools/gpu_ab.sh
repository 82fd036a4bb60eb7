# A/B of library variants on the headline geometry (300 coils) and on globules (100): bash tools/gpu_ab.sh lib1 lib2 ...
# each variant twice, interleaved, so that clock drift between runs shows
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out
for rep in 1 2; do
for lib in "$@"; do
  echo "== $lib (rep $rep)"
  FREESASA_AMD_LIB=$PWD/$lib python tools/gpu_r2_sweep.py ${STRUCTS:-300} "0,0,-1,0" 2>&1 | grep kernel_ms
  FREESASA_AMD_LIB=$PWD/$lib python tools/gpu_r2_sweep.py g100 "0,0,-1,0" 2>&1 | grep kernel_ms
done
done 2>&1 | tee gpurun_out/ab_$(date +%H%M%S).txt
