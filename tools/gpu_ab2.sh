# A/B with environment variants: bash tools/gpu_ab2.sh "ENV=.. lib" ...  (each spec: optional VAR=val words, then the library)
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out
for rep in 1 2; do
for spec in "$@"; do
  echo "== $spec (rep $rep)"
  lib=${spec##* }; envs=${spec% *}; [ "$envs" = "$spec" ] && envs=""
  env $envs FREESASA_AMD_LIB=$PWD/$lib python tools/gpu_r2_sweep.py ${STRUCTS:-300} "0,0,-1,0" 2>&1 | grep kernel_ms
  env $envs FREESASA_AMD_LIB=$PWD/$lib python tools/gpu_r2_sweep.py g100 "0,0,-1,0" ${GSPEC} 2>&1 | grep kernel_ms
done
done 2>&1 | tee gpurun_out/ab2_$(date +%H%M%S).txt
