export PYTHONUNBUFFERED=1
echo "--- lr100"; SLICES=100 STRUCTS=100 bash tools/gpu_ablate.sh "0,0,-1,0" "$@" 2>&1 | grep "==\|lr2_tile<2" | cut -c1-250
echo "--- coils"; STRUCTS=300 bash tools/gpu_ablate.sh "0,0,-1,0" "$@" 2>&1 | grep "==\|lr2_tile<4" | cut -c1-250
