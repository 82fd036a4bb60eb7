# DEV: forced tile shapes (FREESASA_AMD_LR2="TA,pool,ds,refill") on the dense L&R workloads at the bench's sizes: kernel ms
export PYTHONUNBUFFERED=1
for geom in ${GEOMS:-p251 g300}; do
  echo "== $geom"
  python tools/gpu_shapes.py $geom "0,0,-1,0" "3,0,-1,0" "4,0,-1,0" "3,0,-1,0" "4,0,-1,0" 2>/dev/null | cut -c1-150
done
