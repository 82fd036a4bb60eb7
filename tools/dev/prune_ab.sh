# DEV: contained caps (lr2_prune_contained) on / off, the number of caps wanted and the list's capacity: kernel ms on the L&R workloads
#   gpurun -- 'WL="c100" bash tools/dev/prune_ab.sh "0 4,4 8,8 8,12"'        WL: c20 c100 pdb glob (default: all)
export PYTHONUNBUFFERED=1
WL=${WL:-c20 c100 pdb glob}
for p in ${1:-0 8}; do
  echo "== FREESASA_AMD_PRUNE=$p"
  for w in $WL; do
    case $w in
      c20)  echo "coils 300 x 10000, 20 slices";  FREESASA_AMD_PRUNE=$p python tools/gpu_shapes.py 300 "0,0,-1,0" 2>&1 | grep kernel_ms;;
      c100) echo "coils 100 x 10000, 100 slices"; FREESASA_AMD_PRUNE=$p SLICES=100 python tools/gpu_shapes.py 100 "0,0,-1,0" 2>&1 | grep kernel_ms;;
      c50)  echo "coils 100 x 10000, 50 slices";  FREESASA_AMD_PRUNE=$p SLICES=50 python tools/gpu_shapes.py 100 "0,0,-1,0" 2>&1 | grep kernel_ms;;
      pdb)  echo "PDB entries x 84";              FREESASA_AMD_PRUNE=$p python tools/gpu_shapes.py p84 "0,0,-1,0" "3,0,-1,0" 2>&1 | grep kernel_ms;;
      glob) echo "globules 100 x 10000";          FREESASA_AMD_PRUNE=$p python tools/gpu_shapes.py g100 "0,0,-1,0" 2>&1 | grep kernel_ms;;
      pdb100) echo "PDB entries x 84, 100 slices"; FREESASA_AMD_PRUNE=$p SLICES=100 python tools/gpu_shapes.py p84 "0,0,-1,0" 2>&1 | grep kernel_ms;;
    esac
  done
done
