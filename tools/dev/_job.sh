cd $GRAFT_REPO_ROOT
(timeout 600 python -m pytest tests/test_sr_caps.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3)
L=freesasa_amd/lib
for nr in 1 ""; do
  if [ -n "$nr" ]; then export FREESASA_AMD_NO_ROWS=1; else unset FREESASA_AMD_NO_ROWS; fi
  echo "== NO_ROWS=$nr"
  (timeout 600 bash tools/dev/sr_caps_ab.sh $L/libfreesasa_amd.so:16,32 $L/libfreesasa_amd.so:0 2>&1)
done
