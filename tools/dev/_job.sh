cd $GRAFT_REPO_ROOT
(timeout 600 python -m pytest tests/test_sr_caps.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -4)
L=freesasa_amd/lib
(timeout 600 bash tools/dev/sr_caps_ab.sh $L/libfreesasa_amd.so:16,32 2>&1)
