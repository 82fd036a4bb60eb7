cd $GRAFT_REPO_ROOT
L=freesasa_amd/lib
(timeout 300 python -m pytest tests -m gpu -x -q -k "sr or shrake or deep or points" 2>&1 | tail -3)
(timeout 600 bash tools/dev/sr_caps_ab.sh $L/libfreesasa_amd.so:16,32 $L/libfreesasa_amd.so:0 2>&1) > gpurun_out/caps_ab12.txt
cat gpurun_out/caps_ab12.txt
