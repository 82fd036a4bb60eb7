cd $GRAFT_REPO_ROOT
L=freesasa_amd/lib
(timeout 600 bash tools/dev/sr_caps_ab.sh $L/libfreesasa_amd.so:16,32 $L/libvar_cp1.so:16,32 $L/libvar_cp2.so:16,32 2>&1) > gpurun_out/caps_ab11.txt
cat gpurun_out/caps_ab11.txt
