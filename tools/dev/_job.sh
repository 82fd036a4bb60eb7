cd $GRAFT_REPO_ROOT
L=freesasa_amd/lib
(timeout 900 bash tools/dev/sr_caps_ab.sh $L/libfreesasa_amd.so:16,32 $L/libvar_r1w5.so:16,32 $L/libvar_r1w6.so:16,32 $L/libvar_r3w5.so:16,32 2>&1) > gpurun_out/caps_ab17.txt
cat gpurun_out/caps_ab17.txt
