cd $GRAFT_REPO_ROOT
L=freesasa_amd/lib
(timeout 900 bash tools/dev/sr_caps_ab.sh $L/libfreesasa_amd.so:16,32 $L/libvar_u2w5.so:16,32 $L/libvar_u2w6.so:16,32 $L/libvar_u4w5.so:16,32 $L/libvar_u1w6.so:16,32 2>&1) > gpurun_out/caps_ab16.txt
cat gpurun_out/caps_ab16.txt
