cd $GRAFT_REPO_ROOT
L=freesasa_amd/lib
(timeout 900 bash tools/dev/sr_caps_ab.sh $L/libvar_base.so:16,32 $L/libfreesasa_amd.so:16,32 $L/libvar_base.so:16,32 $L/libfreesasa_amd.so:16,32 2>&1)
