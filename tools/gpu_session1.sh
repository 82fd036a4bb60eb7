# DEV: scratch GPU session (edited per call)
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
L=freesasa_amd/lib
(bash tools/gpu_ablate.sh "0,0,-1,0" $L/libvar_base_stop2.so $L/libvar_s32_stop2.so $L/libvar_base_stop3.so $L/libvar_s32_stop3.so $L/libvar_base_stop4.so $L/libvar_s32_stop4.so) 2>&1 | grep "==\|lr2_tile<4" | sed "s/vgpr[^)]*) //" > gpurun_out/s3_pmc.log 2>&1
cat gpurun_out/s3_pmc.log
