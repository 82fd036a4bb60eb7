# DEV: scratch GPU session (edited per call)
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
L=freesasa_amd/lib
REPS=5 bash tools/gpu_ab.sh $L/libvar_pre4.so $L/libvar_pre6.so $L/libvar_pre6sel.so $L/libvar_pre6selh.so $L/libvar_pre6h.so 2>&1 | tee gpurun_out/s14_ab.log
