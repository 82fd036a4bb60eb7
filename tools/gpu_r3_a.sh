# round 3, call A: instruction issue costs + LDS trade-offs of the current kernel
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out
./tools/dev/ubench > gpurun_out/r3_ubench.txt 2>&1; cat gpurun_out/r3_ubench.txt
python tools/gpu_r2_sweep.py 300 "0,0,-1,0" "6,0,1,32" "6,0,0,32" "6,184,2,32" "6,184,1,32" "6,200,2,32" "6,0,2,16" "6,0,2,8" 2>&1 | tee gpurun_out/r3_sweep_a.txt
python tools/gpu_r2_sweep.py g100 "0,0,-1,0" "3,0,1,32" "4,0,-1,32" 2>&1 | tee -a gpurun_out/r3_sweep_a.txt
