# round 2, session A: parity suite on the GPU, then the launch-shape sweep of the new L&R kernel
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/a_pytest.txt
cat gpurun_out/a_pytest.txt
timeout 600 python tools/gpu_r2_sweep.py 300 2>&1 | tee gpurun_out/a_sweep.txt
