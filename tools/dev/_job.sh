cd $GRAFT_REPO_ROOT
(timeout 1500 python tools/dev/sr_caps_fuzz.py 3000 31337 2>&1 | grep -v amdgpu | tail -3) > gpurun_out/r06_sr_caps_fuzz.txt
cat gpurun_out/r06_sr_caps_fuzz.txt
