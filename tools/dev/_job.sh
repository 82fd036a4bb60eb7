cd $GRAFT_REPO_ROOT
L=freesasa_amd/lib
(timeout 600 python -m pytest tests -m gpu -x -q -k "sr or shrake or deep or points or parity" 2>&1 | tail -3)
(FREESASA_AMD_SHOW_SHAPE=1 timeout 600 bash tools/dev/sr_caps_ab.sh $L/libfreesasa_amd.so:16,32 2>&1) > gpurun_out/caps_ab14.txt
cat gpurun_out/caps_ab14.txt
for wl in coil_sr pdb_sr; do FREESASA_AMD_SHOW_SHAPE=1 python bench.py --workload $wl --steps 4 --warmup 2 --no-cpu-baseline --no-end-to-end --no-secondary --no-neighbors --no-live-counters --no-drivers 2>&1 | grep "tile shape" | sort | uniq -c; done
