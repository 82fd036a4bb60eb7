# Full GPU session for a round: everything profiles/ and DESIGN.md quote, in one gpurun call.
#   bash tools/gpu_round.sh r04        (outputs under gpurun_out/<tag>_*; copy the summaries into profiles/)
# Build the phase-ablation variants first if the per-phase instruction counts are wanted:
#   for k in 0 1 2 15 3 4 5; do bash tools/build_variant.sh stop$k -DLR2_STOP_AFTER=$k; done; bash tools/build_variant.sh pt -DSASA_PHASE_TIMING    (15: behind the contained caps, which follow P2)
#   for k in 0 1 12 2 3; do bash tools/build_variant.sh srstop$k -DSR_STOP_AFTER=$k; done
TAG=${1:-r06}
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
REPO=$(pwd)
O=$REPO/gpurun_out
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()") > $O/${TAG}_smoke.log 2>&1; echo "smoke rc=$?" >> $O/${TAG}_smoke.log
(timeout 900 python -m pytest tests -m gpu -q -s --durations=5) > $O/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/${TAG}_pytest_gpu.log
grep "adversarial" $O/${TAG}_pytest_gpu.log > $O/${TAG}_adversarial.txt
(timeout 900 python bench.py) > $O/${TAG}_bench.out 2> $O/${TAG}_bench.err; echo "bench rc=$?" >> $O/${TAG}_bench.err
grep '^{' $O/${TAG}_bench.out | tail -1 > $O/${TAG}_bench.json
export TMPDIR=/tmp
cd /tmp
BENCH="python $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-end-to-end --no-secondary --no-neighbors --no-live-counters --no-drivers --sustain-seconds 0"
(timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$TAG -o trace -- $BENCH) > $O/rocprof_trace.log 2>&1
cp $O/prof_$TAG/trace_kernel_stats.csv $O/${TAG}_kernel_stats.csv
PM="python $REPO/bench.py --steps 1 --warmup 1 --sync-entry --no-cpu-baseline --no-end-to-end --no-secondary --no-neighbors --no-live-counters --no-drivers --sustain-seconds 0"
(timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU --output-format csv -d $O/prof_$TAG -o pmc1 -- $PM) > $O/rocprof_pmc1.log 2>&1
(timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VALU --output-format csv -d $O/prof_$TAG -o pmc2 -- $PM) > $O/rocprof_pmc2.log 2>&1
(timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/prof_$TAG -o pmc3 -- $PM) > $O/rocprof_pmc3.log 2>&1
(timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/prof_$TAG -o pmc4 -- $PM) > $O/rocprof_pmc4.log 2>&1
# FETCH_SIZE / WRITE_SIZE calibration: copy kernels of known size, same session
for c in FETCH_SIZE WRITE_SIZE; do
  (timeout 120 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/calib_$c -o c -- $REPO/tools/dev/fetch_calib) > /dev/null 2>&1
  python - $O/calib_$c/c_counter_collection.csv $c <<'PY'
import csv, sys, collections
t = collections.defaultdict(float)
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] == sys.argv[2]: t[r["Kernel_Name"].split("(")[0]] += float(r["Counter_Value"])
for k, v in t.items():
    if "copy" in k or "gather" in k: print("%s %-14s %9.0f KB reported = %.3f x the 262144 KB the kernel moves" % (sys.argv[2], k, v, v / 262144))
PY
done > $O/${TAG}_fetch_calibration.txt
cd $REPO
python tools/pmc_summary.py $O/prof_$TAG all > $O/${TAG}_pmc_summary.txt
python tools/hbm_counters.py $O/prof_$TAG > $O/${TAG}_hbm_counters.json
# derived counters of the tile kernel (coils, then globules)
(bash tools/gpu_derived.sh "0,0,-1,0"; echo "--- globules"; STRUCTS=g100 bash tools/gpu_derived.sh "0,0,-1,0"; echo "--- the reference PDB entries x 84"; STRUCTS=p84 bash tools/gpu_derived.sh "0,0,-1,0") > $O/${TAG}_derived_counters.txt 2>&1
# VALU instructions by phase (cumulative builds), if the variants are there
if [ -f freesasa_amd/lib/libvar_stop0.so ]; then
  (echo "coils (300 x 10 000 atoms, 5 launches): cumulative after P0, P1, P2, the contained caps (stop 15), P3 .. P5, then the whole kernel"
   bash tools/gpu_ablate.sh "0,0,-1,0" freesasa_amd/lib/libvar_stop0.so freesasa_amd/lib/libvar_stop1.so freesasa_amd/lib/libvar_stop2.so freesasa_amd/lib/libvar_stop15.so freesasa_amd/lib/libvar_stop3.so freesasa_amd/lib/libvar_stop4.so freesasa_amd/lib/libvar_stop5.so freesasa_amd/lib/libfreesasa_amd.so
   echo "the reference's PDB entries x 84 (1.0e6 atoms, 5 launches)"
   STRUCTS=p84 bash tools/gpu_ablate.sh "0,0,-1,0" freesasa_amd/lib/libvar_stop0.so freesasa_amd/lib/libvar_stop1.so freesasa_amd/lib/libvar_stop2.so freesasa_amd/lib/libvar_stop15.so freesasa_amd/lib/libvar_stop3.so freesasa_amd/lib/libvar_stop4.so freesasa_amd/lib/libvar_stop5.so freesasa_amd/lib/libfreesasa_amd.so
   echo "globules (100 x 10 000 atoms, 5 launches)"
   STRUCTS=g100 bash tools/gpu_ablate.sh "0,0,-1,0" freesasa_amd/lib/libvar_stop0.so freesasa_amd/lib/libvar_stop1.so freesasa_amd/lib/libvar_stop2.so freesasa_amd/lib/libvar_stop15.so freesasa_amd/lib/libvar_stop3.so freesasa_amd/lib/libvar_stop4.so freesasa_amd/lib/libvar_stop5.so freesasa_amd/lib/libfreesasa_amd.so) 2>&1 | grep "==\|coils\|globules\|PDB entries\|lr2_tile<4" | sed "s/vgpr[^ ]* //" > $O/${TAG}_phase_valu.txt
fi
# wall clock of a wave by phase (variant built with -DSASA_PHASE_TIMING)
if [ -f freesasa_amd/lib/libvar_pt.so ]; then
  (echo "coils (300 x 10 000 atoms)"; FREESASA_AMD_LIB=$REPO/freesasa_amd/lib/libvar_pt.so python tools/gpu_shapes.py 300 "0,0,-1,0" 2>&1 | grep "phase clocks" | tail -1
   echo "globules (100 x 10 000 atoms)"; FREESASA_AMD_LIB=$REPO/freesasa_amd/lib/libvar_pt.so python tools/gpu_shapes.py g100 "0,0,-1,0" 2>&1 | grep "phase clocks" | tail -1) > $O/${TAG}_phase_clock.txt 2>&1
fi
# configs[2] as written (L&R 100 slices, TA 3): VALU instructions by phase
if [ -f freesasa_amd/lib/libvar_stop0.so ]; then
  (echo "coils, Lee-Richards 100 slices (100 x 10 000 atoms, 5 launches): cumulative after P0, P1, P2, the contained caps (stop 15), P3 .. P5, then the whole kernel"
   SLICES=100 STRUCTS=100 bash tools/gpu_ablate.sh "0,0,-1,0" freesasa_amd/lib/libvar_stop0.so freesasa_amd/lib/libvar_stop1.so freesasa_amd/lib/libvar_stop2.so freesasa_amd/lib/libvar_stop15.so freesasa_amd/lib/libvar_stop3.so freesasa_amd/lib/libvar_stop4.so freesasa_amd/lib/libvar_stop5.so freesasa_amd/lib/libfreesasa_amd.so) 2>&1 | grep "==\|coils\|lr2_tile<" | sed "s/vgpr[^ ]* //" > $O/${TAG}_lr100_phase_valu.txt
fi
(timeout 600 python tools/deep_parity.py 120 24 2>/dev/null | tail -1) > $O/${TAG}_deep_parity.json
(timeout 900 python tools/deep_parity.py 500 100 2>/dev/null | tail -1) > $O/${TAG}_deep_parity_large.json
# Shrake-Rupley: kernel trace + counters on its three workloads, and its phases (variants built with -DSR_STOP_AFTER=k)
(bash tools/gpu_sr.sh $TAG) > $O/${TAG}_sr_session.txt 2>&1
if [ -f freesasa_amd/lib/libvar_srstop0.so ]; then (bash tools/gpu_sr_ablate.sh) > $O/${TAG}_sr_phase_ablation.txt 2>&1; fi
(timeout 300 python tools/dev/driver_tuning.py) > $O/${TAG}_driver_tuning.txt 2>&1
# round 6: the device-side parser's kernels (kernel trace of the bench's sweep_files key), the sweep's workers / batch sizes,
# the 200 000-atom Shrake-Rupley step's timeline, the trajectory driver's kernel + copy trace and per-lane host time
mkdir -p /tmp/fsbench
(cd /tmp && TMPDIR=/tmp timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_sweep -o trace -- python $REPO/tools/dev/sweep_bench.py) > $O/${TAG}_sweep_files_key.txt 2>&1
cp $O/prof_${TAG}_sweep/trace_kernel_stats.csv $O/${TAG}_sweep_kernel_stats.csv 2>/dev/null
(timeout 300 python tools/dev/sweep_devparse.py 2>&1 | grep -v amdgpu) > $O/${TAG}_sweep_workers.txt
(timeout 200 bash tools/dev/sr200k_timeline.sh 2>&1 | grep -v amdgpu) > $O/${TAG}_sr200k_timeline.txt
(timeout 200 bash tools/dev/traj_trace.sh 2>&1 | grep -v amdgpu) > $O/${TAG}_trajectory_trace.txt
(timeout 200 python tools/dev/traj_profile.py 2>&1 | grep -v amdgpu) > $O/${TAG}_trajectory_lanes.txt
(timeout 120 tools/dev/ubench) > $O/${TAG}_ubench.txt 2>&1
# round 6: the contained caps off / on / off / on, kernel ms on the L&R workloads; the same on random inputs, bit for bit
(WL="c20 c50 c100 pdb pdb100 glob" timeout 900 bash tools/dev/prune_ab.sh "0 4 0 4" 2>&1 | grep "==\|slices\|PDB\|globules\|kernel_ms" | cut -c1-130) > $O/${TAG}_prune_ab.txt
(timeout 900 python tools/dev/prune_fuzz.py 3000 777 2>&1 | grep -v amdgpu | tail -3) > $O/${TAG}_prune_fuzz.txt
tail -2 $O/${TAG}_smoke.log; tail -4 $O/${TAG}_pytest_gpu.log; cut -c1-400 $O/${TAG}_bench.json; cat $O/${TAG}_kernel_stats.csv | cut -d, -f1-4 | head -12; cat $O/${TAG}_fetch_calibration.txt; cat $O/${TAG}_deep_parity.json
