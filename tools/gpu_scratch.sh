# DEV: scratch GPU session (edited per call; not part of the product)
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
O=gpurun_out; TAG=r04
rm -rf gpurun_out/abl_*
(echo "coils (300 x 10 000 atoms, 5 launches): cumulative after P0 .. P5, then the whole kernel"
 bash tools/gpu_ablate.sh "0,0,-1,0" freesasa_amd/lib/libvar_stop0.so freesasa_amd/lib/libvar_stop1.so freesasa_amd/lib/libvar_stop2.so freesasa_amd/lib/libvar_stop3.so freesasa_amd/lib/libvar_stop4.so freesasa_amd/lib/libvar_stop5.so freesasa_amd/lib/libfreesasa_amd.so
 echo "globules (100 x 10 000 atoms, 5 launches)"
 STRUCTS=g100 bash tools/gpu_ablate.sh "0,0,-1,0" freesasa_amd/lib/libvar_stop0.so freesasa_amd/lib/libvar_stop1.so freesasa_amd/lib/libvar_stop2.so freesasa_amd/lib/libvar_stop3.so freesasa_amd/lib/libvar_stop4.so freesasa_amd/lib/libvar_stop5.so freesasa_amd/lib/libfreesasa_amd.so) 2>&1 | grep "==\|coils\|globules\|lr2_tile<4" | sed "s/vgpr[^ ]* //" > $O/${TAG}_phase_valu.txt
rm -rf gpurun_out/abl_*
(echo "coils, Lee-Richards 100 slices (100 x 10 000 atoms, 5 launches): cumulative after P0 .. P5, then the whole kernel"
 SLICES=100 STRUCTS=100 bash tools/gpu_ablate.sh "0,0,-1,0" freesasa_amd/lib/libvar_stop0.so freesasa_amd/lib/libvar_stop1.so freesasa_amd/lib/libvar_stop2.so freesasa_amd/lib/libvar_stop3.so freesasa_amd/lib/libvar_stop4.so freesasa_amd/lib/libvar_stop5.so freesasa_amd/lib/libfreesasa_amd.so) 2>&1 | grep "==\|coils\|lr2_tile<[234]" | sed "s/vgpr[^ ]* //" > $O/${TAG}_lr100_phase_valu.txt
python - <<'PY'
import re
for f in ("gpurun_out/r04_phase_valu.txt","gpurun_out/r04_lr100_phase_valu.txt"):
    for l in open(f):
        m=re.search(r"SQ_INSTS_VALU=\S+\((\d+)/wave\).*SQ_WAVE_CYCLES=\S+\((\d+)/wave\)",l)
        if l.startswith("=="): name=l.split(":")[0]
        if m: print(name, "VALU/wave", m.group(1), "wave quad-cycles", m.group(2))
        if "coils" in l or "globules" in l: print(l.strip())
PY
