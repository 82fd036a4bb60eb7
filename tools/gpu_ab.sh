# A/B of library variants on the headline geometry (300 coils) and on globules (100): bash tools/gpu_ab.sh lib1 lib2 ...
# REPS rounds (default 3), variants interleaved; the last lines give every variant's best kernel time
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out
OUT=gpurun_out/ab_$(date +%H%M%S).txt
for rep in $(seq 1 ${REPS:-3}); do
for lib in "$@"; do
  echo "== $lib (rep $rep)"
  FREESASA_AMD_LIB=$PWD/$lib python tools/gpu_shapes.py ${STRUCTS:-300} "0,0,-1,0" 2>&1 | grep kernel_ms | sed "s/^/coil /"
  FREESASA_AMD_LIB=$PWD/$lib python tools/gpu_shapes.py g100 "0,0,-1,0" 2>&1 | grep kernel_ms | sed "s/^/glob /"
done
done 2>&1 | tee $OUT | grep -v "^coil\|^glob" > /dev/null
python - $OUT <<'PY'
import sys, re, collections
best = collections.defaultdict(lambda: [1e9, 1e9]); lib = None
for l in open(sys.argv[1]):
    if l.startswith("=="): lib = l.split()[1]
    m = re.match(r"(coil|glob) .*kernel_ms\s+([0-9.]+)", l)
    if m and lib: best[lib][0 if m.group(1) == "coil" else 1] = min(best[lib][0 if m.group(1) == "coil" else 1], float(m.group(2)))
for k, v in best.items(): print("%-44s coil %.3f ms  globule %.3f ms" % (k, v[0], v[1]))
PY
