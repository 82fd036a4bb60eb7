# A/B of library variants through bench.py itself (whole steps): bash tools/dev/bench_ab.sh name1 name2 ...  (freesasa_amd/lib/libvar_<name>.so)
for rep in 1 2 3; do for lib in "$@"; do echo -n "$lib: "; FREESASA_AMD_LIB=$PWD/freesasa_amd/lib/libvar_$lib.so python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-end-to-end --no-secondary --no-neighbors 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value %.4g ms/step %.3f kernel %.3f prep %.3f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['prep_ms']))"; done; done
