# Round-end GPU session: everything profiles/ and DESIGN.md quote, in one gpurun call.
#   bash tools/gpu_final.sh r01
TAG=${1:-r01}
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out
bash tools/gpu_round.sh $TAG > gpurun_out/round.log 2>&1
for w in "--workload globule_sr --steps 20 --warmup 3" "--slices 100 --structs 200 --steps 3 --warmup 1 --no-cpu-baseline" \
         "--workload sweep_lr --steps 5 --warmup 1" "--workload traj_lr"; do
    timeout 300 python bench.py $w 2>/dev/null | grep '^{' | tail -1
done > gpurun_out/secondary_$TAG.jsonl
timeout 200 bash tools/gpu_mini.sh 2>&1 | grep -v amdgpu.ids > gpurun_out/mini_$TAG.log
bash tools/gpu_secondary_profiles.sh $TAG > gpurun_out/secondary_profiles.log 2>&1
tail -3 gpurun_out/round.log | cut -c1-600
python - <<'PY'
import json
for l in open('gpurun_out/secondary_r01.jsonl'):
    d = json.loads(l); print('%-50s %.4g %s  ms/step %.3f' % (d['metric'], d['value'], d['unit'], d.get('ms_per_step', 0)))
PY
cat gpurun_out/mini_$TAG.log
