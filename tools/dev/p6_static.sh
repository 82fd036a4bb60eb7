# DEV: static instruction mix of the arc pass loops of the main L&R kernel (no GPU needed)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off "$@" -S --cuda-device-only -o /tmp/ge.s freesasa_amd/csrc/gpu_kernels.hip 2>/dev/null
K=_Z10k_lr2_tileILi4ELi0ELi4ELb0ELb1ELi1EEvN4sasa7Lr2ArgsE
awk "/^$K:/{p=1} p{print} /^.Lfunc_end/{if(p){exit}}" /tmp/ge.s > /tmp/k4.s
python tools/dev/isa_loops.py /tmp/ge.s $K > /tmp/loops.txt
# the arc-step loop is the outer loop that contains v_rsq_f64 and a ds_write_b128 and is the smallest such
python - <<'PY'
import re
lines=open('/tmp/k4.s').read().split('\n')
best=None
for l in open('/tmp/loops.txt'):
    m=re.match(r"(inner|outer) lines (\d+)-(\d+) \((\d+)\) (.*)",l)
    if not m: continue
    a,b=int(m.group(2)),int(m.group(3))
    body='\n'.join(lines[a:b+1])
    if 'v_rsq_f64' in body and 'v_ffbl_b32' in body:
        if best is None or (b-a)<best[1]-best[0]: best=(a,b,m.group(5))
        last=(a,b,m.group(5))
print("arc step loop", best)
# P6 loop: smallest loop containing the arc step loop and ds_read_u16 (queue fetch)
cand=None
for l in open('/tmp/loops.txt'):
    m=re.match(r"(inner|outer) lines (\d+)-(\d+) \((\d+)\) (.*)",l)
    if not m: continue
    a,b=int(m.group(2)),int(m.group(3))
    if a<=best[0] and b>=best[1] and (a,b)!=(best[0],best[1]):
        body='\n'.join(lines[a:b+1])
        if 'ds_read_u16' in body and (cand is None or (b-a)<cand[1]-cand[0]): cand=(a,b,m.group(5))
print("arc pass loop", cand)
PY
grep -A8 "k_lr2_tileILi4ELi0ELi4ELb0ELb1ELi1E" freesasa_amd/lib/kernel_resources.txt | grep -i " vgprs:\|VGPRs Spill\|SGPRs Spill" | tr '\n' ' ' | sed "s/freesasa_amd\/csrc\/gpu_engine.hip:[0-9]*:1: remark://g;s/\[-Rpass-analysis=kernel-resource-usage\]//g"; echo
