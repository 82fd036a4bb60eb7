#!/usr/bin/env python3
"""Static instruction mix of the loops of one kernel in hipcc's -S output (dev aid).
usage: isa_loops.py file.s kernel_symbol_prefix"""
import re, sys
lines = open(sys.argv[1]).read().split("\n")
pref = sys.argv[2]
start = next(i for i, l in enumerate(lines) if l.startswith(pref) and l.rstrip().split(";")[0].strip().endswith(":"))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = lines[start:end + 1]
label = {}
for i, l in enumerate(body):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m: label[m.group(1)] = i
loops = []
for i, l in enumerate(body):
    m = re.match(r"^\s*s_c?branch\S*\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in label and label[m.group(1)] <= i:
        loops.append((label[m.group(1)], i))
def mix(a, b):
    c = dict(valu=0, lane=0, salu=0, ds=0, vmem=0, wait=0, trans=0)
    for l in body[a:b + 1]:
        t = l.strip().split(" ")[0]
        if t.startswith("v_readlane") or t.startswith("v_writelane") or t.startswith("v_readfirstlane"): c["lane"] += 1
        elif t.startswith("v_"):
            c["valu"] += 1
            if re.match(r"v_(rcp|rsq|sqrt|div_scale|div_fmas|div_fixup)", t): c["trans"] += 1
        elif t.startswith("s_waitcnt"): c["wait"] += 1
        elif t.startswith("s_"): c["salu"] += 1
        elif t.startswith("ds_"): c["ds"] += 1
        elif t.startswith("global_") or t.startswith("buffer_") or t.startswith("flat_") or t.startswith("scratch_"): c["vmem"] += 1
    return c
print("whole", mix(0, len(body) - 1))
for a, b in sorted(loops):
    inner = not any((a2 >= a and b2 <= b and (a2, b2) != (a, b)) for a2, b2 in loops)
    print(f"{'inner' if inner else 'outer'} lines {a}-{b} ({b - a + 1})", mix(a, b))
