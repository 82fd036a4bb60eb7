"""FETCH_SIZE / WRITE_SIZE per kernel launch (KB) from the two single-counter rocprofv3 --pmc
passes of tools/gpu_round.sh -> profiles/<tag>_hbm_counters.json (what bench.py reports as
roofline.traffic).  Counter values of one dispatch are summed over the XCD instances.

    python tools/hbm_counters.py gpurun_out/prof_r01 > profiles/r01_hbm_counters.json
"""
import collections
import csv
import json
import sys

d = sys.argv[1]
out = collections.OrderedDict()
for f, counter in ((d + "/pmc3_counter_collection.csv", "FETCH_SIZE"), (d + "/pmc4_counter_collection.csv", "WRITE_SIZE")):
    total = collections.defaultdict(float)
    launches = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        total[r["Kernel_Name"]] += float(r["Counter_Value"])
        launches[r["Kernel_Name"]].add(r["Dispatch_Id"])
    for k, v in total.items():
        out.setdefault(k, {})[counter] = {"per_launch_KB": v / len(launches[k]), "launches": len(launches[k])}
# wave-level VALU instructions per launch from the first SQ pass (bench.py turns it into the issue-port
# utilisation, the roofline that actually binds this kernel)
try:
    total = collections.defaultdict(float)
    launches = collections.defaultdict(set)
    for r in csv.DictReader(open(d + "/pmc1_counter_collection.csv")):
        if r["Counter_Name"] == "SQ_INSTS_VALU":
            total[r["Kernel_Name"]] += float(r["Counter_Value"])
            launches[r["Kernel_Name"]].add(r["Dispatch_Id"])
    for k, v in total.items():
        out.setdefault(k, {})["SQ_INSTS_VALU"] = {"per_launch": v / len(launches[k]), "launches": len(launches[k])}
except FileNotFoundError:
    pass
json.dump(out, sys.stdout, indent=1)
