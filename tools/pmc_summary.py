"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel (sums over dispatches)."""
import collections
import csv
import glob
import sys

d = sys.argv[1]
ALL = len(sys.argv) > 2
for f in sorted(glob.glob(d + "/*_counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    meta = {}
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:48]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        meta[k] = (r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r["Workgroup_Size"])
    for k, v in agg.items():
        if "tile" in k or (ALL and ("FETCH_SIZE" in v or "WRITE_SIZE" in v)):
            w = v.get("SQ_WAVES", 0)
            line = f"{f.split('/')[-1][:4]} {k} vgpr/agpr/sgpr/lds/wg={meta[k]} "
            line += " ".join(f"{c}={x:.4g}" + (f"({x / w:.0f}/wave)" if w else "") for c, x in sorted(v.items()))
            print(line)
