"""DEV: bench.py's driver workloads alone (for a kernel trace of the sweep_files key)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, freesasa_amd as fa, tools, bench
from freesasa_amd import ingest
os.makedirs("/tmp/fsbench", exist_ok=True)
out = bench.driver_workloads(fa, tools, 0, "/tmp/fsbench")
import json
print(json.dumps({k: v for k, v in out["sweep_files"].items() if k != "workload"}, indent=1))
