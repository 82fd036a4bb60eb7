import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, freesasa_amd as fa, tools, bench
from freesasa_amd import ingest
out = bench.driver_workloads(fa, tools, 0, "/tmp/fsbench")
import json
print(json.dumps({k: v for k, v in out["sweep_files"].items() if k != "workload"}, indent=1))
