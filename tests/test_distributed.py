"""World-size-2 CPU (gloo) test of the only multi-process logic the path has: independent
structure shards per rank (disjoint seeds, no data-path collective) and the max-over-ranks
timing reduction that bench.py performs."""
import os
import socket
import subprocess
import sys
import textwrap

from conftest import ROOT

WORKER = textwrap.dedent("""
    import os, sys, json
    import numpy as np
    import torch, torch.distributed as dist
    sys.path.insert(0, sys.argv[1])
    import tools, oracle
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    structs, atoms = 6, 300
    xyz, r, offs = tools.coil_batch(structs, atoms, seed0=1000 + rank * structs)   # bench.py's sharding rule
    o = oracle.Oracle()
    tot = np.array([o.total(o.lee_richards(xyz[offs[k]:offs[k+1]], r[offs[k]:offs[k+1]])) for k in range(structs)])
    # every rank's shard is different work ...
    gathered = [torch.zeros(structs, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(gathered, torch.from_numpy(tot))
    # ... and the reported time is the slowest rank's
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"t": t.item(), "distinct": bool(not torch.equal(gathered[0], gathered[1])),
                          "n": sum(len(g) for g in gathered)}))
    dist.barrier()
    dist.destroy_process_group()
""")


def test_two_rank_sharding_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script), ROOT],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res == {"t": 2.0, "distinct": True, "n": 12}
