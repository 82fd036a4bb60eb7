"""World-size-2 CPU (gloo) tests of the only multi-process logic the path has: independent
structure shards per rank (disjoint seeds or an LPT partition of one ragged list, no data-path
collective) and the max-over-ranks / sum-of-atoms reductions that bench.py performs."""
import os
import socket
import subprocess
import sys
import textwrap

import pytest

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


SWEEP_WORKER = textwrap.dedent("""
    import os, sys, json
    import numpy as np
    import torch, torch.distributed as dist
    sys.path.insert(0, sys.argv[1])
    import tools, oracle
    from freesasa_amd import shard
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # bench.py --workload sweep_lr: one global ragged list, LPT on atom count
    sizes = np.exp(np.random.default_rng(2024).uniform(np.log(20), np.log(400), 14)).astype(np.int64)
    mine = shard.lpt(sizes, world)[rank]
    o = oracle.Oracle()
    tot = torch.zeros(len(sizes), dtype=torch.float64)
    for k in mine:
        xyz, r = tools.coil(int(sizes[k]), 5000 + int(k))
        tot[k] = o.total(o.lee_richards(xyz, r))
    dist.all_reduce(tot)                                    # host-side gather of results only
    na = torch.tensor([float(sizes[mine].sum())], dtype=torch.float64)
    dist.all_reduce(na)
    if rank == 0:
        ref = [o.total(o.lee_richards(*tools.coil(int(n), 5000 + k))) for k, n in enumerate(sizes)]
        print(json.dumps({"atoms": int(na.item()), "expect_atoms": int(sizes.sum()),
                          "same": bool(np.array_equal(tot.numpy(), np.array(ref)))}))
    dist.barrier()
    dist.destroy_process_group()
""")


def _run_two_ranks(tmp_path, source):
    script = tmp_path / "worker.py"
    script.write_text(source)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script), ROOT],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])


def test_two_rank_lpt_sweep_gloo(tmp_path):
    res = _run_two_ranks(tmp_path, SWEEP_WORKER)
    assert res["same"] and res["atoms"] == res["expect_atoms"]


def test_partitions_cover_every_item_once_and_balance():
    import numpy as np
    from freesasa_amd import shard
    sizes = np.exp(np.random.default_rng(7).uniform(np.log(500), np.log(50000), 3000)).astype(np.int64)
    for parts in (shard.lpt(sizes, 8), shard.round_robin(len(sizes), 8)):
        assert sorted(np.concatenate(parts).tolist()) == list(range(len(sizes)))
    loads = np.array([sizes[p].sum() for p in shard.lpt(sizes, 8)])
    assert loads.max() - loads.mean() <= sizes.max()          # LPT bound
    assert loads.max() / loads.mean() < 1.001
    # degenerate inputs
    assert [len(p) for p in shard.lpt([5], 4)] == [1, 0, 0, 0]
    assert all(len(p) == 0 for p in shard.lpt([], 3))
    x, r, o = shard.gather_shard(np.arange(30.).reshape(10, 3), np.arange(10.), [0, 2, 5, 10], [2, 0])
    assert o.tolist() == [0, 5, 7] and r.tolist() == [5, 6, 7, 8, 9, 0, 1] and x.shape == (7, 3)


def test_contiguous_atom_balanced_cuts_of_the_single_process_multi_gpu_entry():
    """freesasa_gpu_shard_cuts (host-only part of freesasa_gpu_calc_batch_devices): contiguous runs,
    every structure in exactly one run, atom counts within one structure of equal."""
    import numpy as np
    import freesasa_amd as fa
    rng = np.random.default_rng(3)
    sizes = np.exp(rng.uniform(np.log(500), np.log(50000), 500)).astype(np.int64)
    offs = np.concatenate([[0], np.cumsum(sizes)])
    for parts in (1, 2, 3, 8):
        cuts = fa.shard_cuts(offs, parts)
        assert cuts[0] == 0 and cuts[-1] == len(sizes) and np.all(np.diff(cuts) >= 0)
        loads = np.array([offs[cuts[k + 1]] - offs[cuts[k]] for k in range(parts)])
        assert loads.sum() == offs[-1] and np.all(np.abs(loads - offs[-1] / parts) <= sizes.max())
    # fewer structures than parts: empty runs, nothing lost; empty structures are fine
    assert fa.shard_cuts(np.array([0, 10, 30]), 5).tolist()[-1] == 2
    cuts = fa.shard_cuts(np.array([0, 0, 5, 5, 9]), 2)
    assert cuts[0] == 0 and cuts[-1] == 4


def test_bench_py_two_ranks_dry_run():
    """bench.py itself, launched the way the driver launches it for N > 1 (torch.distributed.run, one process per
    rank), on CPU under gloo with --dry-run (a stand-in engine that computes nothing): ranks build disjoint shards,
    the timed region is bracketed by barriers, rank 0 prints ONE JSON line whose value is the atoms of ALL ranks x
    steps / the slowest rank's time, and every rank leaves the process group cleanly."""
    import json
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--structs", "5", "--atoms", "400", "--dry-run"]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=300, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    res = json.loads(lines[0])
    assert res["dry_run"] is True and res["n_gpus"] == 2 and res["steps"] == 3 and res["warmup"] == 1
    assert res["atoms_all_ranks"] == 2 * 5 * 400 and res["scaling"] == "weak" and res["unit"] == "atoms/s"
    assert abs(res["value"] - res["atoms_all_ranks"] * 3 / (res["ms_per_step"] * 3e-3)) < 1e-6 * res["value"]
    assert res["ms_per_step"] >= 2.0                                   # three steps of the stand-in's 2 ms each, at least
    for key in ("metric", "higher_is_better", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in res
    # round 6: with more than one rank the line also carries configs[3] / configs[4] through the real drivers, every rank
    # on its own device (here: the stand-ins), whole-job rates over the slowest rank's time
    multi = res["drivers_all_ranks"]
    assert set(multi) == {"trajectory_file", "sweep_cache"} and multi["trajectory_file"]["ranks"] == 2
    assert multi["trajectory_file"]["seconds"] >= 0.002 and multi["trajectory_file"]["unit"] == "atom-frames/s"
    # --gpus must agree with the launcher's world size
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--structs", "2", "--atoms", "100"],
                         capture_output=True, text=True, timeout=120, cwd=ROOT, env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK")})
    assert bad.returncode != 0 and "WORLD_SIZE" in bad.stderr


def test_bench_py_eight_ranks_dry_run_share_the_cpus_and_the_cache(tmp_path):
    """The driver's 8-GPU launch of bench.py, on CPU under gloo with --dry-run: eight ranks generate their (disjoint)
    shards at the same time - each with its share of the CPUs the cgroup grants -, write them to one cache directory
    (private name, then rename) and a second launch reads them back: one JSON line, the atoms of all eight ranks."""
    import json
    cache = tmp_path / "cache"
    cache.mkdir()
    res = []
    for attempt in range(2):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), FREESASA_AMD_BENCH_CACHE=str(cache))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1",
               "--structs", "250", "--atoms", "4000", "--dry-run"]
        out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
        assert out.returncode == 0, out.stderr[-2000:]
        lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1
        res.append(json.loads(lines[0]))
        files = sorted(f for f in os.listdir(cache) if not f.startswith("freesasa_amd_bench_u"))   # (not the drivers' scratch directory, round 6)
        assert len(files) == 8 and all(f.endswith(".npy") and ".tmp." not in f for f in files), files   # one per rank, no leftovers
    for r in res:
        assert r["dry_run"] is True and r["n_gpus"] == 8 and r["atoms_all_ranks"] == 8 * 250 * 4000 and r["scaling"] == "weak"


def test_bench_py_one_rank_dry_run_with_the_process_group_forced():
    """FREESASA_AMD_BENCH_FORCE_DIST=1: the N-rank code of bench.py (init_process_group, barrier, both all_reduces,
    destroy_process_group) at world size 1 - here on CPU under gloo, and on the MI355X under nccl in the test below."""
    import json
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), FREESASA_AMD_BENCH_FORCE_DIST="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--structs", "4", "--atoms", "300", "--dry-run"]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=300, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert res["dry_run"] is True and res["config"]["process_group"].startswith("gloo world 1")


@pytest.mark.gpu
def test_bench_py_one_rank_initialises_rccl_on_the_gpu():
    """Round-4 review, item 6: bench.py's multi-rank plumbing ran only for world > 1, i.e. never on hardware before the
    driver's 8-GPU run.  With FREESASA_AMD_BENCH_FORCE_DIST=1 one rank, launched through torch.distributed.run as the
    driver launches N, initialises the nccl (= RCCL) process group on the MI355X, passes the barrier and both
    all_reduces around the timed region and leaves the group; the line it prints says so, and carries the sustained
    figure next to the headline."""
    import json
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), FREESASA_AMD_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
           "--structs", "60", "--atoms", "5000", "--sustain-seconds", "0.5", "--no-live-counters", "--no-cpu-baseline",
           "--no-end-to-end", "--no-secondary", "--no-neighbors", "--no-drivers"]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    res = json.loads(lines[0])
    assert res["config"]["process_group"].startswith("nccl world 1")
    assert res["n_gpus"] == 1 and res["value"] > 1e7 and res["sustained"]["seconds"] >= 0.5 and res["sustained"]["value"] > 1e7
