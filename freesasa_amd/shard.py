"""Sharding of independent structures (or trajectory frames) over the GPUs of a node.

The hot path has no exchange step: a structure is a closed problem (neighbors never cross
structures, ref: src/nb.c:524-557 builds one cell list per call), so multi-GPU is a partition of the
work list and nothing else — no collective on the data path.  Equal-size batches
are dealt round-robin; skewed sweeps (whole-PDB: 500 ... 50 000 atoms) use longest-processing-time
first on the atom count, which is what the kernel time is proportional to.
"""
import heapq

import numpy as np


def round_robin(n_items, n_parts):
    """Item k goes to part k % n_parts.  Returns a list of int64 index arrays."""
    idx = np.arange(n_items, dtype=np.int64)
    return [idx[p::n_parts] for p in range(n_parts)]


def lpt(sizes, n_parts):
    """Longest-processing-time-first: items by descending size, each to the currently lightest part.
    Deterministic (ties by index).  Returns a list of int64 index arrays, each ascending.
    The heaviest part exceeds the mean load by at most the largest item."""
    sizes = np.asarray(sizes, dtype=np.int64)
    order = np.lexsort((np.arange(sizes.size), -sizes))
    heap = [(0, p) for p in range(n_parts)]
    parts = [[] for _ in range(n_parts)]
    for k in order:
        load, p = heapq.heappop(heap)
        parts[p].append(int(k))
        heapq.heappush(heap, (load + int(sizes[k]), p))
    return [np.array(sorted(p), dtype=np.int64) for p in parts]


def gather_shard(xyz, radii, offsets, items):
    """Concatenate the structures `items` of a CSR batch: (xyz, radii, offsets) of the shard."""
    offsets = np.asarray(offsets, dtype=np.int64)
    xyz = np.asarray(xyz, dtype=np.float64).reshape(-1, 3)
    sel = [np.arange(offsets[k], offsets[k + 1]) for k in items]
    rows = np.concatenate(sel) if sel else np.zeros(0, dtype=np.int64)
    lens = np.array([offsets[k + 1] - offsets[k] for k in items], dtype=np.int64)
    return xyz[rows], np.asarray(radii, dtype=np.float64)[rows], np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
