#!/usr/bin/env python3
"""bfs_sharded at world 1 (and thread ranks) on AK(3), L = 25: wall time per batch_parents, next to the fused search.
    python3 tools/shard_bench.py [budget] [bp_log2 ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "ac-solver_amd"), ROOT]
import numpy as np
import torch

from ac_solver import _acx
from ac_solver.search._common import run_search
from ac_solver.search.sharded import bfs_sharded

budget = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10**8
bps = [int(x) for x in sys.argv[2:]] or [18, 19, 20, 21, 22]
ak3 = np.zeros(50, np.int8)
ak3[:7] = [1, 1, 1, -2, -2, -2, -2]
ak3[25:31] = [1, 2, 1, -2, -1, -2]
run_search(_acx.SEARCH_BFS, ak3, budget, False)
t0 = time.perf_counter()
ok, path, st = run_search(_acx.SEARCH_BFS, ak3, budget, False)
print(f"fused acx_search: {time.perf_counter() - t0:.4f} s wall, {st['seconds']:.4f} s device, nodes {st['nodes']} expanded {st['expanded']}", flush=True)
want = (st["nodes"], st["expanded"])
for bp in bps:
    bfs_sharded(ak3, budget, batch_parents=1 << bp)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        ok, path, s2 = bfs_sharded(ak3, budget, batch_parents=1 << bp, want_stats=True)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    flag = "" if (s2["nodes"], s2["expanded"]) == want else "  MISMATCH"
    print(f"bfs_sharded world 1, 2^{bp} parents per chunk: {best * 1e3:.2f} ms, {s2['nodes'] / best:.3e} nodes/s, chunks {s2['chunks']} levels {s2['levels']} "
          f"nodes {s2['nodes']} expanded {s2['expanded']}{flag}", flush=True)

# thread ranks sharing the one GPU (tests/shard_helpers.py: ThreadComm): what the orchestration costs per rank when the
# kernels of all ranks run on the same device (no hardware scaling number: a box has one GPU)
if os.environ.get("ACX_SHARD_THREADS"):
    from tests.shard_helpers import run_threads

    for world in [int(x) for x in os.environ["ACX_SHARD_THREADS"].split(",")]:
        def work(comm):
            bfs_sharded(ak3, budget, comm=comm, batch_parents=1 << 21)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ok, path, st = bfs_sharded(ak3, budget, comm=comm, batch_parents=1 << 21, want_stats=True)
            torch.cuda.synchronize()
            return time.perf_counter() - t0, st
        res = run_threads(world, work)
        dt = max(r[0] for r in res)
        st = res[0][1]
        flag = "" if (st["nodes"], st["expanded"]) == want else "  MISMATCH"
        print(f"bfs_sharded, {world} thread ranks on one GPU: {dt * 1e3:.1f} ms, nodes {st['nodes']} expanded {st['expanded']} chunks {st['chunks']}{flag}", flush=True)
