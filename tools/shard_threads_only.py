#!/usr/bin/env python3
"""bfs_sharded on AK(3) with W thread ranks sharing the one GPU of a box (tests/shard_helpers.py: ThreadComm), nothing else: for
`rocprofv3 --kernel-trace --stats`, whose k_shard_* totals / (searches) are the device work of ALL ranks of one search at world W.
    python3 tools/shard_threads_only.py W [budget] [searches] [log2 of the global parents per chunk]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "ac-solver_amd"), ROOT]
import numpy as np
import torch

from ac_solver.search.sharded import bfs_sharded
from tests.shard_helpers import run_threads

world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
budget = int(float(sys.argv[2])) if len(sys.argv) > 2 else 10**8
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
bp = 1 << (int(sys.argv[4]) if len(sys.argv) > 4 else 21)
ak3 = np.zeros(50, np.int8)
ak3[:7] = [1, 1, 1, -2, -2, -2, -2]
ak3[25:31] = [1, 2, 1, -2, -1, -2]


def work(comm):
    out = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ok, path, st = bfs_sharded(ak3, budget, comm=comm, batch_parents=bp, want_stats=True)
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0, st["nodes"], st["expanded"], st["local_nodes"], st.get("region_fill_all"), st["chunks"]))
    return out


res = run_threads(world, work) if world > 1 else [work(None)]
for k in range(reps):
    local = [r[k][3] for r in res]
    print(f"world {world} search {k}: {max(r[k][0] for r in res) * 1e3:.1f} ms wall (thread ranks on one GPU), nodes {res[0][k][1]} expanded {res[0][k][2]}, "
          f"chunks {res[0][k][5]}, local nodes max/mean {max(local) * len(local) / sum(local):.3f}, region fill per level (1/256 of the even share) {res[0][k][4]}", flush=True)
