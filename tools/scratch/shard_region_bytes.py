"""Exchanged bytes of bfs_sharded with thread ranks on one GPU: adaptive region capacity against the safe default."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "ac-solver_amd"), ROOT]
import numpy as np, torch
import bench
from ac_solver.search.sharded import bfs_sharded, FILL_DEFAULT
from tests.shard_helpers import run_threads
ak3 = bench.ak3_at_L()
budget = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10**8
for world in [int(x) for x in sys.argv[2:]] or [8]:
    for fill in (FILL_DEFAULT, None):
        def work(comm):
            return bfs_sharded(ak3, budget, comm=comm, batch_parents=1 << 21, want_stats=True, region_fill=fill)
        res = run_threads(world, work)
        st = res[0][2]
        keys = sorted(k for k in st if k.startswith("comm_") or k.startswith("region"))
        print(f"world {world} region_fill {fill}: nodes {st['nodes']} chunks {st['chunks']}", {k: st[k] for k in keys}, flush=True)
