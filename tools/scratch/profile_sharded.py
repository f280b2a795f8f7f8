#!/usr/bin/env python3
"""Where does bfs_sharded spend its time?  Thread-simulated ranks on one GPU; per-phase wall time with syncs."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "ac-solver_amd"), ROOT]
import numpy as np
import torch
import ac_solver.search.sharded as sh
from tests.shard_helpers import run_threads

ak3 = np.zeros(50, np.int8); ak3[:7] = [1, 1, 1, -2, -2, -2, -2]; ak3[25:31] = [1, 2, 1, -2, -1, -2]
budget = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10**8
world = int(sys.argv[2]) if len(sys.argv) > 2 else 1
bp = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 20

def run(comm):
    sh.bfs_sharded(ak3, budget, comm=comm, batch_parents=bp)  # warm-up at full size: the first run pays for every device allocation
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ok, path, st = sh.bfs_sharded(ak3, budget, comm=comm, batch_parents=bp, want_stats=True)
    torch.cuda.synchronize()
    return time.perf_counter() - t0, st

if world == 1:
    from torch.profiler import profile, ProfilerActivity
    dt, st = run(sh.SingleComm())
    print(f"world=1 budget={budget}: {dt:.3f}s {st['nodes'] / dt:.3e} nodes/s levels={st['levels']}")
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        run(sh.SingleComm())
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=18, max_name_column_width=60))
else:
    res = run_threads(world, run)
    dt = max(r[0] for r in res)
    print(f"world={world} (threads on one GPU) budget={budget}: {dt:.3f}s {res[0][1]['nodes'] / dt:.3e} nodes/s")
    if len(sys.argv) > 4:
        from torch.profiler import profile, ProfilerActivity
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
            run_threads(world, run)
        print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=22, max_name_column_width=60))
