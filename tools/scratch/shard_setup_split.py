"""bfs_sharded at world 1: setup / loop split of the wall time (stats), a few runs."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "ac-solver_amd"), ROOT]
import numpy as np, torch
import bench
from ac_solver.search.sharded import bfs_sharded
ak3 = bench.ak3_at_L()
for ov in (False, False, "commit"):
    for _ in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ok, path, st = bfs_sharded(ak3, 10**8, batch_parents=1 << 21, want_stats=True, overlap=ov)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(ov, f"total {dt*1e3:.2f} setup {st['setup_seconds']*1e3:.2f} loop {st['loop_seconds']*1e3:.2f} chunks {st['chunks']}", flush=True)
