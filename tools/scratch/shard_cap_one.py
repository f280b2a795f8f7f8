import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "ac-solver_amd"), ROOT]
import numpy as np, torch
import bench
from ac_solver.search.sharded import bfs_sharded
ak3 = bench.ak3_at_L()
for spec in sys.argv[1:]:
    cap, ov = spec.split(":")
    os.environ["ACX_SHARD_INSERT_WGS"] = cap
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        bfs_sharded(ak3, 10**8, batch_parents=1 << 21, overlap={'0': False, '1': 'insert', '2': 'commit'}[ov])
        torch.cuda.synchronize(); print(spec, f"{(time.perf_counter() - t0) * 1e3:.2f} ms", flush=True)
