"""bfs_sharded at world 1: resident-workgroup bound of k_shard_insert (ACX_SHARD_INSERT_WGS) x side stream on/off."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "ac-solver_amd"), ROOT]
import numpy as np, torch
import bench
from ac_solver.search.sharded import bfs_sharded
ak3 = bench.ak3_at_L()
caps = [int(x) for x in sys.argv[1:]] or [0, 512, 768, 1024, 1536, 2048]
for cap in caps:
    os.environ["ACX_SHARD_INSERT_WGS"] = str(cap)
    for ov in (False, True):
        ts = []
        for _ in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            ok, path, st = bfs_sharded(ak3, 10**8, batch_parents=1 << 21, want_stats=True, overlap=ov)
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        print(f"cap {cap:5d} overlap {ov!s:5}", " ".join(f"{t:.2f}" for t in ts[1:]), "ms  nodes", st["nodes"], flush=True)
