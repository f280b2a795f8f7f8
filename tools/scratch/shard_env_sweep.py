"""bfs_sharded at world 1 under environment settings: argv = VAR=v1,v2,... [VAR2=...] [overlap=0|1|2]; every combination, 3 timed runs."""
import os, sys, time, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "ac-solver_amd"), ROOT]
import numpy as np, torch
import bench
from ac_solver.search.sharded import bfs_sharded
ak3 = bench.ak3_at_L()
axes = [(a.split("=")[0], a.split("=")[1].split(",")) for a in sys.argv[1:]]
for combo in itertools.product(*[v for _, v in axes]):
    ov = None
    late = None
    for (k, _), v in zip(axes, combo):
        if k == "overlap": ov = {"0": False, "1": "insert", "2": "commit"}[v]
        elif k == "late": late = bool(int(v))
        else: os.environ[k] = v
    ts = []
    for _ in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        bfs_sharded(ak3, 10**8, batch_parents=1 << 21, overlap=ov)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print(" ".join(f"{k}={v}" for (k, _), v in zip(axes, combo)), " ".join(f"{t:.2f}" for t in ts[1:]), "ms", flush=True)
