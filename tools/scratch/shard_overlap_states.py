"""bfs_sharded with and without the side stream in three process states (fresh, after bench.py's extras, after a fused search):
every run printed, so that a first-use cost shows as such."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "ac-solver_amd"), ROOT]
import numpy as np, torch
import bench
from ac_solver.search.sharded import bfs_sharded
from ac_solver.search._common import run_search
from ac_solver import _acx
ak3 = bench.ak3_at_L()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 5
def run(tag, **kw):
    ts = []
    for _ in range(N):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ok, path, st = bfs_sharded(ak3, 10**8, batch_parents=1 << 21, want_stats=True, **kw)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print(tag, kw, " ".join(f"{t:.2f}" for t in ts), "ms", flush=True)
run("fresh", overlap=True); run("fresh", overlap=False); run("fresh", overlap=True)
if "--extras" in sys.argv:
    pool = bench.ms_pool_at_L(25)
    ex = bench.extra_env_numbers(torch.device("cuda", 0), pool)
    print("extras done", flush=True)
    run("after extras", overlap=True); run("after extras", overlap=False)
run_search(_acx.SEARCH_BFS, ak3, 10**8, False)
run("after fused bfs", overlap=True); run("after fused bfs", overlap=False)
