"""bfs_sharded timing in the process state bench.py leaves behind (graph capture, extras), with and without the side stream."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "ac-solver_amd"), ROOT]
import numpy as np, torch
import bench
from ac_solver.search.sharded import bfs_sharded
from ac_solver.search._common import run_search
from ac_solver import _acx
ak3 = bench.ak3_at_L()
def run(tag, **kw):
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ok, path, st = bfs_sharded(ak3, 10**8, batch_parents=1 << 21, want_stats=True, **kw)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(tag, kw, f"{dt*1e3:.2f} ms setup {st['setup_seconds']*1e3:.2f} loop {st['loop_seconds']*1e3:.2f}", flush=True)
run("fresh"); run("fresh", overlap=False)
pool = bench.ms_pool_at_L(25)
dev = torch.device("cuda", 0)
ex = bench.extra_env_numbers(dev, pool)
print("extras done", flush=True)
run("after extras"); run("after extras", overlap=False)
run_search(_acx.SEARCH_BFS, ak3, 10**8, False)
run("after fused bfs"); run("after fused bfs", overlap=False)
