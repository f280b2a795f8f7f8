#!/usr/bin/env python3
"""Does a live RCCL process group slow down the overlapped searches (CPU quota)?  python tools/sweep_with_pg.py [pg] [threads]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "ac-solver_amd"), ROOT]
import numpy as np
import torch
from ac_solver import _acx
from ac_solver.search._common import run_search_many

def throttled():
    try:
        return {k: int(v) for k, v in (l.split() for l in open("/sys/fs/cgroup/cpu.stat"))}
    except Exception:
        return {}

use_pg = len(sys.argv) > 1 and sys.argv[1] == "pg"
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 16
torch.zeros(1, device="cuda")
if use_pg:
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29534")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    t = torch.ones(4, device="cuda"); dist.all_reduce(t); torch.cuda.synchronize()
g = json.load(open(os.path.join(ROOT, "tests/golden/ms_pool.json")))
pool = []
for n in range(1, 8):
    for w in range(1, 8):
        pool += g["by_n"][str(n)][str(w)]
s0 = throttled()
t0 = time.perf_counter()
for n in (1, 2, 3):
    rows = np.array(pool[(n - 1) * 170:n * 170], dtype=np.int8)
    run_search_many(_acx.SEARCH_BFS, rows, 10**6, True, n_threads=threads)
dt = time.perf_counter() - t0
s1 = throttled()
print(f"pg={use_pg} threads={threads}: {dt:.2f}s  throttled periods +{s1.get('nr_throttled', 0) - s0.get('nr_throttled', 0)} "
      f"throttled_usec +{s1.get('throttled_usec', 0) - s0.get('throttled_usec', 0)} usage_usec +{s1.get('usage_usec', 0) - s0.get('usage_usec', 0)}")
