#!/usr/bin/env python3
"""Timing of the device frontier: acx_search (single GPU) and bfs_sharded (world 1) vs the CPU oracle."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "ac-solver_amd"), ROOT]
import numpy as np
import torch

from ac_solver import _acx
from ac_solver.search._common import run_search
from ac_solver.search.sharded import bfs_sharded
from oracle import ac_oracle as O

ak3 = np.zeros(50, np.int8)
ak3[:7] = [1, 1, 1, -2, -2, -2, -2]
ak3[25:31] = [1, 2, 1, -2, -1, -2]

budgets = [int(b) for b in sys.argv[1:]] or [100000, 1000000]
for b in budgets:
    for kind, name in ((_acx.SEARCH_BFS, "bfs"), (_acx.SEARCH_GREEDY, "greedy")):
        run_search(kind, ak3, 1000, False)
        t0 = time.perf_counter()
        ok, path, st = run_search(kind, ak3, b, False)
        dt = time.perf_counter() - t0
        print(f"acx_search {name:6s} budget={b}: nodes={st['nodes']} expanded={st['expanded']} batches={st['levels']} wall={dt:.3f}s dev={st['seconds']:.3f}s "
              f"-> {st['nodes'] / dt:.3e} nodes/s", flush=True)
    t0 = time.perf_counter()
    ok, path, st = bfs_sharded(ak3, b, want_stats=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"bfs_sharded(world=1) budget={b}: nodes={st['nodes']} wall={dt:.3f}s -> {st['nodes'] / dt:.3e} nodes/s", flush=True)
    if b <= 2000000:
        for fn, name in ((O.bfs, "bfs"), (O.greedy_search, "greedy")):
            t0 = time.perf_counter()
            ok, path, st = fn(ak3, b, stats=True)
            dt = time.perf_counter() - t0
            print(f"oracle     {name:6s} budget={b}: nodes={st['nodes']} wall={dt:.3f}s -> {st['nodes'] / dt:.3e} nodes/s", flush=True)
