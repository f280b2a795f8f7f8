#!/usr/bin/env python3
"""In-process repeat of the bfs sweep over the 1190 Miller-Schupp presentations (warm vs first call), per-group kernel time."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "ac-solver_amd"), ROOT]
import numpy as np
from ac_solver import _acx
from ac_solver.search._common import run_search_groups, run_search_many
g = json.load(open(os.path.join(ROOT, "tests/golden/ms_pool.json")))
pool = []
for n in range(1, 8):
    for w in range(1, 8):
        pool += g["by_n"][str(n)][str(w)]
groups = [np.array(pool[(n - 1) * 170:n * 170], dtype=np.int8) for n in range(1, 8)]
budget = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10**6
for rep in range(3):
    t0 = time.perf_counter()
    res = run_search_groups(_acx.SEARCH_BFS, groups, budget, True)
    dt = time.perf_counter() - t0
    nodes = sum(st["nodes"] for r in res for _, _, st in r)
    kern = [max(st["seconds"] for _, _, st in r) for r in res]
    print(f"together rep {rep}: {dt:.3f}s {nodes / dt:.3e} nodes/s; group kernel seconds {[round(k, 3) for k in kern]}")
for rep in range(2):
    t0 = time.perf_counter()
    res = [run_search_many(_acx.SEARCH_BFS, gr, budget, True) for gr in groups]
    dt = time.perf_counter() - t0
    kern = [max(st["seconds"] for _, _, st in r) for r in res]
    print(f"one by one rep {rep}: {dt:.3f}s; group kernel seconds {[round(k, 3) for k in kern]}")
