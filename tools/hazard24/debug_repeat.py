#!/usr/bin/env python3
"""Repeat one bfs and print its stats each time: python tools/hazard24/debug_repeat.py pool_index budget cyclical repeats"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "ac-solver_amd"), ROOT]
import numpy as np
from ac_solver import _acx
from ac_solver.search._common import run_search
from oracle import ac_oracle as O
from tests.conftest import ms_pool_generator_order
k, budget, cyc, reps = int(sys.argv[1]), int(float(sys.argv[2])), bool(int(sys.argv[3])), int(sys.argv[4])
pool = ms_pool_generator_order(json.load(open(os.path.join(ROOT, "tests/golden/ms_pool.json"))))
p = np.array(pool[k], np.int8)
w = O.bfs(p, budget, cyclically_reduce_after_moves=cyc, stats=True)
print("oracle", w[0], w[2]["nodes"], w[2]["expanded"], "path len", len(w[1] or []))
for _ in range(reps):
    ok, path, st = run_search(_acx.SEARCH_BFS, p, budget, cyc, True)
    print("gpu   ", ok, st["nodes"], st["expanded"], st["levels"], "path len", len(path or []), "same path" if path == w[1] else "PATH DIFFERS")
