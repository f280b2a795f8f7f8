#!/usr/bin/env python3
"""the greedy sweep repeated with a pause between the calls: is the slow start of the later calls device work left over from the call before?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "ac-solver_amd"), ROOT]
import json
import numpy as np
from ac_solver import _acx
from ac_solver.search._common import run_search_groups
g = json.load(open(os.path.join(ROOT, "tests/golden/ms_pool.json")))
groups = [np.array([p for w in range(1, 8) for p in g["by_n"][str(n)][str(w)]], dtype=np.int8) for n in range(1, 8)]
pause = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
keep = len(sys.argv) > 2 and sys.argv[2] == "drop"
for rep in range(7):
    t0 = time.perf_counter()
    res = run_search_groups(_acx.SEARCH_GREEDY, groups, 10**6, False)
    dt = time.perf_counter() - t0
    if keep:
        res = None
    print(f"pause {pause}: run {rep}: {dt:.3f} s", flush=True)
    time.sleep(pause)
