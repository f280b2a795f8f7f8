#!/usr/bin/env python3
"""bfs over the 1190 Miller-Schupp presentations vs the oracle: python tools/debug_sweep.py [budget] [cyclical 0/1] [single 0/1]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "ac-solver_amd"), ROOT]
import numpy as np
from ac_solver import _acx
from ac_solver.search._common import run_search, run_search_many
from oracle import ac_oracle as O
from tests.conftest import ms_pool_generator_order

budget = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10**6
cyc = bool(int(sys.argv[2])) if len(sys.argv) > 2 else True
single = bool(int(sys.argv[3])) if len(sys.argv) > 3 else False
g = json.load(open(os.path.join(ROOT, "tests/golden/ms_pool.json")))
pool = ms_pool_generator_order(g)
want = set(g["bfs_solved_order"])
got = {}
for lo in range(0, 1190, 170):
    rows = np.array(pool[lo:lo + 170], dtype=np.int8)
    res = [run_search(_acx.SEARCH_BFS, r, budget, cyc, True) for r in rows] if single else run_search_many(_acx.SEARCH_BFS, rows, budget, cyc, n_threads=16)
    for k, r in enumerate(res):
        got[lo + k] = r
solved = {k for k, r in got.items() if r[0]}
print("solved", len(solved), "want", len(want), "missing", sorted(want - solved)[:20], "extra", sorted(solved - want)[:20])
for k in sorted((want ^ solved))[:6]:
    wok, wpath, wst = O.bfs(np.array(pool[k], np.int8), budget, cyclically_reduce_after_moves=cyc, stats=True)
    ok, path, st = got[k]
    print(k, "L", len(pool[k]) // 2, "gpu", ok, st["nodes"], st["expanded"], "oracle", wok, wst["nodes"], wst["expanded"])
    ok1, path1, st1 = run_search(_acx.SEARCH_BFS, np.array(pool[k], np.int8), budget, cyc, True)
    print("   alone:", ok1, st1["nodes"], st1["expanded"], st1["levels"])
