#!/usr/bin/env python3
"""where the wall time of run_search_groups goes: the C call (acx_search_groups) and the conversion of its output arrays (_collect)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "ac-solver_amd"), ROOT]
import numpy as np
from ac_solver import _acx
from ac_solver.search import _common
from ac_solver.search.miller_schupp.miller_schupp import generate_miller_schupp_presentations

groups = []
for n in range(1, 8):
    d = generate_miller_schupp_presentations(n, 7)
    groups.append(np.array([q for w in range(1, 8) for q in d[w]], dtype=np.int8))
orig = _common._collect
spent = []
def timed_collect(*a, **k):
    t0 = time.perf_counter()
    out = orig(*a, **k)
    spent.append(time.perf_counter() - t0)
    return out
_common._collect = timed_collect
for kind, cyc, name in ((_acx.SEARCH_BFS, True, "bfs"), (_acx.SEARCH_GREEDY, False, "greedy")):
    for rep in range(4):
        spent.clear()
        t0 = time.perf_counter()
        _common.run_search_groups(kind, groups, 10**6, cyc)
        dt = time.perf_counter() - t0
        print(f"{name}: total {dt * 1e3:.1f} ms, _collect {sum(spent) * 1e3:.1f} ms", flush=True)
