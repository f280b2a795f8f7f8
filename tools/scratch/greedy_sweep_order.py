#!/usr/bin/env python3
"""experiment: the greedy sweep with the jobs in ORACLE order (each group's rows sorted by the duration a first run measured, longest first;
the groups themselves longest first) -- the upper bound of what any job-length prediction can buy"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "ac-solver_amd"), ROOT]
import numpy as np
from ac_solver import _acx
from ac_solver.search._common import run_search_groups
from ac_solver.search.miller_schupp.miller_schupp import generate_miller_schupp_presentations

groups = []
for n in range(1, 8):
    d = generate_miller_schupp_presentations(n, 7)
    groups.append(np.array([q for w in range(1, 8) for q in d[w]], dtype=np.int8))


def timed(gs, reps=5):
    ts = []
    res = None
    for _ in range(reps):
        t0 = time.perf_counter()
        res = run_search_groups(_acx.SEARCH_GREEDY, gs, 10**6, False)
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2], ts, res


base, ts, res = timed(groups)
print("given order:", round(base, 4), [round(t, 4) for t in ts])
work = [[(st["nodes"] if not ok else st["nodes"] * 0.5) for ok, _, st in r] for r in res]   # proxy: nodes (unsolved ones reach the budget)
secs = [[st["seconds"] for ok, _, st in r] for r in res]
print("per-search seconds available:", secs[2][:5])
_acx.check(_acx.lib.acx_set_option(_acx.OPT_GREEDY_KEEP_ORDER, 1))
t, ts, _ = timed(groups)
print("caller's order kept, given order:", round(t, 4), [round(x, 4) for x in ts])
for name, key in (("by nodes", work),):
    perm = [np.argsort(-np.array(k), kind="stable") for k in key]
    gs = [g[p] for g, p in zip(groups, perm)]
    t, ts, _ = timed(gs)
    print(f"rows longest first ({name}), groups n = 1..7:", round(t, 4), [round(x, 4) for x in ts])
    order = [2, 1, 3, 4, 0, 5, 6]  # narrow groups: n = 3, 2, 4, 5, 1 (the L = 20 / 18 unsolved ones are the longest)
    t, ts, _ = timed([gs[i] for i in order])
    print(f"rows longest first ({name}), groups n = 3, 2, 4, 5, 1, 6, 7:", round(t, 4), [round(x, 4) for x in ts])
