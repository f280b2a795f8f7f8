#!/usr/bin/env python3
"""the greedy sweep three times with ACX_DEBUG=1: the library's own timeline of the third call (slots from the pool, clean)"""
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
for rep in range(int(os.environ.get("REPS", "3"))):
    print(f"---- call {rep}", file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    run_search_groups(_acx.SEARCH_GREEDY, groups, 10**6, False)
    print(f"---- call {rep}: {time.perf_counter() - t0:.4f} s", file=sys.stderr, flush=True)
