#!/usr/bin/env python3
"""How many of the n=1 Miller-Schupp presentations does bfs solve as the budget grows? (the reference publishes 120 of 170)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "ac-solver_amd"), ROOT]
import numpy as np
from ac_solver import _acx
from ac_solver.search._common import run_search_many
g = json.load(open(os.path.join(ROOT, "tests/golden/ms_pool.json")))
pool = []
for n in range(1, 8):
    for w in range(1, 8):
        pool += g["by_n"][str(n)][str(w)]
want = set(g["bfs_solved_order"])
nsel = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rows = np.array(pool[(nsel - 1) * 170:nsel * 170], dtype=np.int8)
for budget in [int(float(b)) for b in sys.argv[2:]]:
    t0 = time.perf_counter()
    res = run_search_many(_acx.SEARCH_BFS, rows, budget, bool(int(os.environ.get("CYC", "0"))), n_threads=8)
    solved = {(nsel - 1) * 170 + k for k, r in enumerate(res) if r[0]}
    pub = {k for k in want if (nsel - 1) * 170 <= k < nsel * 170}
    print(f"n={nsel} budget={budget}: {len(solved)} solved (published {len(pub)}), subset={solved <= pub}, equal={solved == pub}, {time.perf_counter() - t0:.1f}s", flush=True)
