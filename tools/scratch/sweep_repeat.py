"""The 1190-presentation bfs (or greedy) sweep N times in one process (pools warm after the first): seconds per sweep."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "ac-solver_amd"), ROOT]
import numpy as np
from ac_solver import _acx
from ac_solver.search._common import run_search_groups
g = json.load(open(os.path.join(ROOT, "tests/golden/ms_pool.json")))
pool = []
for n in range(1, 8):
    for w in range(1, 8):
        pool += g["by_n"][str(n)][str(w)]
algo = sys.argv[1] if len(sys.argv) > 1 else "bfs"
budget = int(float(sys.argv[2])) if len(sys.argv) > 2 else 10**6
cyc = bool(int(sys.argv[3])) if len(sys.argv) > 3 else True
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 4
kind = _acx.SEARCH_BFS if algo == "bfs" else _acx.SEARCH_GREEDY
groups = [np.array(pool[(n - 1) * 170:n * 170], dtype=np.int8) for n in range(1, 8)]
for r in range(reps):
    t0 = time.perf_counter()
    res = run_search_groups(kind, groups, budget, cyc, n_threads=16)
    dt = time.perf_counter() - t0
    nodes = sum(st["nodes"] for grp in res for (_, _, st) in grp)
    solved = sum(1 for grp in res for (ok, _, _) in grp if ok)
    print(f"{algo} sweep {r}: {dt:.3f} s, {nodes / dt:.3e} nodes/s, solved {solved}", flush=True)
