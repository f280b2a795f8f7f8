#!/usr/bin/env python3
"""bfs / greedy over the 1190 Miller-Schupp presentations, all seven widths in flight, timed on the second run (device pools warm).
python tools/ms_sweep_warm.py [bfs|greedy] [budget] [greedy slots]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "ac-solver_amd"), ROOT]
import numpy as np
from ac_solver import _acx
from ac_solver.search._common import run_search_groups
g = json.load(open(os.path.join(ROOT, "tests/golden/ms_pool.json")))
algo = sys.argv[1] if len(sys.argv) > 1 else "bfs"
budget = int(float(sys.argv[2])) if len(sys.argv) > 2 else 10**6
slots = int(sys.argv[3]) if len(sys.argv) > 3 else -1  # ACX_OPT_GREEDY_SLOTS (-1: the library's default)
_acx.check(_acx.lib.acx_set_option(_acx.OPT_GREEDY_SLOTS, slots))
kind, cyc = (_acx.SEARCH_BFS, True) if algo == "bfs" else (_acx.SEARCH_GREEDY, False)
groups = [np.array([p for w in range(1, 8) for p in g["by_n"][str(n)][str(w)]], dtype=np.int8) for n in range(1, 8)]
for rep in range(int(os.environ.get("REPS", "3"))):
    t0 = time.perf_counter()
    res = run_search_groups(kind, groups, budget, cyc)
    dt = time.perf_counter() - t0
    solved = sum(ok for r in res for ok, _, _ in r)
    nodes = sum(st["nodes"] for r in res for _, _, st in r)
    print(f"{algo} run {rep}: {solved} solved, {nodes} nodes in {dt:.3f} s = {nodes / dt:.3e} nodes/s", flush=True)
