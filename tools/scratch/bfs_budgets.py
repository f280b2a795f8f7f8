"""bfs on AK(3) for a list of budgets: nodes / expanded next to the oracle's."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "ac-solver_amd"), ROOT]
import numpy as np
from ac_solver import _acx
from ac_solver.search._common import run_search
from oracle import ac_oracle as O
ak3 = np.zeros(50, np.int8); ak3[:7] = [1, 1, 1, -2, -2, -2, -2]; ak3[25:31] = [1, 2, 1, -2, -1, -2]
if os.environ.get("PRES") == "ak2": ak3 = np.array([1, 1, -2, -2, -2, 0, 0, 1, 2, 1, -2, -1, -2, 0], np.int8)
for b in [int(x) for x in sys.argv[1:]]:
    ok, path, st = run_search(_acx.SEARCH_BFS, ak3, b, False)
    wok, wpath, wst = O.bfs(ak3, b, cyclically_reduce_after_moves=False, stats=True)
    print(b, st["nodes"], wst["nodes"], st["expanded"], wst["expanded"], "OK" if st["nodes"] == wst["nodes"] else "MISMATCH", flush=True)
