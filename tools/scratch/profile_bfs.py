#!/usr/bin/env python3
"""One bfs on AK(3)@L=25 through acx_search (for rocprofv3 kernel traces)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "ac-solver_amd"), ROOT]
import numpy as np
from ac_solver import _acx
from ac_solver.search._common import run_search
ak3 = np.zeros(50, np.int8); ak3[:7] = [1, 1, 1, -2, -2, -2, -2]; ak3[25:31] = [1, 2, 1, -2, -1, -2]
budget = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2 * 10**7
kind = _acx.SEARCH_GREEDY if len(sys.argv) > 2 and sys.argv[2] == "greedy" else _acx.SEARCH_BFS
run_search(kind, ak3, 1000, False)
ok, path, st = run_search(kind, ak3, budget, False)
print(st)
