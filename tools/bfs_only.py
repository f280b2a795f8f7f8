#!/usr/bin/env python3
"""One fused BFS (acx_search) on AK(3) at L=25 for profiling: python tools/bfs_only.py [budget]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "ac-solver_amd"), ROOT]
import numpy as np

from ac_solver import _acx
from ac_solver.search._common import run_search

ak3 = np.zeros(50, np.int8)
ak3[:7] = [1, 1, 1, -2, -2, -2, -2]
ak3[25:31] = [1, 2, 1, -2, -1, -2]
budget = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10**8
run_search(_acx.SEARCH_BFS, ak3, 20000, False)
ok, path, st = run_search(_acx.SEARCH_BFS, ak3, budget, False)
print(f"bfs budget={budget}: nodes={st['nodes']} expanded={st['expanded']} batches={st['levels']} dev={st['seconds']:.4f}s -> {st['nodes'] / st['seconds']:.3e} nodes/s")
