import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "ac-solver_amd"), ROOT]
import numpy as np
from ac_solver import _acx
from ac_solver.search._common import run_search
ak3 = np.zeros(50, np.int8); ak3[:7] = [1, 1, 1, -2, -2, -2, -2]; ak3[25:31] = [1, 2, 1, -2, -1, -2]
for k in range(5):
    t0 = time.perf_counter()
    ok, path, st = run_search(_acx.SEARCH_BFS, ak3, 10**8, False)
    print(f"bfs 1e8 run {k}: nodes={st['nodes']} dev={st['seconds']*1e3:.2f} ms wall={(time.perf_counter()-t0)*1e3:.2f} ms", flush=True)
