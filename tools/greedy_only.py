"""greedy_search on AK(3) at a given budget, N times; ACX_DEBUG=1 prints the frontier kernel's own counters."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "ac-solver_amd"), ROOT]
from ac_solver import _acx
from ac_solver.search._common import run_search
budget = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10**7
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cyc = bool(int(sys.argv[3])) if len(sys.argv) > 3 else False
ak3 = np.zeros(50, np.int8); ak3[:7] = [1, 1, 1, -2, -2, -2, -2]; ak3[25:31] = [1, 2, 1, -2, -1, -2]
for _ in range(reps):
    ok, path, st = run_search(_acx.SEARCH_GREEDY, ak3, budget, cyc)
    print(ok, st["nodes"], st["expanded"], st["levels"], f"{st['seconds']*1e3:.2f} ms", f"{st['nodes']/st['seconds']:.3e} nodes/s", flush=True)
