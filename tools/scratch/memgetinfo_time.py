import time, torch
torch.cuda.init(); x = torch.empty(1 << 28, device="cuda"); torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter()
    for _ in range(100): torch.cuda.mem_get_info()
    print(f"hipMemGetInfo: {(time.perf_counter() - t0) * 1e4:.1f} us per call")
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "ac-solver_amd"), ROOT]
import numpy as np
from ac_solver import _acx
from ac_solver.search._common import run_search
ak3 = np.zeros(50, np.int8); ak3[:7] = [1, 1, 1, -2, -2, -2, -2]; ak3[25:31] = [1, 2, 1, -2, -1, -2]
run_search(_acx.SEARCH_BFS, ak3, 10**8, False)
for _ in range(4):
    t0 = time.perf_counter(); ok, path, st = run_search(_acx.SEARCH_BFS, ak3, 10**8, False); dt = time.perf_counter() - t0
    print(f"fused bfs: wall {dt*1e3:.2f} ms device {st['seconds']*1e3:.2f} ms")
