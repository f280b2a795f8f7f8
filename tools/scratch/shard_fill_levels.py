"""The adaptive region capacity level by level (bfs_sharded with thread ranks on one GPU) for a few presentations: how the fullest
region of a level moves from level to level decides how much margin the next level's capacity needs."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "ac-solver_amd"), ROOT]
import numpy as np, torch
import bench
import ac_solver.search.sharded as sh
from tests.shard_helpers import run_threads
g = json.load(open(os.path.join(ROOT, "tests/golden/ms_pool.json")))
pool = []
for n in range(1, 8):
    for w in range(1, 8):
        pool += g["by_n"][str(n)][str(w)]
cases = [("AK(3) L=25", bench.ak3_at_L(), 3 * 10**7, False)] + [(f"MS[{i}]", np.array(pool[i], np.int8), 10**7, c) for i, c in ((1100, False), (700, False), (300, True), (900, True))]
world = int(sys.argv[1]) if len(sys.argv) > 1 else 4
for name, p, budget, cyc in cases:
    def work(comm):
        return sh.bfs_sharded(p, budget, cyclically_reduce_after_moves=cyc, comm=comm, batch_parents=1 << 19, want_stats=True)
    res = run_threads(world, work)
    ok, path, st = res[0]
    print(name, "cyclical" if cyc else "", "solved" if ok else "", "nodes", st["nodes"], "levels", st["levels"], "fill_q8 per level (capacity of the next level = 1.3 x the fullest region of this one; 320 = the default):", st["region_fill_all"], flush=True)
