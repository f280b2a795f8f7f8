"""Does bfs_sharded slow down after other torch work in the process (bench.py order)?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "ac-solver_amd"), ROOT]
import numpy as np, torch
from ac_solver.search.sharded import bfs_sharded
from ac_solver.envs.vec_env import ACVecEnv
from bench import ms_pool_at_L
ak3 = np.zeros(50, np.int8); ak3[:7] = [1, 1, 1, -2, -2, -2, -2]; ak3[25:31] = [1, 2, 1, -2, -1, -2]
def run(tag):
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ok, path, st = bfs_sharded(ak3, 10**8, batch_parents=1 << 21, want_stats=True)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(tag, f"{dt*1e3:.2f} ms setup {st['setup_seconds']*1e3:.2f} loop {st['loop_seconds']*1e3:.2f}", flush=True)
run("fresh")
pool = ms_pool_at_L(25)
env = ACVecEnv(pool[np.arange(1 << 22) % len(pool)], horizon_length=1000, record_actions=False, final_info=False)
x = torch.empty((8, 1 << 22, 50), dtype=torch.int8, device="cuda"); del x, env
run("after big torch tensors")
g = torch.cuda.CUDAGraph()
y = torch.zeros(1000, device="cuda")
with torch.cuda.graph(g):
    y += 1
g.replay(); torch.cuda.synchronize()
run("after a graph capture")
