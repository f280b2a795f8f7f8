#!/usr/bin/env python3
"""Wall time of train_ppo at BASELINE config 5's per-GPU shape (131 072 envs): rollout vs. the torch f32 update.
python tools/train_probe.py [num_steps] [updates]"""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "ac-solver_amd"), ROOT]
import torch
from ac_solver.agents.ppo import train_ppo
T = int(sys.argv[1]) if len(sys.argv) > 1 else 32
U = int(sys.argv[2]) if len(sys.argv) > 2 else 12
os.chdir(tempfile.mkdtemp())
N = 131072
def run(updates):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    train_ppo(["--num-envs", str(N), "--num-steps", str(T), "--total-timesteps", str(updates * T * N), "--tile-initial-states", "--fused-policy",
               "--horizon-length", "200", "--num-minibatches", "4"])
    torch.cuda.synchronize()
    return time.perf_counter() - t0

run(2)  # warm-up: dataset files, allocations, kernels
# (the fixed cost of a train_ppo call -- files, allocations -- varies by a few hundred ms between calls: the difference of two runs only
# says something when it spans many updates; the smaller of two samples of each)
a, b = min(run(2), run(2)), min(run(2 + U), run(2 + U))
print(f"2 updates {a:.2f} s, {2 + U} updates {b:.2f} s")
print(f"per update (rollout of {T} steps x {N} envs + PPO update): {(b - a) / U * 1e3:.0f} ms = {(T * N) / ((b - a) / U):.3e} env-steps/s end to end")
