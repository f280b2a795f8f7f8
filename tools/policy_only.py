#!/usr/bin/env python3
"""acx_policy_sample alone at 131 072 environments: microseconds per call and MFMA TFLOP/s.  python tools/policy_only.py [n_env] [in_dim]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "ac-solver_amd"), ROOT]
from types import SimpleNamespace
import torch
from ac_solver.agents.ppo_agent import Agent
from ac_solver.agents.fused_policy import FusedPolicy

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 17
in_dim = int(sys.argv[2]) if len(sys.argv) > 2 else 50
dev = torch.device("cuda")
agent = Agent(SimpleNamespace(single_observation_space=SimpleNamespace(shape=(in_dim,)), single_action_space=SimpleNamespace(n=12)), [256, 256]).to(dev)
fp = FusedPolicy(agent, in_dim)
obs = torch.randint(-2, 3, (n, in_dim), device=dev).float()
act = torch.zeros(n, dtype=torch.int64, device=dev)
logp, val = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
for _ in range(5):
    fp.sample(obs, act, logp, val)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 50
e0.record()
for _ in range(reps):
    fp.sample(obs, act, logp, val)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / reps * 1e3
ks1 = (in_dim + 15) // 16
flops = 2 * 2 * n * (256 * 16 * ks1 + 256 * 256 + 32 * 256)  # MFMA work incl. padding, both networks
print(f"acx_policy_sample n={n} in={in_dim}: {us:.1f} us per call, {flops / us * 1e-6:.1f} TFLOP/s (padded), {n / us:.3e} env/us")
