#!/usr/bin/env python3
"""Diagnostic: per-chunk shader-clock stamps of acx_policy_sample built with -DACX_POLICY_STAMP (ACX_LIB=...)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "ac-solver_amd"), ROOT]
from types import SimpleNamespace
import torch
from ac_solver.agents.ppo_agent import Agent
from ac_solver.agents.fused_policy import FusedPolicy
n, in_dim = 1 << 17, 50
dev = torch.device("cuda")
agent = Agent(SimpleNamespace(single_observation_space=SimpleNamespace(shape=(in_dim,)), single_action_space=SimpleNamespace(n=12)), [256, 256]).to(dev)
fp = FusedPolicy(agent, in_dim)
obs = torch.randint(-2, 3, (n, in_dim), device=dev).float()
act = torch.zeros(n, dtype=torch.int64, device=dev)
logp, val = torch.zeros(n + 1024, device=dev), torch.zeros(n, device=dev)
for _ in range(20):
    fp.sample(obs, act, logp, val)
torch.cuda.synchronize()
st = logp[n + 512:n + 512 + 8 * 64].view(8, 64).cpu()  # a block with odd index
for w in (0, 1, 4, 7):
    row = st[w]
    print("wave", w, "stamps:", [int(x) for x in row[:22]], "end", int(row[63]))
    d = [int(row[i + 1] - row[i]) for i in range(21)]
    print("   deltas:", d)

