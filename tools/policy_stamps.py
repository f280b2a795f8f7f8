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
logp, val = torch.zeros(n + 1024 + 4 * 512, device=dev), torch.zeros(n, device=dev)
for _ in range(20):
    fp.sample(obs, act, logp, val)
torch.cuda.synchronize()
st = logp[n + 512:n + 512 + 8 * 64].view(8, 64).cpu()  # a block with odd index
for w in (0, 4, 7):
    row = st[w]
    for t, (lo, end) in enumerate(((0, 29), (30, 63))):  # a workgroup's first and second tile of 256 environments
        print("wave", w, "tile", t, "stamps:", [int(x) for x in row[lo:lo + 26]], "end", int(row[end]))
        print("   deltas:", [int(row[lo + i + 1] - row[lo + i]) for i in range(25)])


# where the workgroups sit inside the launch (100 MHz counter shared by the chip): entry, first tile staged, heads of the last tile done, end
import numpy as np
n_wg = min(n // 256, torch.cuda.get_device_properties(0).multi_processor_count)
w = logp[n + 1024:n + 1024 + 4 * n_wg].view(-1, 4).cpu().double().numpy()
w = (w - w[:, 0].min()) * 0.01  # microseconds
order = np.argsort(w[:, 0])
print("workgroups by entry time (us): entry / first tile staged / last heads done / end")
for i in list(order[:3]) + list(order[-3:]):
    print(f"  wg {i:4d}: " + " ".join(f"{v:7.2f}" for v in w[i]))
print(f"{n_wg} workgroups: entry {w[:, 0].min():.2f} .. {w[:, 0].max():.2f}, end {w[:, 3].min():.2f} .. {w[:, 3].max():.2f};"
      f" entry -> staged median {np.median(w[:, 1] - w[:, 0]):.2f}, sampling epilogue median {np.median(w[:, 3] - w[:, 2]):.2f}")
