#!/usr/bin/env python3
"""The PPO update of ac_solver/agents/training.py alone, at BASELINE config 5's per-GPU shape (4 minibatches of 1 Mi samples through the
actor and critic MLPs, f32): wall time per update and, with `profile`, torch's per-kernel table.   python tools/update_probe.py [profile]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "ac-solver_amd"), ROOT]
from types import SimpleNamespace

import numpy as np
import torch
from torch import nn

from ac_solver.agents.ppo_agent import Agent

dev = torch.device("cuda")
N, T, L = 131072, 32, 25
agent = Agent(SimpleNamespace(single_observation_space=SimpleNamespace(shape=(2 * L,)), single_action_space=SimpleNamespace(n=12)), [256, 256]).to(dev)
opt = torch.optim.Adam(agent.parameters(), lr=2.5e-4, eps=1e-5)
b_obs = torch.randint(-2, 3, (N * T, 2 * L), dtype=torch.int8, device=dev)
b_actions = torch.randint(0, 12, (N * T,), device=dev)
b_logprobs = torch.full((N * T,), -2.48, device=dev)
b_adv = torch.randn(N * T, device=dev)
b_ret = torch.randn(N * T, device=dev)
b_val = torch.randn(N * T, device=dev)
mbs = N * T // 4


def update():
    inds = np.arange(N * T)
    np.random.shuffle(inds)
    for start in range(0, N * T, mbs):
        mb = torch.as_tensor(inds[start:start + mbs], device=dev)
        _, newlogprob, entropy, newvalue = agent.get_action_and_value(b_obs[mb].float(), b_actions[mb])
        logratio = newlogprob - b_logprobs[mb]
        ratio = logratio.exp()
        mb_adv = b_adv[mb]
        mb_adv = (mb_adv - mb_adv.mean()) / (mb_adv.std() + 1e-8)
        pg_loss = torch.max(-mb_adv * ratio, -mb_adv * torch.clamp(ratio, 0.8, 1.2)).mean()
        newvalue = newvalue.view(-1)
        v_clipped = b_val[mb] + torch.clamp(newvalue - b_val[mb], -0.2, 0.2)
        v_loss = 0.5 * torch.max((newvalue - b_ret[mb]) ** 2, (v_clipped - b_ret[mb]) ** 2).mean()
        loss = pg_loss - 0.01 * entropy.mean() + v_loss * 0.5
        opt.zero_grad()
        loss.backward()
        nn.utils.clip_grad_norm_(agent.parameters(), 0.5)
        opt.step()


for _ in range(2):
    update()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    update()
torch.cuda.synchronize()
print(f"update: {(time.perf_counter() - t0) / 3 * 1e3:.1f} ms")
if len(sys.argv) > 1 and sys.argv[1] == "profile":
    from torch.profiler import ProfilerActivity, profile

    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        update()
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=70))
