#!/usr/bin/env python3
"""How fast can one PPO rollout step (policy sample + env step) go at 131 072 envs?  eager / hipGraph x fp32 / bf16 / fp16."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "ac-solver_amd"), ROOT]
import numpy as np
import torch
from types import SimpleNamespace
from ac_solver.agents.ppo_agent import Agent
from ac_solver.envs.vec_env import ACVecEnv
import bench

L, n, T = 25, 1 << 17, 32
pool = bench.ms_pool_at_L(L)
dev = torch.device("cuda")
env = ACVecEnv(pool[np.arange(n) % len(pool)], horizon_length=1000, obs_dtype="float32", clip_rewards=(-10, 1000), record_actions=False, final_info=False)
agent = Agent(SimpleNamespace(single_observation_space=SimpleNamespace(shape=(2 * L,)), single_action_space=SimpleNamespace(n=12)), [256, 256]).to(dev)
obs = torch.zeros((T + 1, n, 2 * L), device=dev)
rew = torch.zeros((T, n), device=dev)
term = torch.zeros((T + 1, n), dtype=torch.bool, device=dev)
trunc = torch.zeros(n, dtype=torch.bool, device=dev)
act = torch.zeros((T, n), dtype=torch.int64, device=dev)
logp = torch.zeros((T, n), device=dev)
val = torch.zeros((T, n), device=dev)
obs[0].copy_(env.reset()[0])

def rollout(dtype):
    for t in range(T):
        with torch.no_grad(), torch.autocast("cuda", dtype=dtype, enabled=dtype is not None):
            a, lp, _, v = agent.sample_action_and_value(obs[t])
        act[t], logp[t], val[t] = a, lp.float(), v.flatten().float()
        env.step(a, out=(obs[t + 1], rew[t], term[t + 1], trunc), check_errors=False)

e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for name, dtype in (("fp32", None), ("bf16", torch.bfloat16), ("fp16", torch.float16)):
    rollout(dtype); torch.cuda.synchronize()
    e0.record(); rollout(dtype); e1.record(); torch.cuda.synchronize()
    print(f"eager {name}: {n * T / e0.elapsed_time(e1) * 1e3:.3e} env-steps/s ({e0.elapsed_time(e1) / T * 1e3:.1f} us/step)")
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        rollout(dtype)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            rollout(dtype)
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    print(f"graph {name}: {n * T / e0.elapsed_time(e1) * 1e3:.3e} env-steps/s ({e0.elapsed_time(e1) / T * 1e3:.1f} us/step)")

from ac_solver.agents.fused_policy import FusedPolicy
fp = FusedPolicy(agent, 2 * L)
def rollout_fused():
    for t in range(T):
        fp.sample(obs[t], act[t], logp[t], val[t])
        env.step(act[t], out=(obs[t + 1], rew[t], term[t + 1], trunc), check_errors=False)
rollout_fused(); torch.cuda.synchronize()
e0.record(); rollout_fused(); e1.record(); torch.cuda.synchronize()
print(f"eager fused bf16 MFMA policy: {n * T / e0.elapsed_time(e1) * 1e3:.3e} env-steps/s ({e0.elapsed_time(e1) / T * 1e3:.1f} us/step)")
e0.record()
for t in range(T):
    fp.sample(obs[t], act[t], logp[t], val[t])
e1.record(); torch.cuda.synchronize()
print(f"policy kernel alone: {e0.elapsed_time(e1) / T * 1e3:.1f} us per {n} environments")
