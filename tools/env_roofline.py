#!/usr/bin/env python3
"""k_env_step at N envs, eager launches back to back, timed with HIP events on the launch stream: the driver of the round-3
kernel-trace / PMC passes (profiles/r3_env_step_*).   python3 tools/env_roofline.py N [steps] [obs: int8|float32]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "ac-solver_amd"), ROOT]
import numpy as np
import torch

from ac_solver import _acx
from ac_solver.envs.vec_env import ACVecEnv
from bench import ms_pool_at_L

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
K = int(sys.argv[2]) if len(sys.argv) > 2 else 40
obs_dtype = sys.argv[3] if len(sys.argv) > 3 else "int8"
L, W = 25, 10
pool = ms_pool_at_L(L)
env = ACVecEnv(pool[np.arange(N) % len(pool)], horizon_length=1000, obs_dtype=obs_dtype, record_actions=False, final_info=False)
dev = env.device
S = 4  # rollout rows the outputs rotate over (PPO buffers are far larger: every row is written once)
tape = torch.as_tensor(np.random.default_rng(0).integers(0, 12, size=(W + K, N), dtype=np.uint8), device=dev)
obs = torch.empty((S, N, 2 * L), dtype=env.obs_torch_dtype, device=dev)
rew = torch.empty((S, N), dtype=torch.float32, device=dev)
done = torch.empty((S, N), dtype=torch.bool, device=dev)
trunc = torch.empty((S, N), dtype=torch.bool, device=dev)


def launch(k):
    s = k % S
    _acx.check(_acx.lib.acx_env_step(env._h.ptr, tape[k].data_ptr(), _acx.U8, obs[s].data_ptr(), env._obs_code, rew[s].data_ptr(), 0.0, 0.0, done[s].data_ptr(),
                                     trunc[s].data_ptr(), None, 1, env._stream()))


env.reset()
for k in range(W):
    launch(k)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for k in range(W, W + K):
    launch(k)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / K
algo = (4 * L + 7 if obs_dtype == "int8" else 12 * L + 10) * N
print(json.dumps({"envs": N, "steps": K, "obs": obs_dtype, "hip_event_us_per_launch": us, "algorithmic_bytes_per_launch": algo,
                  "algorithmic_GBps": algo / us / 1e3, "frac_of_8TBps": algo / us / 1e3 / 8000.0}))
