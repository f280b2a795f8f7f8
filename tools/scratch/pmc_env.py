#!/usr/bin/env python3
"""Small driver for rocprofv3 --pmc runs: N envs, a few step launches (int8 obs)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "ac-solver_amd"), ROOT]
import numpy as np
import torch

from ac_solver import _acx
from ac_solver.envs.vec_env import ACVecEnv
from bench import ms_pool_at_L

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
L = 25
pool = ms_pool_at_L(L)
env = ACVecEnv(pool[np.arange(N) % len(pool)], horizon_length=1000, record_actions=False, final_info=False)
tape = torch.as_tensor(np.random.default_rng(0).integers(0, 12, size=(reps, N), dtype=np.uint8), device="cuda")
obs = torch.empty((reps, N, 2 * L), dtype=torch.int8, device="cuda")
rew = torch.empty((reps, N), dtype=torch.float32, device="cuda")
done = torch.empty((reps, N), dtype=torch.bool, device="cuda")
trunc = torch.empty((reps, N), dtype=torch.bool, device="cuda")
for k in range(reps):
    _acx.lib.acx_env_step(env._h.ptr, tape[k].data_ptr(), _acx.U8, obs[k].data_ptr(), _acx.I8, rew[k].data_ptr(), 0.0, 0.0, done[k].data_ptr(),
                          trunc[k].data_ptr(), None, 1, env._stream())
torch.cuda.synchronize()
