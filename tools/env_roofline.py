#!/usr/bin/env python3
"""k_env_step at N envs, eager launches back to back, timed with HIP events on the launch stream: the driver of the round-3
kernel-trace / PMC passes (profiles/r3_env_step_*).   python3 tools/env_roofline.py N [steps per block] [obs: int8|float32] [rollout rows] [blocks]"""
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
S = int(sys.argv[4]) if len(sys.argv) > 4 else 8   # rollout rows the outputs rotate over: as bench.py's throughput_regime (round 3's passes used 4)
BLOCKS = int(sys.argv[5]) if len(sys.argv) > 5 else 5
L = 25
pool = ms_pool_at_L(L)
env = ACVecEnv(pool[np.arange(N) % len(pool)], horizon_length=1000, obs_dtype=obs_dtype, record_actions=False, final_info=False)
dev = env.device
T = 64
tape = torch.as_tensor(np.random.default_rng(0).integers(0, 12, size=(T, N), dtype=np.uint8), device=dev)
obs = torch.empty((S, N, 2 * L), dtype=env.obs_torch_dtype, device=dev)
rew = torch.empty((S, N), dtype=torch.float32, device=dev)
done = torch.empty((S, N), dtype=torch.bool, device=dev)
trunc = torch.empty((S, N), dtype=torch.bool, device=dev)


def launch(k):
    s = k % S
    _acx.check(_acx.lib.acx_env_step(env._h.ptr, tape[k % T].data_ptr(), _acx.U8, obs[s].data_ptr(), env._obs_code, rew[s].data_ptr(), 0.0, 0.0, done[s].data_ptr(),
                                     trunc[s].data_ptr(), None, 1, env._stream()))


env.reset()
# Warm-up until the device has been busy for >= 0.3 s: round 3's passes warmed up for 10 launches (0.8 ms at 4 Mi envs) and timed the
# next 40 while the clocks were still ramping -- 76-79 us per launch where the same kernel inside bench.py (a process that has been
# running for seconds) takes 67-68 us.  `warm_launches` is reported; the kernel trace's average includes them, its MINIMUM and the
# steady blocks below do not.
import time

t0 = time.perf_counter()
k = warm = 0
while time.perf_counter() - t0 < 0.3:
    for _ in range(20):
        launch(k)
        k += 1
    warm += 20
    torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
blocks = []
for _ in range(BLOCKS):
    e0.record()
    for _ in range(K):
        launch(k)
        k += 1
    e1.record()
    torch.cuda.synchronize()
    blocks.append(e0.elapsed_time(e1) * 1e3 / K)
us = sorted(blocks)[len(blocks) // 2]
algo = (4 * L + 7 if obs_dtype == "int8" else 12 * L + 10) * N
print(json.dumps({"envs": N, "steps": K, "blocks_us": blocks, "warm_launches": warm, "rows": S, "obs": obs_dtype, "hip_event_us_per_launch": us,
                  "algorithmic_bytes_per_launch": algo, "algorithmic_GBps": algo / us / 1e3, "frac_of_8TBps": algo / us / 1e3 / 8000.0}))
