#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer env step (acx_env_step_host: actions up, observations / rewards / flags down)."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "ac-solver_amd"), ROOT]
import numpy as np
from ac_solver import _acx
from ac_solver.envs.vec_env import ACVecEnv
from bench import ms_pool_at_L

N, L, T = 65536, 25, 50
pool = ms_pool_at_L(L)
env = ACVecEnv(pool[np.arange(N) % len(pool)], horizon_length=1000, record_actions=False, final_info=False)
act = np.random.default_rng(0).integers(0, 12, size=(T, N)).astype(np.int64)
obs = np.empty((N, 2 * L), np.int8); rew = np.empty(N, np.float32); done = np.empty(N, np.uint8); trunc = np.empty(N, np.uint8)
p = _acx.ptr
def step(t):
    _acx.check(_acx.lib.acx_env_step_host(env._h.ptr, p(act[t], C.c_int64), p(obs, C.c_int8), p(rew, C.c_float), p(done, C.c_uint8), p(trunc, C.c_uint8), None, 1, None, None))
step(0)
t0 = time.perf_counter()
for t in range(1, T):
    step(t)
dt = (time.perf_counter() - t0) / (T - 1)
print(f"acx_env_step_host, {N} envs: {dt * 1e6:.0f} us per step -> {N / dt:.3e} env-steps/s (actions int64 up, obs int8 + reward f32 + 2 flags down, pageable host buffers)")
