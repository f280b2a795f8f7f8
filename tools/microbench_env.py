#!/usr/bin/env python3
"""Device-time microbenchmarks of the env kernels (HIP events, graph-free back-to-back launches).
Usage: python tools/microbench_env.py [N ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "ac-solver_amd"), ROOT]
import numpy as np
import torch

from ac_solver import _acx
from ac_solver.envs.vec_env import ACVecEnv
from bench import ms_pool_at_L

L = int(os.environ.get("ACX_MB_L", "25"))


def timeit(fn, reps):
    torch.cuda.synchronize()
    fn(0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(reps):
        fn(k)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps  # us


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [65536, 1 << 20]
    pool = ms_pool_at_L(L)
    for N in sizes:
        states = pool[np.arange(N) % len(pool)]
        T = 200
        tape = torch.as_tensor(np.random.default_rng(0).integers(0, 12, size=(T, N), dtype=np.uint8), device="cuda")
        for obs_dtype, code in (("int8", _acx.I8), ("float32", _acx.F32)):
            env = ACVecEnv(states, horizon_length=1000, obs_dtype=obs_dtype, record_actions=False, final_info=False)
            obs = torch.empty((N, 2 * L), dtype=env.obs_torch_dtype, device="cuda")
            rew = torch.empty(N, dtype=torch.float32, device="cuda")
            done = torch.empty(N, dtype=torch.bool, device="cuda")
            trunc = torch.empty(N, dtype=torch.bool, device="cuda")
            st = env._stream()

            def step_obs(k):
                _acx.lib.acx_env_step(env._h.ptr, tape[k % T].data_ptr(), _acx.U8, obs.data_ptr(), code, rew.data_ptr(), 0.0, 0.0, done.data_ptr(),
                                      trunc.data_ptr(), None, 1, st)

            def step_noobs(k):
                _acx.lib.acx_env_step(env._h.ptr, tape[k % T].data_ptr(), _acx.U8, None, code, rew.data_ptr(), 0.0, 0.0, done.data_ptr(),
                                      trunc.data_ptr(), None, 1, st)

            def observe(k):
                _acx.lib.acx_env_observe(env._h.ptr, obs.data_ptr(), code, st)

            res = {"step+obs": timeit(step_obs, 200), "step": timeit(step_noobs, 200), "observe": timeit(observe, 200)}
            if obs_dtype == "int8":
                rw = torch.empty((T, N), dtype=torch.float32, device="cuda")
                dn = torch.empty((T, N), dtype=torch.bool, device="cuda")

                def rollout(k):
                    _acx.lib.acx_env_rollout(env._h.ptr, tape.data_ptr(), T, rw.data_ptr(), 0.0, 0.0, dn.data_ptr(), None, 1, st)

                res["rollout/step"] = timeit(rollout, 5) / T
            print(f"N={N} obs={obs_dtype}: " + "  ".join(f"{k}={v:.2f}us" for k, v in res.items()), flush=True)
            bytes_step = (4 * L + 7) * N
            print(f"    step+obs: {N / res['step+obs'] * 1e6:.3e} env-steps/s, {bytes_step / res['step+obs'] / 1e3:.1f} GB/s algorithmic")


if __name__ == "__main__":
    main()
