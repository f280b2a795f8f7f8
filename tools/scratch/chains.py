#!/usr/bin/env python3
"""experiment: 65 536 envs stepped K times as C independent chains (C ACVecEnv objects of 65 536 / C envs, C parallel branches of one
hipGraph): does the boundary between dependent launches of one chain hide behind the kernels of the other chains?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "ac-solver_amd"), ROOT]
import numpy as np
import torch
from ac_solver import _acx
from ac_solver.envs.vec_env import ACVecEnv
from bench import ms_pool_at_L

L, N, K, ROWS = 25, 65536, 128, 128
pool = ms_pool_at_L(L)
dev = torch.device("cuda")
tape = torch.as_tensor(np.random.default_rng(0).integers(0, 12, size=(K, N), dtype=np.uint8), device=dev)
obs = torch.empty((ROWS, N, 2 * L), dtype=torch.int8, device=dev)
rew = torch.zeros((ROWS, N), dtype=torch.float32, device=dev)
done = torch.empty((ROWS, N), dtype=torch.bool, device=dev)
trunc = torch.empty((ROWS, N), dtype=torch.bool, device=dev)
for C in (1, 2, 4, 8, 16):
    n = N // C
    envs = [ACVecEnv(pool[(np.arange(n) + c * n) % len(pool)], horizon_length=1000, obs_dtype="int8", record_actions=False, final_info=False) for c in range(C)]
    for e in envs:
        e.reset()
    streams = [torch.cuda.Stream() for _ in range(C)]

    def launch(c, t, slot):
        e, o = envs[c], c * n
        _acx.check(_acx.lib.acx_env_step(e._h.ptr, tape[t, o:o + n].data_ptr(), _acx.U8, obs[slot, o:o + n].data_ptr(), _acx.I8, rew[slot, o:o + n].data_ptr(), 0.0, 0.0,
                                         done[slot, o:o + n].data_ptr(), trunc[slot, o:o + n].data_ptr(), None, 1, e._stream()), "acx_env_step")

    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        main = torch.cuda.current_stream()
        for c in range(C):
            streams[c].wait_stream(main)
            with torch.cuda.stream(streams[c]):
                for k in range(K):
                    launch(c, k, k % ROWS)
        for c in range(C):
            main.wait_stream(streams[c])
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 50
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (reps * K)
    print(f"chains {C:2d} x {n:6d} envs: {us:.3f} us per step of {N} envs -> {N / us * 1e6:.3e} env-steps/s, {107 * N / us / 1e3 / 8000:.3f} of 8 TB/s", flush=True)
    del g, envs
