#!/usr/bin/env python3
"""Per-kernel time of K back-to-back launches replayed from one hipGraph (device time by HIP events)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "ac-solver_amd"), ROOT]
import numpy as np
import torch

from ac_solver import _acx
from ac_solver.envs.vec_env import ACVecEnv
from bench import ms_pool_at_L

L, K = int(os.environ.get("ACX_MB_L", "25")), 500


def graph_time(body):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for k in range(K):
            body(k)
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / K)
    return best


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    pool = ms_pool_at_L(L)
    states = pool[np.arange(N) % len(pool)]
    tape = torch.as_tensor(np.random.default_rng(0).integers(0, 12, size=(K, N), dtype=np.uint8), device="cuda")
    env = ACVecEnv(states, horizon_length=1000, record_actions=False, final_info=False)
    obs1 = torch.empty((N, 2 * L), dtype=torch.int8, device="cuda")
    obsK = torch.empty((K, N, 2 * L), dtype=torch.int8, device="cuda")
    rew = torch.empty((K, N), dtype=torch.float32, device="cuda")
    done = torch.empty((K, N), dtype=torch.bool, device="cuda")
    trunc = torch.empty((K, N), dtype=torch.bool, device="cuda")
    x = torch.zeros(N, device="cuda")
    tiny = torch.zeros(64, device="cuda")
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        st = lambda: torch.cuda.current_stream().cuda_stream  # noqa: E731
        r = {}
        r["torch tiny add_ (64 elts)"] = graph_time(lambda k: tiny.add_(1.0))
        r["torch add_ (N elts)"] = graph_time(lambda k: x.add_(1.0))
        src_obs = torch.zeros((N, 2 * L), dtype=torch.int8, device="cuda")
        src_state = torch.zeros((N, 3), dtype=torch.int64, device="cuda")
        dst_state = torch.zeros((N, 3), dtype=torch.int64, device="cuda")
        r["torch copy_ 50 B/env (obs-sized stream)"] = graph_time(lambda k: obsK[k].copy_(src_obs))
        r["torch copy_ 24 B/env (state-sized stream)"] = graph_time(lambda k: dst_state.copy_(src_state))
        r["observe -> one buffer"] = graph_time(lambda k: _acx.lib.acx_env_observe(env._h.ptr, obs1.data_ptr(), _acx.I8, st()))
        r["step, no outputs but state"] = graph_time(lambda k: _acx.lib.acx_env_step(env._h.ptr, tape[k].data_ptr(), _acx.U8, None, _acx.I8, None, 0.0, 0.0, None, None, None, 1, st()))
        r["step, rew/done/trunc"] = graph_time(lambda k: _acx.lib.acx_env_step(env._h.ptr, tape[k].data_ptr(), _acx.U8, None, _acx.I8, rew[k].data_ptr(), 0.0, 0.0, done[k].data_ptr(), trunc[k].data_ptr(), None, 1, st()))
        r["step + obs -> one buffer"] = graph_time(lambda k: _acx.lib.acx_env_step(env._h.ptr, tape[k].data_ptr(), _acx.U8, obs1.data_ptr(), _acx.I8, rew[k].data_ptr(), 0.0, 0.0, done[k].data_ptr(), trunc[k].data_ptr(), None, 1, st()))
        r["step + obs -> [K,N,50]"] = graph_time(lambda k: _acx.lib.acx_env_step(env._h.ptr, tape[k].data_ptr(), _acx.U8, obsK[k].data_ptr(), _acx.I8, rew[k].data_ptr(), 0.0, 0.0, done[k].data_ptr(), trunc[k].data_ptr(), None, 1, st()))
    for k, v in r.items():
        print(f"N={N} {k:32s} {v:7.2f} us/kernel")


if __name__ == "__main__":
    main()
