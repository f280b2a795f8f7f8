#!/usr/bin/env python3
"""Where the period of a replayed k_env_step launch goes at 65 536 envs: in-kernel wave stamps (100 MHz constant clock) from a
DIAGNOSTIC build of the library:

    bash tools/build_variant.sh stamp -DACX_STEP_STAMP
    ACX_LIB=ac-solver_amd/lib/var_stamp.so python3 tools/step_stamps.py [envs] [steps per graph] [replays]

The same graph replay as bench.py (K steps captured once, replayed back to back).  Every wave logs (first instruction, last
store retired).  Launches do not overlap (each depends on its predecessor), so sorting the waves by begin and cutting where a
wave begins after everything before it has ended recovers the launches: ACTIVE = last end - first begin of a launch, PERIOD =
first begin to the next launch's first begin, GAP = PERIOD - ACTIVE (launch overhead between dependent kernels).  The stamps
cost the kernel a wait for its own stores and one atomic per wave: read the shares, not the absolute period."""
import ctypes as C
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

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
K = int(sys.argv[2]) if len(sys.argv) > 2 else 100
R = int(sys.argv[3]) if len(sys.argv) > 3 else 10
L = 25
assert hasattr(_acx.lib, "acx_debug_stamps"), "needs the -DACX_STEP_STAMP build (ACX_LIB=.../var_stamp.so)"
pool = ms_pool_at_L(L)
env = ACVecEnv(pool[np.arange(N) % len(pool)], horizon_length=1000, record_actions=False, final_info=False)
dev = env.device
tape = torch.as_tensor(np.random.default_rng(0).integers(0, 12, size=(K, N), dtype=np.uint8), device=dev)
obs = torch.empty((K, N, 2 * L), dtype=torch.int8, device=dev)
rew = torch.empty((K, N), dtype=torch.float32, device=dev)
done = torch.empty((K, N), dtype=torch.bool, device=dev)
trunc = torch.empty((K, N), dtype=torch.bool, device=dev)


def launch(k):
    _acx.check(_acx.lib.acx_env_step(env._h.ptr, tape[k].data_ptr(), _acx.U8, obs[k].data_ptr(), _acx.I8, rew[k].data_ptr(), 0.0, 0.0, done[k].data_ptr(),
                                     trunc[k].data_ptr(), None, 1, env._stream()))


env.reset()
for k in range(K):
    launch(k)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for k in range(K):
        launch(k)
g.replay()
torch.cuda.synchronize()
n = C.c_int64()
waves = -(-N // 256) * 4  # four waves per workgroup, 64 envs per wave
_acx.lib.acx_debug_stamps.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.POINTER(C.c_int64), C.c_int]
_acx.check(_acx.lib.acx_debug_stamps(None, 0, waves, C.byref(n), 1))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(R):
    g.replay()
e1.record()
torch.cuda.synchronize()
period_events = e0.elapsed_time(e1) * 1e3 / (K * R)
buf = np.zeros((1 << 21, 2), np.uint64)
_acx.check(_acx.lib.acx_debug_stamps(buf.ctypes.data, len(buf), waves, C.byref(n), 1))
nl = int(n.value)
st = buf[: nl * waves].astype(np.int64).reshape(nl, waves, 2)  # [launch][wave](begin, end), 10 ns ticks
first, last = st[:, :, 0].min(axis=1), st[:, :, 1].max(axis=1)
active = (last - first) * 10.0                     # ns: first instruction of the launch's first wave -> last store of its last wave retired
period = np.diff(first) * 10.0
gap = (first[1:] - last[:-1]) * 10.0
inside = np.ones(len(period), bool)
inside[K - 1::K] = False  # the boundary between two graph replays is not a kernel-to-kernel boundary
wave = (st[:, :, 1] - st[:, :, 0]) * 10.0
spread = (st[:, :, 0].max(axis=1) - first) * 10.0  # how long the dispatcher takes to start all waves of a launch
out = {"envs": N, "graph_steps": K, "replays": R, "launches_logged": nl, "waves_per_launch": waves,
       "hip_event_period_us": period_events,
       "stamp_period_us_median": float(np.median(period[inside])) / 1e3, "active_us_median": float(np.median(active)) / 1e3,
       "active_us_p10_p90": [float(np.percentile(active, 10)) / 1e3, float(np.percentile(active, 90)) / 1e3],
       "gap_us_median": float(np.median(gap[inside])) / 1e3, "one_wave_us_median": float(np.median(wave)) / 1e3,
       "wave_start_spread_us_median": float(np.median(spread)) / 1e3,
       "algorithmic_bytes_per_launch": 107 * N,
       "frac_of_8TBps_by_active_time": 107 * N / (float(np.median(active)) * 1e-9) / 8e12,
       "frac_of_8TBps_by_period": 107 * N / (float(np.median(period[inside])) * 1e-9) / 8e12,
       "note": "diagnostic build: every wave waits for its own stores before the end stamp and logs 16 bytes behind it, which lengthens the period; "
               "read ACTIVE and the wave-start spread, take the period from the unstamped bench"}
print(json.dumps(out, indent=1))
