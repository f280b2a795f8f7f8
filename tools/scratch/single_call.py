"""Single-call surface timing: ACEnv.step on ONE environment and ACMove on one presentation (bench.py's env_context.single_call_surface)."""
import sys, time
import numpy as np
sys.path.insert(0, "ac-solver_amd")
from ac_solver.envs.ac_env import ACEnv, ACEnvConfig
from ac_solver.envs.ac_moves import ACMove

L = 18
rng = np.random.default_rng(0)
st0 = np.zeros(2 * L, dtype=np.int8); st0[:3] = [1, 1, -2]; st0[L:L + 3] = [2, 1, 2]
for rep in range(3):
    e1 = ACEnv(ACEnvConfig(initial_state=st0, horizon_length=2000))
    acts = rng.integers(0, 12, size=1200)
    for a in acts[:200]: e1.step(int(a))
    t0 = time.perf_counter()
    for a in acts[200:]: e1.step(int(a))
    step_us = (time.perf_counter() - t0) / 1000 * 1e6
    st, ln = st0.copy(), None
    for a in acts[:200]: st, ln = ACMove(int(a), st, L, ln)
    t0 = time.perf_counter()
    for a in acts[200:]: st, ln = ACMove(int(a), st, L, ln)
    move_us = (time.perf_counter() - t0) / 1000 * 1e6
    print(f"step {step_us:.1f} us  ACMove {move_us:.1f} us", flush=True)
