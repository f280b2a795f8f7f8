"""BASELINE config 5 at its stated shape: 2^20 parallel environments x horizon 1000, float32 observations and clipped
rewards written into PyTorch-ROCm tensors, autoreset on (gymnasium vector semantics), more steps than the horizon so that
every environment ends an episode at least once.

Checked two ways, every step:
  * a strided sample of 65 536 environments against the CPU oracle (ACEnv.step of ac_solver/envs/ac_env.py:95-113,
    the clip of agents/environment.py:48-52, autoreset on the host side of the mirror);
  * the WHOLE batch against the int8 / unclipped kernel of BASELINE config 2 run on the same states and actions:
    per-environment equality of observations, rewards (after the clip) and the two flags.
"""
import numpy as np
import pytest

from tests.conftest import ms_pool_generator_order, ms_pool_rows

pytestmark = pytest.mark.gpu

N, L, HORIZON, STEPS, STRIDE = 1 << 20, 25, 1000, 1101, 16
CLIP = (-10.0, 1000.0)  # agents/args.py:231-242 (min_rew, max_rew)


@pytest.mark.timeout(1500)
def test_config5_full_shape_against_oracle_and_config2_kernel(golden_json):
    import torch

    from ac_solver import _acx
    from ac_solver.envs.vec_env import ACVecEnv
    from oracle import ac_oracle as O

    _acx.require_device()
    rows = ms_pool_rows(ms_pool_generator_order(golden_json("ms_pool.json")), L)
    states = rows[np.arange(N) % len(rows)]
    f32 = ACVecEnv(states, horizon_length=HORIZON, obs_dtype="float32", clip_rewards=CLIP, record_actions=False, final_info=False)
    i8 = ACVecEnv(states, horizon_length=HORIZON, obs_dtype="int8", clip_rewards=None, record_actions=False, final_info=False)
    assert f32.max_reward == HORIZON * L * 2 == 50000
    o32, _ = f32.reset()
    o8, _ = i8.reset()
    assert o32.dtype == torch.float32 and o32.shape == (N, 2 * L) and torch.equal(o32, o8.float())

    sample = torch.arange(0, N, STRIDE, device="cuda")
    want = states[::STRIDE].copy()
    init = want.copy()
    counts = np.zeros(len(want), np.int32)
    err_any = np.zeros(len(want), bool)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(5)
    bad = torch.zeros(4, dtype=torch.int64, device="cuda")  # whole-batch mismatches: obs, reward, terminated, truncated
    n_term = n_trunc = 0
    ever_finished = torch.zeros(N, dtype=torch.bool, device="cuda")
    for t in range(STEPS):
        act = torch.randint(0, 12, (N,), dtype=torch.uint8, device="cuda", generator=gen)
        obs, rew, term, trunc, _ = f32.step(act, check_errors=False)
        obs8, rew8, term8, trunc8, _ = i8.step(act, check_errors=False)
        # ---- whole batch: the f32 / clipped epilogue against the config-2 kernel -----------------------------------------
        bad[0] += (obs != obs8.float()).any(dim=1).sum()
        bad[1] += (rew != rew8.clamp(CLIP[0], CLIP[1])).sum()
        bad[2] += (term != term8).sum()
        bad[3] += (trunc != trunc8).sum()
        ever_finished |= term | trunc
        # ---- strided sample against the oracle -------------------------------------------------------------------------
        a_h = act[sample].cpu().numpy()
        r, d, tr, err = O.env_rollout(want, counts, HORIZON, a_h[None])
        err_any |= err != 0  # a move on which the reference's ACMove raises: state and counter untouched on both sides
        fin = (d[0] | tr[0]).astype(bool)
        want[fin], counts[fin] = init[fin], 0  # gymnasium autoreset: back to the environment's own initial state
        assert np.array_equal(rew[sample].cpu().numpy(), np.clip(r[0].astype(np.float32), *CLIP)), t
        assert np.array_equal(term[sample].cpu().numpy().astype(np.uint8), d[0]), t
        assert np.array_equal(trunc[sample].cpu().numpy().astype(np.uint8), tr[0]), t
        got = obs[sample].cpu().numpy()
        assert got.dtype == np.float32 and np.array_equal(got, want.astype(np.float32)), t
        n_term += int(d[0].sum())
        n_trunc += int(tr[0].sum())
    assert bad.cpu().tolist() == [0, 0, 0, 0]
    assert bool(ever_finished.all())          # STEPS > HORIZON: every environment ended an episode (most by truncation)
    assert n_trunc >= len(want) - n_term and n_term > 0
    assert np.array_equal(f32.get_counts(np.arange(0, N, STRIDE)), counts)
    errs = np.empty(N, np.uint8)
    import ctypes as C

    _acx.check(_acx.lib.acx_env_get_errors(f32._h.ptr, _acx.ptr(errs, C.c_uint8), 0, f32._stream()))
    assert np.array_equal(errs[::STRIDE] != 0, err_any)
