"""acx_policy_sample (csrc/acx_policy.hip): the PPO agent's actor + critic on the matrix cores, fused with the action draw.
Numerics against a plain PyTorch f32 forward of the same nn.Modules (bf16 tolerance), tightly against a torch emulation that
rounds weights and activations to bf16 where the kernel does, and the sampled actions against the policy's distribution."""
from types import SimpleNamespace

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _agent(in_dim, n_act, seed):
    import torch

    from ac_solver.agents.ppo_agent import Agent

    torch.manual_seed(seed)
    agent = Agent(SimpleNamespace(single_observation_space=SimpleNamespace(shape=(in_dim,)), single_action_space=SimpleNamespace(n=n_act)), [256, 256]).cuda()
    with torch.no_grad():  # a policy head that is far from uniform, biases that are not zero
        for seq in (agent.actor, agent.critic):
            for m in seq:
                if hasattr(m, "bias"):
                    m.bias.uniform_(-0.5, 0.5)
        agent.actor[-1].weight.mul_(40.0)
    return agent


def _emulate(seq, x):
    """the kernel's arithmetic in torch: bf16 operands, f32 accumulation, bf16 activations; the hidden layers' weights and
    biases carry the factor 2 / ln 2 of tanh(x) = 1 - 2 / (2^(2x / ln 2) + 1) before they are rounded (fused_policy.TANH_SCALE)"""
    import torch

    from ac_solver.agents.fused_policy import TANH_SCALE

    bf = lambda t: t.to(torch.bfloat16).to(torch.float32)  # noqa: E731
    lin = [m for m in seq if isinstance(m, torch.nn.Linear)]
    h = bf(x)
    for k, m in enumerate(lin):
        if k < 2:
            y = h @ bf(m.weight * TANH_SCALE).T + m.bias * TANH_SCALE
            h = bf(1.0 - 2.0 / (torch.exp2(y) + 1.0))
        else:
            h = h @ bf(m.weight).T + m.bias
    return h


# (more than 65 536 environments: a workgroup walks several tiles of 256, the last one ragged; odd widths read the staged rows 16 bits at a time)
@pytest.mark.parametrize("in_dim,n_act,n", [(50, 12, 4099), (72, 12, 1000), (14, 12, 33), (80, 16, 257), (3, 2, 64), (72, 12, 70001), (51, 12, 131333)])
def test_fused_policy_matches_torch(in_dim, n_act, n):
    import torch

    from ac_solver.agents.fused_policy import FusedPolicy

    agent = _agent(in_dim, n_act, 7 + in_dim)
    fp = FusedPolicy(agent, in_dim, seed=1)
    obs = torch.randint(-2, 3, (n, in_dim), device="cuda").float()
    action = torch.full((n,), -1, dtype=torch.int64, device="cuda")
    logp = torch.zeros(n, device="cuda")
    val = torch.zeros(n, device="cuda")
    fp.sample(obs, action, logp, val)
    torch.cuda.synchronize()
    assert int(action.min()) >= 0 and int(action.max()) < n_act
    with torch.no_grad():
        want_lp = torch.log_softmax(agent.actor(obs), -1).gather(-1, action[:, None])[:, 0]
        want_v = agent.critic(obs)[:, 0]
        emu_lp = torch.log_softmax(_emulate(agent.actor, obs), -1).gather(-1, action[:, None])[:, 0]
        emu_v = _emulate(agent.critic, obs)[:, 0]
    # tolerance written out: bf16 operands (8 significant bits) through two 256-wide tanh layers
    assert float((logp - want_lp).abs().max()) < 0.15 and float((val - want_v).abs().max()) < 0.08 * (1 + float(want_v.abs().max()))
    assert float((logp - emu_lp).abs().max()) < 5e-3 and float((val - emu_v).abs().max()) < 5e-3


def test_fused_policy_samples_the_policys_distribution():
    import torch

    from ac_solver.agents.fused_policy import FusedPolicy

    agent = _agent(50, 12, 3)
    fp = FusedPolicy(agent, 50, seed=5)
    row = torch.randint(-2, 3, (1, 50), device="cuda").float()
    n = 1 << 18
    obs = row.repeat(n, 1).contiguous()
    action = torch.zeros(n, dtype=torch.int64, device="cuda")
    logp, val = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    counts = torch.zeros(12, dtype=torch.float64, device="cuda")
    for _ in range(4):  # a fresh seed per call
        fp.sample(obs, action, logp, val)
        counts += torch.bincount(action, minlength=12).double()
    with torch.no_grad():
        p = torch.softmax(_emulate(agent.actor, row)[0].double(), -1)
    freq = counts / counts.sum()
    sigma = torch.sqrt(p * (1 - p) / counts.sum())
    assert float(((freq - p).abs() / (sigma + 1e-9)).max()) < 6.0, (freq.tolist(), p.tolist())
    assert float(p.max()) < 0.9  # the test policy is not degenerate
    # refresh() picks up new weights
    with torch.no_grad():
        agent.actor[-1].bias[3] += 50.0
    fp.refresh()
    fp.sample(obs, action, logp, val)
    assert float((action == 3).double().mean()) > 0.999


def test_int8_observations_give_the_same_outputs():
    """acx_env_step writes int8 or float32 observation rows; the policy kernel reads either (the letters -2..2 are exact in both):
    same seed -> same actions, log-probabilities and values"""
    import torch

    from ac_solver.agents.fused_policy import FusedPolicy

    n, in_dim = 3001, 50
    agent = _agent(in_dim, 12, 21)
    obs = torch.randint(-2, 3, (n, in_dim), device="cuda")
    outs = []
    for o in (obs.float(), obs.to(torch.int8)):
        fp = FusedPolicy(agent, in_dim, seed=9)
        a = torch.zeros(n, dtype=torch.int64, device="cuda")
        lp, v = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
        fp.sample(o.contiguous(), a, lp, v)
        outs.append((a.clone(), lp.clone(), v.clone()))
    for x, y in zip(*outs):
        assert torch.equal(x, y)


@pytest.mark.parametrize("dtype", ["float32", "int8"])
def test_observation_rows_at_any_address(dtype):
    """the kernel loads a tile's observations four elements at a time when the matrix starts on a 16-byte (int8: 4-byte) boundary and
    element by element otherwise: the same outputs from a copy that starts one element off"""
    import torch

    from ac_solver.agents.fused_policy import FusedPolicy

    n, in_dim = 66001, 50
    agent = _agent(in_dim, 12, 33)
    dt = getattr(torch, dtype)
    obs = torch.randint(-2, 3, (n, in_dim), device="cuda").to(dt)
    shifted = torch.zeros(n * in_dim + 1, dtype=dt, device="cuda")[1:].view(n, in_dim)
    shifted.copy_(obs)
    assert shifted.data_ptr() % 4 != 0 if dtype == "int8" else shifted.data_ptr() % 16 != 0
    outs = []
    for o in (obs, shifted):
        fp = FusedPolicy(agent, in_dim, seed=4)
        a = torch.zeros(n, dtype=torch.int64, device="cuda")
        lp, v = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
        fp.sample(o, a, lp, v)
        outs.append((a.clone(), lp.clone(), v.clone()))
    for x, y in zip(*outs):
        assert torch.equal(x, y)
