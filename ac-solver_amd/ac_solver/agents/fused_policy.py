"""Rollout-time inference of the PPO agent on the matrix cores: `acx_policy_sample` (csrc/acx_policy.hip) evaluates actor
and critic (reference: ac_solver/agents/ppo_agent.py:11-109, two tanh MLPs of width 256) on every environment and samples the
action in one kernel -- bf16 operands, f32 accumulation.  Opt-in (`--fused-policy` of ac_solver.agents.ppo): the reference's
policy is f32 torch, which stays the default and is what the PPO update always differentiates."""
import numpy as np
import torch

from ac_solver import _acx


TANH_SCALE = 2.0 / float(np.log(2.0))  # tanh(x) = 1 - 2 / (2^(x * TANH_SCALE) + 1): folded into the layers that feed a tanh


def _fragments(w, b, rows, cols, hidden_input, scale=1.0):
    """nn.Linear weight [out, in] / bias zero-padded to [rows, cols] (multiples of 32 / 16) -> the A-operand fragments of
    v_mfma_f32_32x32x16_bf16 in the order the kernel reads them: [out block][bias step, k steps][lane][8], lane = 32 * h + row.
    Element j of lane (row, h) in k-step ks is input 16 ks + 8 h + j for the first layer; for a layer fed by a hidden layer it
    is hidden unit 32 (ks >> 1) + 16 (ks & 1) + 8 (j >> 2) + 4 h + (j & 3): the eight accumulator registers the lane of the
    previous layer already holds (csrc/acx_policy.hip).  The bias step carries b as bf16 hi + lo in j = 0, 1 of h = 0.
    `scale` multiplies weight and bias before the rounding to bf16 (TANH_SCALE for the two hidden layers)."""
    out, inp = w.shape
    dev = w.device
    pad = torch.zeros((rows, cols), dtype=torch.float32, device=dev)
    pad[:out, :inp] = w.detach().float() * scale
    nks = cols // 16
    ks = torch.arange(nks, device=dev).view(nks, 1, 1)
    hh = torch.arange(2, device=dev).view(1, 2, 1)
    j = torch.arange(8, device=dev).view(1, 1, 8)
    idx = (32 * (ks >> 1) + 16 * (ks & 1) + 8 * (j >> 2) + 4 * hh + (j & 3)) if hidden_input else (16 * ks + 8 * hh + j)
    t = pad[:, idx.reshape(-1)].view(rows // 32, 32, nks, 2, 8).permute(0, 2, 3, 1, 4)  # [ob][ks][h][row][j]
    bias = torch.zeros(rows, dtype=torch.float32, device=dev)
    bias[:out] = b.detach().float() * scale
    hi = bias.to(torch.bfloat16).float()
    bf = torch.zeros((rows // 32, 1, 2, 32, 8), dtype=torch.float32, device=dev)
    bf[:, 0, 0, :, 0] = hi.view(-1, 32)
    bf[:, 0, 0, :, 1] = (bias - hi).view(-1, 32)
    return torch.cat([bf, t], dim=1).contiguous().view(-1).to(torch.bfloat16)


def supported(agent, in_dim):
    def shape_ok(seq, out):
        lin = [m for m in seq if isinstance(m, torch.nn.Linear)]
        return (len(lin) == 3 and lin[0].in_features == in_dim and lin[0].out_features == 256 and lin[1].in_features == 256
                and lin[1].out_features == 256 and lin[2].in_features == 256 and lin[2].out_features == out)

    n_act = [m for m in agent.actor if isinstance(m, torch.nn.Linear)][-1].out_features
    return in_dim <= 80 and n_act <= 16 and shape_ok(agent.actor, n_act) and shape_ok(agent.critic, 1)


def pack_network(seq, in_dim):
    """[Linear, Tanh, Linear, Tanh, Linear] -> one uint8 device tensor: the fragments of the three layers, bias fragments included"""
    lin = [m for m in seq if isinstance(m, torch.nn.Linear)]
    ks1 = (in_dim + 15) // 16
    parts = [_fragments(lin[0].weight, lin[0].bias, 256, 16 * ks1, False, TANH_SCALE), _fragments(lin[1].weight, lin[1].bias, 256, 256, True, TANH_SCALE),
             _fragments(lin[2].weight, lin[2].bias, 32, 256, True)]
    out = torch.cat([p.view(torch.uint8) for p in parts]).contiguous()
    assert out.numel() == _acx.lib.acx_policy_packed_bytes(in_dim)
    return out


class FusedPolicy:
    """`sample(obs, action, logprob, value)` = agent.get_action_and_value(obs) for a rollout step, written into the caller's
    tensors; `refresh()` re-packs the weights (call it after every optimizer step)."""

    def __init__(self, agent, in_dim, seed=0):
        if not supported(agent, in_dim):
            raise ValueError("the fused policy kernel handles in -> 256 -> 256 -> out tanh networks with in <= 80 and out <= 16")
        _acx.require_device()
        self.agent, self.in_dim = agent, int(in_dim)
        self.n_actions = [m for m in agent.actor if isinstance(m, torch.nn.Linear)][-1].out_features
        self._seed = np.random.default_rng(seed)
        self.refresh()

    def refresh(self):
        with torch.no_grad():
            self.actor = pack_network(self.agent.actor, self.in_dim)
            self.critic = pack_network(self.agent.critic, self.in_dim)

    def sample(self, obs, action, logprob, value):
        assert obs.dtype in (torch.float32, torch.int8) and obs.is_contiguous() and obs.shape[-1] == self.in_dim
        assert action.dtype == torch.int64 and logprob.dtype == torch.float32 and value.dtype == torch.float32
        n = obs.numel() // self.in_dim
        seed = int(self._seed.integers(0, 1 << 63))
        _acx.check(_acx.lib.acx_policy_sample(obs.data_ptr(), _acx.I8 if obs.dtype == torch.int8 else _acx.F32, n, self.in_dim, self.actor.data_ptr(), self.critic.data_ptr(), self.n_actions, seed,
                                              action.data_ptr(), logprob.data_ptr(), value.data_ptr(), torch.cuda.current_stream(obs.device).cuda_stream),
                   "acx_policy_sample")
