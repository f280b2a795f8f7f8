"""Actor and critic networks (reference: ac_solver/agents/ppo_agent.py:11-109): two tanh MLPs over the
2L-entry observation, orthogonal initialisation (gain sqrt(2); 0.01 for the policy head, 1.0 for the value head)."""
import math

import numpy as np
import torch
from torch import nn
from torch.distributions import Categorical


class _LinearSplitK(torch.autograd.Function):
    """y = x W^T + b with a SPLIT-K weight gradient.  The update of BASELINE config 5 pushes minibatches of 1 Mi samples through
    256-wide layers: dW = dY^T X contracts over those 1 Mi rows into a 256 x 256 (or 256 x 50, 12 x 256, 1 x 256) output, and the
    library's single GEMM for that shape runs at 4-60 TFLOP/s in small-tile kernels -- 48 of the update's 100 ms of device time
    (torch's kernel table, tools/update_probe.py), five to ten times the time it takes to read the two operands once.  Here the rows are cut
    into chunks, one batched GEMM forms a partial product per chunk and a sum folds them: the same numbers up to f32 summation
    order.  Forward and data gradient are the library's GEMMs (they run at the f32 matrix peak)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        return torch.addmm(bias, x, weight.t())

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gy = gy.contiguous()
        gx = gy @ weight if ctx.needs_input_grad[0] else None
        n = x.shape[0]
        chunks = 1
        while chunks < 512 and n % (2 * chunks) == 0 and n // (2 * chunks) >= 2048:
            chunks *= 2
        gw = torch.bmm(gy.view(chunks, n // chunks, -1).transpose(1, 2), x.reshape(chunks, n // chunks, -1)).sum(0)
        return gx, gw, gy.sum(0)


class Linear(nn.Linear):
    """nn.Linear (same parameters, same state_dict keys) whose large-batch training passes on the GPU take the split-K weight gradient."""

    SPLIT_K_ROWS = 1 << 16

    def forward(self, x):
        if x.is_cuda and x.dim() == 2 and x.shape[0] >= self.SPLIT_K_ROWS and torch.is_grad_enabled() and self.weight.requires_grad and self.bias is not None:
            return _LinearSplitK.apply(x, self.weight, self.bias)
        return super().forward(x)


def initialize_layer(layer, std=math.sqrt(2), bias_const=0.0):
    nn.init.orthogonal_(layer.weight, std)
    nn.init.constant_(layer.bias, bias_const)
    return layer


def build_network(nodes_counts, std=0.01):
    """[Linear, Tanh, ..., Linear]: one Linear per consecutive pair of `nodes_counts`, Tanh between them; the last
    Linear is initialised with gain `std`."""
    layers = []
    last = len(nodes_counts) - 2
    for k in range(last + 1):
        lin = Linear(int(nodes_counts[k]), int(nodes_counts[k + 1]))
        layers.append(initialize_layer(lin, std) if k == last else initialize_layer(lin))
        if k != last:
            layers.append(nn.Tanh())
    return layers


class Agent(nn.Module):
    def __init__(self, envs, nodes_counts):
        super().__init__()
        input_dim = int(np.prod(envs.single_observation_space.shape))
        self.critic_nodes = [input_dim] + list(nodes_counts) + [1]
        self.actor_nodes = [input_dim] + list(nodes_counts) + [int(envs.single_action_space.n)]
        self.critic = nn.Sequential(*build_network(self.critic_nodes, 1.0))
        self.actor = nn.Sequential(*build_network(self.actor_nodes, 0.01))

    def get_value(self, x):
        return self.critic(x)

    def get_action_and_value(self, x, action=None):
        """-> (action, log-probability of the action, entropy of the policy, value)"""
        dist = Categorical(logits=self.actor(x), validate_args=False)  # (the validation reads the logits back: a device synchronisation per call)
        if action is None:
            action = dist.sample()
        return action, dist.log_prob(action), dist.entropy(), self.critic(x)

    def sample_action_and_value(self, x):
        """`get_action_and_value(x)` without a host synchronisation, so that a rollout step can be captured into a hipGraph
        (torch's Categorical validates its arguments with a device-to-host read).  The action is drawn with the Gumbel-max
        trick -- argmax(log p + G), G ~ Gumbel(0, 1), is an exact sample of Categorical(p) -- from the same logits; the
        log-probability, entropy and value are the same expressions.  -> (action, log-probability, entropy, value)"""
        logp = torch.log_softmax(self.actor(x).float(), dim=-1)
        u = torch.rand_like(logp).clamp_(min=1e-20, max=1.0 - 1e-7)
        action = torch.argmax(logp - torch.log(-torch.log(u)), dim=-1)
        return action, logp.gather(-1, action.unsqueeze(-1)).squeeze(-1), -(logp.exp() * logp).sum(-1), self.critic(x)
