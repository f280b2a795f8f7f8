"""Actor and critic networks (reference: ac_solver/agents/ppo_agent.py:11-109): two tanh MLPs over the
2L-entry observation, orthogonal initialisation (gain sqrt(2); 0.01 for the policy head, 1.0 for the value head)."""
import math

import numpy as np
import torch
from torch import nn
from torch.distributions import Categorical


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
        lin = nn.Linear(int(nodes_counts[k]), int(nodes_counts[k + 1]))
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
