"""Train a PPO agent on the AC environment (reference: ac_solver/agents/ppo.py:23-62):

    python -m ac_solver.agents.ppo --num-envs 1024 --num-steps 200 --total-timesteps 2000000

One process per GPU under torchrun (`python -m torch.distributed.run --nproc-per-node 8 -m ac_solver.agents.ppo ...`):
every rank rolls out its own `--num-envs` environments and the gradients are averaged over RCCL.
All flags: ac_solver/agents/args.py.
"""
import os
import random

import numpy as np
import torch
from torch.optim import Adam

from ac_solver.agents.args import parse_args
from ac_solver.agents.environment import get_env
from ac_solver.agents.ppo_agent import Agent
from ac_solver.agents.training import ppo_training_loop


def train_ppo(argv=None):
    args = parse_args(argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not (torch.cuda.is_available() and args.cuda):
        raise SystemExit("ac_solver.agents.ppo steps its environments with HIP kernels: a GPU (and --cuda true) is required")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.distributed.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    args.seed += rank  # every rank explores with its own random stream
    random.seed(args.seed)
    np.random.seed(args.seed)
    torch.manual_seed(args.seed)
    torch.backends.cudnn.deterministic = args.torch_deterministic

    envs, initial_states, curr_states, success_record, ACMoves_hist, states_processed = get_env(args, device=device, rank=rank, world=world)
    agent = Agent(envs, args.nodes_counts).to(device)
    if world > 1:  # identical initial weights on every rank
        for p in agent.parameters():
            torch.distributed.broadcast(p.data, 0)
    optimizer = Adam(agent.parameters(), lr=args.learning_rate, eps=args.epsilon)
    stats = ppo_training_loop(envs, args, device, optimizer, agent, curr_states, success_record, ACMoves_hist, states_processed,
                              initial_states)
    envs.close()
    if world > 1:
        torch.distributed.destroy_process_group()
    return stats


if __name__ == "__main__":
    train_ppo()
