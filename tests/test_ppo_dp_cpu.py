"""Data-parallel PPO on CPU: two gloo processes run `ppo_training_loop` (the same code path RCCL drives on the GPUs) over
a vector environment stepped by the CPU oracle, with different states and random streams per rank.  What must hold:
nobody hangs (the ranks take the same early-stop / KL-penalty decisions, so their per-minibatch all-reduces pair up) and
every rank ends with bit-identical weights; the curriculum deals the initial states rank::world and shares what is solved."""
import os
import socket
from types import SimpleNamespace

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

L = 7
AK2 = [1, 1, -2, -2, -2, 0, 0, 1, 2, 1, -2, -1, -2, 0]


def _states():
    """A few easy presentations at L = 7 (some are solved within a handful of random moves)."""
    base = [[1, 0, 0, 0, 0, 0, 0, 2, 1, 0, 0, 0, 0, 0], [1, 2, 0, 0, 0, 0, 0, 2, 0, 0, 0, 0, 0, 0], AK2,
            [1, 1, 2, 0, 0, 0, 0, 2, 1, 0, 0, 0, 0, 0], [2, 1, -2, 0, 0, 0, 0, 2, 1, 1, 0, 0, 0, 0], [1, -2, 0, 0, 0, 0, 0, 2, 2, 1, 0, 0, 0, 0],
            [-1, 2, 0, 0, 0, 0, 0, 2, 0, 0, 0, 0, 0, 0], [1, 0, 0, 0, 0, 0, 0, -2, 1, 1, 0, 0, 0, 0]]
    return [np.array(b, np.int8) for b in base]


class OracleVecEnv:
    """The slice of ACVecEnv the training loop uses, stepped by oracle/ac_oracle.c on the CPU (test stand-in only)."""

    def __init__(self, rows, horizon):
        from oracle import ac_oracle as O

        self.O = O
        self.init = np.ascontiguousarray(rows, np.int8).copy()
        self.state = self.init.copy()
        self.count = np.zeros(len(rows), np.int32)
        self.horizon = horizon
        self.max_reward = horizon * L * 2
        self.single_observation_space = SimpleNamespace(shape=(2 * L,))
        self.single_action_space = SimpleNamespace(n=12, shape=())
        self.hist = [[] for _ in rows]
        self.last = [[] for _ in rows]

    def reset(self):
        self.state[:] = self.init
        self.count[:] = 0
        return torch.as_tensor(self.state.astype(np.float32)), {}

    def step(self, action, out=None, check_errors=True):
        a = action.cpu().numpy().astype(np.uint8)[None]
        rew, done, trunc, err = self.O.env_rollout(self.state, self.count, self.horizon, np.ascontiguousarray(a))
        assert not err.any()
        for i, x in enumerate(a[0]):
            self.hist[i].append(int(x))
        fin = (done[0] | trunc[0]).astype(bool)
        o, r, t, tr = out
        r.copy_(torch.as_tensor(np.clip(rew[0], -10, 1000).astype(np.float32)))
        t.copy_(torch.as_tensor(done[0].astype(bool)))
        tr.copy_(torch.as_tensor(trunc[0].astype(bool)))
        for i in np.nonzero(fin)[0]:  # autoreset to the env's initial state, as ACVecEnv does
            self.last[i], self.hist[i] = self.hist[i], []
            self.state[i] = self.init[i]
            self.count[i] = 0
        o.copy_(torch.as_tensor(self.state.astype(np.float32)))

    def reset_envs(self, idx, rows):
        for i, row in zip(idx, rows):
            self.init[i] = row
            self.state[i] = row
            self.count[i] = 0

    def get_actions(self, i, finished=False):
        return list(self.last[i] if finished else self.hist[i])

    def _raise_on_errors(self):
        pass


def _worker(rank, world, port, q, is_loss_clip):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ac_solver.agents.args import parse_args
        from ac_solver.agents.ppo_agent import Agent
        from ac_solver.agents.training import ppo_training_loop

        # target_kl far below any real KL: with the clipped objective every rank must leave the epoch loop after the
        # SAME epoch; with the KL penalty every rank must double beta together
        args = parse_args(["--num-envs", "3", "--num-steps", "24", "--total-timesteps", str(3 * 24 * 3), "--update-epochs", "3", "--num-minibatches", "2",
                           "--horizon-length", "12", "--target-kl", "1e-7" if is_loss_clip else "0.01", "--is-loss-clip", str(is_loss_clip),
                           "--nodes-counts", "16", "16", "--seed", str(1 + rank)])
        torch.manual_seed(args.seed)
        np.random.seed(args.seed)
        states = _states()
        curr = [(rank + i * world) % len(states) for i in range(args.num_envs)]  # what get_env deals
        envs = OracleVecEnv(np.stack(states)[curr], args.horizon_length)
        agent = Agent(envs, args.nodes_counts)
        for p in agent.parameters():
            dist.broadcast(p.data, 0)
        opt = torch.optim.Adam(agent.parameters(), lr=args.learning_rate, eps=args.epsilon)
        rec = {"solved": set(), "unsolved": set(range(len(states)))}
        processed, hist = set(curr), {}
        ppo_training_loop(envs, args, torch.device("cpu"), opt, agent, curr, rec, hist, processed, states, progress=False)
        flat = torch.cat([p.detach().reshape(-1) for p in agent.parameters()]).numpy()
        q.put((rank, flat, sorted(rec["solved"]), sorted(processed), sorted(hist)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("is_loss_clip", [True, False])
def test_two_ranks_finish_with_identical_weights(is_loss_clip):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, is_loss_clip)) for r in range(2)]
    [p.start() for p in procs]
    got = {}
    for _ in range(2):
        r, flat, solved, processed, hist = q.get(timeout=300)
        got[r] = (flat, solved, processed, hist)
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert np.array_equal(got[0][0], got[1][0]), "ranks ended with different weights"
    assert np.isfinite(got[0][0]).all()
    assert got[0][1] == got[1][1], "the solved sets were not shared"
    # the states were dealt rank::world: during the first round a rank only starts states of its own residue class,
    # afterwards it samples from the shared record
    assert {0, 2, 4} <= set(got[0][2]) and {1, 3, 5} <= set(got[1][2])
    # a rank records move histories only for states it solved itself; together they cover the shared solved set
    assert set(got[0][3]) | set(got[1][3]) == set(got[0][1])


def test_tile_initial_states_flag_and_dealing():
    from ac_solver.agents.args import parse_args

    a = parse_args(["--tile-initial-states", "--num-envs", "5000"])
    assert a.tile_initial_states is True and parse_args([]).tile_initial_states is False
