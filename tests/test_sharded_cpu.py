"""CPU tests of the sharded-BFS orchestration (world size > 1): gloo processes and in-process threads,
with the NumPy/oracle engine standing in for the per-GPU HIP engine.  The result must equal the
reference's bfs (via the oracle) for every world size."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import ac_oracle as O
from tests.shard_helpers import OracleShardEngine, run_threads

AK2 = [1, 1, -2, -2, -2, 0, 0, 1, 2, 1, -2, -1, -2, 0]
MS = [-1, 2, 1, -2, -2, 0, 0, 0, 0, 0, -1, 2, 1, 2, 0, 0, 0, 0, 0, 0]
CASES = [(AK2, 100000, False), (AK2, 10, False), (AK2, 500, False), (AK2, 1, False), (AK2, 3000, True), (MS, 2000, False), (MS, 40, True)]


def _run(comm, batch):
    from ac_solver.search.sharded import bfs_sharded

    res = []
    for p, budget, cyc in CASES:
        res.append(bfs_sharded(p, budget, cyclically_reduce_after_moves=cyc, comm=comm, engine_factory=OracleShardEngine,
                               batch_parents=batch, want_stats=True))
    return res


def _check(results):
    for (p, budget, cyc), (ok, path, st) in zip(CASES, results):
        wok, wpath, wst = O.bfs(p, budget, cyclically_reduce_after_moves=cyc, stats=True)
        assert (ok, path) == (wok, wpath), (budget, cyc)
        assert st["nodes"] == wst["nodes"] and st["expanded"] == wst["expanded"], (budget, cyc, st, wst)


@pytest.mark.parametrize("world,batch", [(1, 1 << 18), (2, 1 << 18), (3, 7), (4, 64)])
def test_thread_ranks_match_reference(world, batch):
    from ac_solver.search.sharded import SingleComm

    if world == 1:
        _check(_run(SingleComm(), batch))
        return
    for res in run_threads(world, lambda comm: _run(comm, batch)):
        _check(res)


def _gloo_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ac_solver.search.sharded import TorchDistComm

        res = _run(TorchDistComm(torch.device("cpu")), 50)
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


def test_gloo_world2_matches_reference():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    got = dict(q.get(timeout=240) for _ in range(2))
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    _check(got[0])
    assert got[0] == got[1]  # every rank returns the same answer


def test_a_failing_rank_takes_every_rank_down_without_deadlock():
    """A rank whose engine fails (capacity, device error) keeps taking part in the collectives of the chunk and reports
    through the `solved` all-reduce, so that every rank raises instead of waiting forever."""
    from ac_solver.search.sharded import bfs_sharded

    class Flaky(OracleShardEngine):
        def insert(self, recv, c0, n_parents):
            if self.rank == 1 and len(self.states) > 20:
                raise RuntimeError("engine capacity exceeded (simulated)")
            return super().insert(recv, c0, n_parents)

    def run(comm):
        try:
            bfs_sharded(AK2, 100000, comm=comm, engine_factory=Flaky, batch_parents=64)
        except RuntimeError as e:
            return str(e)
        return "no error"

    msgs = run_threads(3, run)
    assert all("sharded bfs failed" in m for m in msgs), msgs
    assert "simulated" in msgs[1]
