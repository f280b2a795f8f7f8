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


SMALL = [(AK2, 4000, False), (AK2, 10, False), (AK2, 500, False), (AK2, 1, False), (AK2, 1500, True), (MS, 2000, False), (MS, 40, True)]


def _strip(results):
    """(ok, path, stats without the wall-clock fields)"""
    return [(ok, path, {k: v for k, v in st.items() if not k.endswith("_seconds") and k != "local_nodes"}) for ok, path, st in results]


@pytest.fixture(autouse=True)
def _every_level_exchanged_unless_asked(monkeypatch):
    """The tests of this module that do not say otherwise run the exchange from the root on (rounds 3-5: `replicate_below` 0); the
    replicated phase of round 6 is what the tests that pass `replicate_below` themselves exercise."""
    from ac_solver.search import sharded

    monkeypatch.setattr(sharded, "REPLICATE_BELOW", 0)


def _run(comm, batch, cases=None, repl=0):
    from ac_solver.search.sharded import bfs_sharded

    res = []
    for p, budget, cyc in cases or CASES:
        res.append(bfs_sharded(p, budget, cyclically_reduce_after_moves=cyc, comm=comm, engine_factory=OracleShardEngine,
                               batch_parents=batch, want_stats=True, replicate_below=repl))
    return res


def _check(results, cases=None):
    for (p, budget, cyc), (ok, path, st) in zip(cases or CASES, results):
        wok, wpath, wst = O.bfs(p, budget, cyclically_reduce_after_moves=cyc, stats=True)
        assert (ok, path) == (wok, wpath), (budget, cyc)
        assert st["nodes"] == wst["nodes"] and st["expanded"] == wst["expanded"], (budget, cyc, st, wst)


@pytest.mark.parametrize("world,batch,repl", [(1, 1 << 18, 0), (2, 1 << 18, 0), (3, 7, 0), (4, 64, 0), (2, 1 << 18, 1 << 18), (2, 64, 6), (3, 7, 40), (4, 64, 150)])
def test_thread_ranks_match_reference(world, batch, repl):
    """`repl` = replicate_below: 0 exchanges every level; 2^18 keeps these searches replicated to their end (ONE all-reduce per search);
    6 / 40 / 150 partition the frontier after a few levels, in the middle of the search"""
    from ac_solver.search.sharded import SingleComm

    if world == 1:
        _check(_run(SingleComm(), batch))
        return
    cases = CASES if batch > 1000 else SMALL  # (the NumPy engine is slow: tiny chunks get smaller budgets)
    for res in run_threads(world, lambda comm: _run(comm, batch, cases, repl)):
        _check(res, cases)
        for k, ((p, budget, cyc), (ok, path, st)) in enumerate(zip(cases, res)):
            if repl >= 1 << 18:  # (the communicator's counters run on from search to search)
                assert st["comm_all_to_all_calls"] == 0 and st["comm_all_reduce_calls"] == k + 1 and st["replicated_levels"] == st["levels"], st
            elif repl and st["levels"] > st["replicated_levels"]:
                assert st["replicated_levels"] >= 1 and st["comm_all_to_all_calls"] > 0, st


def test_replicated_phase_saves_the_small_levels_collectives():
    """the same search with every level exchanged and with the levels below 40 parents replicated: identical result, fewer collectives"""
    def work(comm):
        a = _run(comm, 64, SMALL[:1], 0)[0]
        b = _run(comm, 64, SMALL[:1], 40)[0]
        return a, b

    for a, b in run_threads(3, work):
        assert a[:2] == b[:2] and a[2]["nodes"] == b[2]["nodes"] and a[2]["expanded"] == b[2]["expanded"]
        assert b[2]["replicated_levels"] >= 3
        # (the communicator's counters run on from search to search: b's are a's + its own)
        assert b[2]["comm_all_reduce_calls"] - a[2]["comm_all_reduce_calls"] < a[2]["comm_all_reduce_calls"]
        assert b[2]["comm_all_to_all_calls"] - a[2]["comm_all_to_all_calls"] < a[2]["comm_all_to_all_calls"]


@pytest.mark.parametrize("where", ["chunk_expand", "chunk_insert", "chunk_commit", "partition"])
def test_a_failure_in_the_replicated_phase_ends_every_rank(where):
    """An engine call of ONE rank raises while the levels are still replicated (no per-level collective pairs the ranks there), or in
    the partition itself: every rank must raise, with the same collectives issued -- the healthy ranks at the phase's closing all-reduce
    or, for the partition, at the first exchanged chunk's headers."""
    from ac_solver.search.sharded import bfs_sharded

    for fail_at in (1, 2, 5):
        calls = {}

        class Flaky(OracleShardEngine):
            def _maybe(self, name):
                if name == where and self.rank == 1:
                    calls[name] = calls.get(name, 0) + 1
                    if calls[name] == fail_at:
                        raise RuntimeError("engine call failed (simulated)")

            def chunk_expand(self, *a, **k):
                self._maybe("chunk_expand")
                return super().chunk_expand(*a, **k)

            def chunk_insert(self, n_par):
                self._maybe("chunk_insert")
                return super().chunk_insert(n_par)

            def chunk_commit(self, max_nodes):
                self._maybe("chunk_commit")
                return super().chunk_commit(max_nodes)

            def partition(self):
                self._maybe("partition")
                return super().partition()

        def run(comm):
            try:
                bfs_sharded(AK2, 4000, comm=comm, engine_factory=Flaky, batch_parents=32, replicate_below=60)
            except RuntimeError as e:
                return str(e), dict(comm.stats)
            return "no error", dict(comm.stats)

        out = run_threads(3, run)
        msgs = [m for m, _ in out]
        if calls.get(where, 0) < fail_at:  # (one partition per search)
            assert all(m == "no error" for m in msgs), (fail_at, msgs)
            continue
        assert all("sharded bfs failed" in m for m in msgs), (where, fail_at, msgs)
        assert "simulated" in msgs[1], (where, fail_at, msgs)
        assert out[0][1] == out[1][1] == out[2][1], (where, fail_at, [st for _, st in out])


def _gloo_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ac_solver.search.sharded import TorchDistComm

        res = _run(TorchDistComm(torch.device("cpu")), 50)
        res_repl = _run(TorchDistComm(torch.device("cpu")), 50, SMALL, 25)  # small levels replicated, the frontier partitioned at 25 parents
        # the mask all-reduce on a communicator of its own (dist.new_group): what bench.py times next to the shared one on N > 1 GPUs
        own = TorchDistComm(torch.device("cpu"), mask_group="own")
        res_own = _run(own, 50, SMALL)
        assert own.stats["mask_all_reduce_calls"] > 0 and own.mask_group is not own.group
        q.put((rank, res, res_own, res_repl))
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
    rows = [q.get(timeout=400) for _ in range(2)]
    got = {r: a for r, a, _, _ in rows}
    got_own = {r: b for r, _, b, _ in rows}
    got_repl = {r: c for r, _, _, c in rows}
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    _check(got[0])
    assert _strip(got[0]) == _strip(got[1])  # every rank returns the same answer
    _check(got_own[0], SMALL)
    _check(got_repl[0], SMALL)
    assert _strip(got_repl[0]) == _strip(got_repl[1])
    assert any(0 < st["replicated_levels"] < st["levels"] for _, _, st in got_repl[0])  # some search was partitioned in mid-flight
    strip_comm = lambda res: [(ok, path, {k: v for k, v in st.items() if not k.startswith("comm_")}) for ok, path, st in _strip(res)]  # noqa: E731
    assert strip_comm(got_own[0]) == strip_comm(got_own[1])


def test_a_failing_rank_takes_every_rank_down_without_deadlock():
    """A rank whose engine call fails (a HIP error, an exhausted allocation) keeps taking part in the collectives of the chunk and
    marks itself failed; the flag travels in the headers of its next chunk, every rank stops at that chunk and raises."""
    from ac_solver.search.sharded import bfs_sharded

    class Flaky(OracleShardEngine):
        def chunk_insert(self, n_par):
            if self.rank == 1 and len(self.states) > 20:
                raise RuntimeError("engine capacity exceeded (simulated)")
            return super().chunk_insert(n_par)

    def run(comm):
        try:
            bfs_sharded(AK2, 100000, comm=comm, engine_factory=Flaky, batch_parents=64)
        except RuntimeError as e:
            return str(e)
        return "no error"

    msgs = run_threads(3, run)
    assert all("sharded bfs failed" in m for m in msgs), msgs
    assert "simulated" in msgs[1]


@pytest.mark.timeout(300)
@pytest.mark.parametrize("lag", [0, 2])
@pytest.mark.parametrize("where", ["chunk_expand", "chunk_insert", "chunk_commit"])
@pytest.mark.parametrize("overlap", ["insert", "commit"])
def test_a_host_side_failure_in_any_engine_call_ends_every_rank_at_the_same_collective(where, lag, overlap):
    """An engine call that raises on ONE rank (chunk_expand before its chunk exists in the engine, chunk_insert, chunk_commit), with
    the control block read `lag` chunks late as on the GPU: every rank must issue exactly the same collectives and raise -- a rank
    left alone in a collective would hang this test (thread ranks meet at barriers).  The failing call is placed in the middle of
    a level with many chunks, at a level's last chunk and in the search's last chunks."""
    from ac_solver.search import sharded
    from ac_solver.search.sharded import bfs_sharded

    for fail_at in (3, 9, 10, 17, 30):
        calls = {}

        class Flaky(OracleShardEngine):
            def _maybe(self, name):
                if name == where and self.rank == 1:
                    calls[name] = calls.get(name, 0) + 1
                    if calls[name] == fail_at:
                        raise RuntimeError("engine call failed (simulated)")

            def chunk_expand(self, *a, **k):
                self._maybe("chunk_expand")
                return super().chunk_expand(*a, **k)

            def chunk_insert(self, n_par):
                self._maybe("chunk_insert")
                return super().chunk_insert(n_par)

            def chunk_commit(self, max_nodes):
                self._maybe("chunk_commit")
                return super().chunk_commit(max_nodes)

        def run(comm):
            try:
                bfs_sharded(AK2, 600, comm=comm, engine_factory=Flaky, batch_parents=8, overlap=overlap)
            except RuntimeError as e:
                return str(e), dict(comm.stats)
            return "no error", dict(comm.stats)

        sharded._FORCE_LAG = lag
        try:
            out = run_threads(3, run)
        finally:
            sharded._FORCE_LAG = None
        msgs = [m for m, _ in out]
        if calls.get(where, 0) < fail_at:  # the search ended before the failing call: nothing to see
            assert all(m == "no error" for m in msgs), (fail_at, msgs)
            continue
        assert all("sharded bfs failed" in m for m in msgs), (fail_at, msgs)
        assert "simulated" in msgs[1], (fail_at, msgs)
        assert out[0][1] == out[1][1] == out[2][1], (fail_at, [st for _, st in out])  # the same number of each collective on every rank


@pytest.mark.timeout(300)
@pytest.mark.parametrize("lag", [0, 2])
@pytest.mark.parametrize("where", ["chunk_expand", "chunk_insert", "chunk_commit"])
def test_a_sticky_engine_failure_ends_every_rank_at_the_same_collective(where, lag):
    """What a HIP fault looks like: from the first failing call on EVERY call of that rank's engine raises -- the control-block
    snapshot and wait included, which the failing rank still makes in order to leave the chunk loop in step with the others.  The
    orchestrator must keep pairing the collectives (dead chunks, guarded control-block calls) and every rank must raise."""
    from ac_solver.search import sharded
    from ac_solver.search.sharded import bfs_sharded

    for fail_at in (3, 10, 17):
        calls = {}
        broken = []

        class Sticky(OracleShardEngine):
            def _maybe(self, name):
                if self.rank != 1:
                    return
                if name == where:
                    calls[name] = calls.get(name, 0) + 1
                    if calls[name] == fail_at:
                        broken.append(name)
                if broken:
                    raise RuntimeError("device lost (simulated)")

            def chunk_expand(self, *a, **k):
                self._maybe("chunk_expand")
                return super().chunk_expand(*a, **k)

            def chunk_insert(self, n_par):
                self._maybe("chunk_insert")
                return super().chunk_insert(n_par)

            def chunk_insert_dead(self, n_par):
                self._maybe("chunk_insert_dead")
                return super().chunk_insert_dead(n_par)

            def chunk_commit(self, max_nodes):
                self._maybe("chunk_commit")
                return super().chunk_commit(max_nodes)

            def ctl_snapshot(self, slot):
                self._maybe("ctl_snapshot")
                return super().ctl_snapshot(slot)

            def ctl_wait(self, slot):
                self._maybe("ctl_wait")
                return super().ctl_wait(slot)

            def fail_local(self):
                self._maybe("fail_local")
                return super().fail_local()

        def run(comm):
            try:
                bfs_sharded(AK2, 600, comm=comm, engine_factory=Sticky, batch_parents=8)
            except RuntimeError as e:
                return str(e), dict(comm.stats)
            return "no error", dict(comm.stats)

        sharded._FORCE_LAG = lag
        try:
            out = run_threads(3, run)
        finally:
            sharded._FORCE_LAG = None
        msgs = [m for m, _ in out]
        if not broken:
            assert all(m == "no error" for m in msgs), (fail_at, msgs)
            continue
        assert all("sharded bfs failed" in m for m in msgs), (fail_at, msgs)
        assert "simulated" in msgs[1], (fail_at, msgs)
        assert out[0][1] == out[1][1] == out[2][1], (fail_at, [st for _, st in out])


def test_a_device_side_capacity_failure_reaches_every_rank():
    """node capacity exhausted on one rank: refused on that rank before anything is written, carried to the others by the next
    chunk's headers (or by the closing all-reduce when the search ends first)"""
    from ac_solver.search.sharded import bfs_sharded

    class Small(OracleShardEngine):
        def __init__(self, *a):
            super().__init__(*a)
            if self.rank == 0:
                self.node_cap = 30

    def run(comm):
        try:
            bfs_sharded(AK2, 100000, comm=comm, engine_factory=Small, batch_parents=16)
        except RuntimeError as e:
            return str(e)
        return "no error"

    msgs = run_threads(2, run)
    assert "node capacity" in msgs[0] and "another rank" in msgs[1], msgs


def test_layout_mirror_matches_the_library():
    """tests/shard_helpers.py:layout (what the NumPy engine uses) == acx_shard_layout (pure host arithmetic, no GPU needed)"""
    import ctypes as C

    from ac_solver import _acx
    from tests.shard_helpers import layout

    s, cap, rw = C.c_int64(), C.c_int64(), C.c_int64()
    for world in (1, 2, 3, 8, 64):
        for n_par in (1, 7, 85, 86, 1000, 1365, 1366, 1 << 18, (1 << 21) + 5):
            for kw in (2, 4):
                for fill in (0, 24, 64, 121, 320, 999, 1 << 20, (1 << 20) + 5):
                    _acx.check(_acx.lib.acx_shard_layout(n_par, world, kw, fill, C.byref(s), C.byref(cap), C.byref(rw)))
                    assert (s.value, cap.value, rw.value) == layout(n_par, world, kw, fill), (world, n_par, kw, fill)


def _random_reduced(rng, n):
    w = []
    while len(w) < n:
        c = int(rng.integers(0, 4))
        if not w or w[-1] != c ^ 3:
            w.append(c)
    return w


def test_owner_function_python_mirror_matches_the_library():
    """sharded.owner_of (what the NumPy engine and the orchestrator's root placement use) == acx_shard_owner (csrc/acx_owner.h
    compiled for the host: the arithmetic the kernels run), both key widths, reduced and unreduced words"""
    import ctypes as C

    from ac_solver import _acx
    from ac_solver.search.sharded import owner_of

    rng = np.random.default_rng(5)
    for L, half in ((7, 1), (25, 1), (29, 1), (36, 2), (61, 2)):
        bits = 64 * half
        for _ in range(400):
            row = []
            for _r in range(2):
                n = int(rng.integers(1, L + 1))
                w = _random_reduced(rng, n) if rng.random() < 0.8 else [int(c) for c in rng.integers(0, 4, n)]
                k = sum(c << (2 * i) for i, c in enumerate(w)) | (n << (bits - 6))
                for j in range(half):
                    v = (k >> (64 * j)) & 0xFFFFFFFFFFFFFFFF
                    row.append(v - (1 << 64) if v >= (1 << 63) else v)
            row = np.array(row, np.int64)
            for world in (1, 2, 3, 8, 64):
                assert _acx.lib.acx_shard_owner(L, _acx.ptr(row, C.c_int64), world) == int(owner_of(row[None], world)[0]), (L, row, world)


@pytest.mark.parametrize("cyclical", [False, True])
def test_conjugation_moves_keep_the_class_hashes(cyclical):
    """What the HIP engine relies on (csrc/acx_owner.h): in a normal-form search the eight conjugation children of a node (actions
    4 .. 11) have the node's own class hashes, a move leaves the class hash AND the inner letter of the relator it does not rewrite
    alone, and a conjugation only changes the inner letter of its relator while the conjugator u (r = u c u^-1) has at most one
    letter -- so most children have their parent's owner.  Checked with the oracle's ACMove on every node of a few searches' trees,
    tight max_relator_length included (where conjugations stop fitting and children come back unchanged)."""
    from ac_solver.search.sharded import class_hash, conj_prefix, inner_letter, owner_of
    from tests.shard_helpers import _CODE, key_of_state

    def classes(state, L):
        return [class_hash([_CODE[int(a)] for a in state[h * L:(h + 1) * L] if a != 0]) for h in (0, 1)]

    def inners(state, L):
        words = [[_CODE[int(a)] for a in state[h * L:(h + 1) * L] if a != 0] for h in (0, 1)]
        return [inner_letter(w) for w in words], [conj_prefix(w) for w in words]

    concat_moved = conj_total = conj_stayed = 0
    for p, budget in ((AK2, 400), (MS, 400), ([1, 1, 1, -2, -2, -2, -2, 0, 0, 0, 1, 2, 1, -2, -1, -2, 0, 0, 0, 0], 300)):
        p = np.array(p, np.int8)
        L = len(p) // 2
        seen, queue = {tuple(p.tolist())}, [p]
        while queue and len(seen) < budget:
            st = queue.pop(0)
            pc = classes(st, L)
            pin, pp = inners(st, L)
            po = int(owner_of(np.array([key_of_state(st, L)], np.int64), 8)[0])
            out, lens, errs = O.move_batch(np.repeat(st[None], 12, axis=0), np.arange(12, dtype=np.uint8), L, cyclical=cyclical)
            for a in range(12):
                cc = classes(out[a], L)
                untouched = 0 if a % 2 == 0 else 1  # even action ids rewrite r_1 (ac_moves.py:192-206)
                cin, _ = inners(out[a], L)
                assert cc[untouched] == pc[untouched] and cin[untouched] == pin[untouched], (st, a)
                if a >= 4:
                    assert cc == pc, (st, a, out[a])
                    if pp[1 - untouched] >= 2:  # a conjugator of two letters or more keeps its last letter
                        assert cin == pin, (st, a, out[a])
                    if not np.array_equal(out[a], st):
                        conj_total += 1
                        conj_stayed += int(owner_of(np.array([key_of_state(out[a], L)], np.int64), 8)[0]) == po
                elif cc != pc:
                    concat_moved += 1
                key = tuple(out[a].tolist())
                if key not in seen:
                    seen.add(key)
                    queue.append(out[a])
    assert concat_moved > 100  # (the concatenations DO change the class: the function is not constant)
    assert conj_stayed > 0.6 * conj_total, (conj_stayed, conj_total)  # most conjugation children stay on their parent's rank (of 8)


def test_undo_children_are_visited_states():
    """world 1: every child the engines do not send because it undoes its parent's move (normal-form root, cyclical = False) is
    a visited state -- asserted inside OracleShardEngine.chunk_expand, where the single rank holds the whole visited set"""
    from ac_solver.search.sharded import SingleComm, bfs_sharded

    made = []

    def factory(*a):
        made.append(OracleShardEngine(*a))
        return made[-1]

    for p, L in ((AK2, 7), (MS, 10)):
        ok, path, st = bfs_sharded(p, 3000, comm=SingleComm(), engine_factory=factory, batch_parents=97, want_stats=True)
        wok, wpath, wst = O.bfs(p, 3000, stats=True)
        assert (ok, path, st["nodes"], st["expanded"]) == (wok, wpath, wst["nodes"], wst["expanded"])
        assert made[-1].inverse_dropped > 100


def test_adaptive_regions_and_the_rerun_after_an_overflow():
    """world 2-3 thread ranks over the NumPy engine: (a) the adaptive region capacity (the previous level's fullest region x 1.5)
    leaves the result untouched and is really used (the engine's threshold for "a chunk large enough to tell" is lowered to
    the sizes of this test); (b) so does a given capacity."""
    from ac_solver.search.sharded import bfs_sharded

    def factory(*a):
        eng = OracleShardEngine(*a)
        eng.fill_min_even = 1
        return eng

    for world in (2, 3):
        for (p, budget, cyc) in SMALL[:3]:
            wok, wpath, wst = O.bfs(p, budget, cyclically_reduce_after_moves=cyc, stats=True)

            def work(comm):
                a = bfs_sharded(p, budget, cyclically_reduce_after_moves=cyc, comm=comm, engine_factory=factory, batch_parents=64, want_stats=True)
                b = bfs_sharded(p, budget, cyclically_reduce_after_moves=cyc, comm=comm, engine_factory=factory, batch_parents=64, want_stats=True, region_fill=1)
                return a, b

            for a, b in run_threads(world, work):
                for ok, path, st in (a, b):
                    assert (ok, path) == (wok, wpath) and st["nodes"] == wst["nodes"] and st["expanded"] == wst["expanded"], (world, budget, cyc)
                assert "region_overflow_reruns" not in a[2]  # (chunks this small never exceed the two tiles of slack: the overflow itself is
                # forced in tests/test_gpu_search.py::test_sharded_bfs_mid_size_thread_ranks_equal_the_fused_search)
