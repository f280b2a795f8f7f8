"""Differential fuzz of the device frontiers against the CPU oracle on random presentations (not only presentations of
the trivial group: searches in which a move empties a relator must raise like the reference), random budgets, both
algorithms, both cyclic-reduction settings, 64-bit and 128-bit keys."""
import os

import numpy as np
import pytest

SEED = int(os.environ.get("ACX_FUZZ_SEED", "0"))  # soak runs: other seeds, other cases

pytestmark = pytest.mark.gpu


def _random_word(rng, n):
    w = []
    while len(w) < n:
        c = int(rng.choice([1, -1, 2, -2]))
        if not w or w[-1] != -c:
            w.append(c)
    return w


@pytest.mark.timeout(600)
@pytest.mark.parametrize("L", [6, 12, 25, 33])
def test_random_presentations_against_oracle(L):
    from ac_solver import _acx
    from ac_solver.search._common import run_search
    from oracle import ac_oracle as O

    _acx.require_device()
    rng = np.random.default_rng(100 + L + 1000 * SEED)
    n_cases = 40 if L <= 12 else 24
    raised = solved = 0
    for case in range(n_cases):
        row = np.zeros(2 * L, np.int8)
        for h in (0, 1):
            w = _random_word(rng, int(rng.integers(1, min(L, 9) + 1)))
            row[h * L:h * L + len(w)] = w
        budget = int(rng.choice([1, 2, 7, 60, 500, 4000, 30000]))
        cyc = bool(rng.integers(0, 2))
        for kind, ofn in ((_acx.SEARCH_BFS, O.bfs), (_acx.SEARCH_GREEDY, O.greedy_search)):
            try:
                want = ofn(row, budget, cyclically_reduce_after_moves=cyc, stats=True)
            except AssertionError:
                want = "raises"
            except IndexError:
                want = "raises"
            try:
                ok, path, st = run_search(kind, row, budget, cyc)
                got = (ok, path, {"nodes": st["nodes"], "expanded": st["expanded"]})
            except (AssertionError, IndexError):
                got = "raises"
            if want == "raises":
                raised += 1
                assert got == "raises", (L, case, kind, budget, cyc, row.tolist())
            else:
                wok, wpath, wst = want
                solved += bool(wok)
                assert got != "raises" and (got[0], got[1]) == (wok, wpath), (L, case, kind, budget, cyc, row.tolist())
                assert got[2]["nodes"] == wst["nodes"] and got[2]["expanded"] == wst["expanded"], (L, case, kind, budget, cyc, row.tolist(), got[2], wst)
    assert raised + solved >= 0


@pytest.fixture
def many_bmax():
    from ac_solver import _acx

    yield
    _acx.lib.acx_set_option(_acx.OPT_BFS_MANY_BMAX, -1)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("L", [6, 25, 33])
def test_random_presentation_groups_through_search_many(L, many_bmax):
    """acx_search_many on groups of random presentations (roots in and out of normal form in one group, so both move codes; small
    batches per round so that the searches of a group are many rounds apart): every search as the oracle's, for bfs (the searches
    share their launches, acx_bfs_many.h) and greedy_search (one workgroup per search); a group with a search in which the
    reference raises makes the call raise."""
    from ac_solver import _acx
    from ac_solver.search._common import run_search_many
    from oracle import ac_oracle as O

    _acx.require_device()
    rng = np.random.default_rng(500 + L + 1000 * SEED)
    _acx.check(_acx.lib.acx_set_option(_acx.OPT_BFS_MANY_BMAX, int(rng.choice([128, 512, 32768]))))  # (the fixture restores the default)
    for budget, cyc in ((int(rng.choice([1, 7, 60])), False), (500, True), (4000, False), (30000, True)):
        rows = []
        for _ in range(40):
            row = np.zeros(2 * L, np.int8)
            for h in (0, 1):
                w = _random_word(rng, int(rng.integers(1, min(L, 9) + 1)))
                if rng.random() < 0.15 and len(w) + 2 <= L:  # not freely reduced: the general move code
                    w = w[:1] + [1, -1] + w[1:]
                row[h * L:h * L + len(w)] = w
            rows.append(row)
        for kind, ofn in ((_acx.SEARCH_BFS, O.bfs), (_acx.SEARCH_GREEDY, O.greedy_search)):
            want, good, bad = [], [], []
            for row in rows:
                try:
                    want.append(ofn(row, budget, cyclically_reduce_after_moves=cyc, stats=True))
                    good.append(row)
                except (AssertionError, IndexError):
                    bad.append(row)
            got = run_search_many(kind, np.stack(good), budget, cyc)
            for k, ((ok, path, st), (wok, wpath, wst)) in enumerate(zip(got, want)):
                assert (ok, path) == (wok, wpath), (L, kind, budget, cyc, good[k].tolist())
                assert st["nodes"] == wst["nodes"] and st["expanded"] == wst["expanded"], (L, kind, budget, cyc, good[k].tolist(), st, wst)
            if bad:
                with pytest.raises(AssertionError):
                    run_search_many(kind, np.stack(good[:3] + bad[:1] + good[3:6]), budget, cyc)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [1, 3])
def test_random_presentations_sharded_bfs(world):
    """the same comparison through the sharded frontier (HIP engine; thread ranks share the GPU when world > 1)"""
    from ac_solver import _acx
    from ac_solver.search.sharded import SingleComm, bfs_sharded
    from oracle import ac_oracle as O
    from tests.shard_helpers import run_threads

    _acx.require_device()
    rng = np.random.default_rng(7 + 1000 * SEED)
    cases = []
    for _ in range(16):
        L = int(rng.choice([8, 25, 33]))
        row = np.zeros(2 * L, np.int8)
        for h in (0, 1):
            w = _random_word(rng, int(rng.integers(1, 8)))
            row[h * L:h * L + len(w)] = w
        cases.append((row, int(rng.choice([1, 9, 300, 5000, 40000])), bool(rng.integers(0, 2)), int(rng.choice([7, 100, 1 << 14]))))

    def run(comm):
        out = []
        for row, budget, cyc, bp in cases:
            try:
                ok, path, st = bfs_sharded(row, budget, cyclically_reduce_after_moves=cyc, comm=comm, batch_parents=bp, want_stats=True)
                out.append((ok, path, st["nodes"], st["expanded"]))
            except AssertionError:
                out.append("raises")
        return out

    results = [run(SingleComm())] if world == 1 else run_threads(world, run)
    n_raise = 0
    for res in results:
        for (row, budget, cyc, bp), got in zip(cases, res):
            try:
                wok, wpath, wst = O.bfs(row, budget, cyclically_reduce_after_moves=cyc, stats=True)
                want = (wok, wpath, wst["nodes"], wst["expanded"])
            except (AssertionError, IndexError):
                want = "raises"
                n_raise += 1
            assert got == want, (world, budget, cyc, bp, row.tolist())
    assert n_raise >= 0


@pytest.mark.timeout(900)
@pytest.mark.parametrize("L", [6, 25, 33, 63])
def test_random_presentations_through_the_sharded_engine(L, monkeypatch):
    """bfs_sharded on 2 / 3 / 5 thread ranks of the HIP engine (one GPU plays all of them) on random presentations -- roots in and out
    of normal form (the general move code computes every child's owner from scratch, the normal-form codes inherit it along
    conjugations), searches in which a move empties a relator (every rank raises like the reference), tiny chunks so that records of
    a chunk meet born children of the next: result, path and counts as the oracle's, and every node on the rank the owner function
    names (csrc/acx_owner.h)."""
    from ac_solver import _acx
    from ac_solver.search import sharded
    from ac_solver.search.sharded import bfs_sharded
    from oracle import ac_oracle as O
    from tests.shard_helpers import run_threads

    _acx.require_device()
    monkeypatch.setattr(sharded, "_CHECK_OWNERS", True)
    rng = np.random.default_rng(900 + L + 1000 * SEED)
    raised = crossed = 0
    for case in range(10):
        row = np.zeros(2 * L, np.int8)
        for h in (0, 1):
            w = _random_word(rng, int(rng.integers(1, min(L, 9) + 1)))
            if L <= 61 and rng.random() < 0.3 and len(w) + 2 <= L:  # not freely reduced: the general move code (keys of 62 .. 64 letters need a reduced root)
                w = w[:1] + [w[0], -w[0]] + w[1:]
            row[h * L:h * L + len(w)] = w
        budget = int(rng.choice([40, 700, 6000, 40000]))
        cyc = bool(rng.integers(0, 2))
        world = int(rng.choice([2, 3, 5]))
        bp = int(rng.choice([64, 1024, 1 << 15]))
        try:
            want = O.bfs(row, budget, cyclically_reduce_after_moves=cyc, stats=True)
        except (AssertionError, IndexError):
            want = "raises"

        def run(comm):
            try:
                # (round 6: every level exchanged / the frontier partitioned by owner at the first level of >= 30 or 400 parents / the default;
                # the Python orchestration and acx_bfs_sharded -- the same loop in C++ -- in turn)
                repl = (0, 30, 400, None)[case % 4]
                if (case // 4 + SEED) % 2:
                    from ac_solver.search.sharded import NativeComm, bfs_sharded_native

                    nat = NativeComm.from_python(comm)
                    ok, path, st = bfs_sharded_native(row, budget, cyclically_reduce_after_moves=cyc, comm=nat, batch_parents=bp, want_stats=True, replicate_below=repl)
                    assert not nat.errors, nat.errors[:1]
                    return ok, path, dict(st, owner_mismatches=0)
                return bfs_sharded(row, budget, cyclically_reduce_after_moves=cyc, comm=comm, batch_parents=bp, want_stats=True, replicate_below=repl)
            except (AssertionError, IndexError):
                return "raises"

        got = run_threads(world, run)
        if want == "raises":
            raised += 1
            assert all(g == "raises" for g in got), (L, case, budget, cyc, world, row.tolist())
            continue
        wok, wpath, wst = want
        for ok, path, st in got:
            assert (ok, path) == (wok, wpath), (L, case, budget, cyc, world, bp, row.tolist())
            assert st["nodes"] == wst["nodes"] and st["expanded"] == wst["expanded"], (L, case, budget, cyc, world, bp, row.tolist(), st, wst)
            assert st["owner_mismatches"] == 0
        crossed += sum(st["local_nodes"] > 0 for _, _, st in got) > 1
    assert crossed >= 3  # (the searches really were spread over several ranks)
