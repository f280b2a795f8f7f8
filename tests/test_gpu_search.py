"""GPU parity of bfs / greedy_search (device frontier through acx_search) against the reference's
golden results and the oracle: identical (solved, path), budget-exhaustion returns included."""
import os

import numpy as np
import pytest

from tests.conftest import PKG, ROOT, ms_pool_generator_order

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def search():
    import ac_solver
    from ac_solver import _acx

    _acx.require_device()
    return ac_solver


def _as_tuples(path):
    return None if path is None else [tuple(x) for x in path]


def test_reference_own_golden_paths(search, golden_json):
    """tests/search/test_bfs.py + test_gs.py of the reference: AK(2)"""
    ak2 = np.array([1, 1, -2, -2, -2, 0, 0, 1, 2, 1, -2, -1, -2, 0])
    rows = {(r["algo"], r["budget"], r["cyclical"]): r for r in golden_json("search.json") if r["tag"] == "ak2"}
    ok, path = search.bfs(presentation=ak2, max_nodes_to_explore=int(1e6))
    assert ok and path == _as_tuples(rows[("bfs", 10**6, False)]["path"]) and len(path) == 20
    assert search.bfs(presentation=ak2, max_nodes_to_explore=10) == (False, None)
    ok, path = search.greedy_search(presentation=ak2, max_nodes_to_explore=int(1e6))
    assert ok and path == _as_tuples(rows[("greedy", 10**6, False)]["path"]) and len(path) == 23
    assert search.bfs.__name__ == "bfs" and search.greedy_search.__name__ == "greedy_search"


def test_all_search_fixtures(search, golden_json):
    for r in golden_json("search.json"):
        fn = search.bfs if r["algo"] == "bfs" else search.greedy_search
        ok, path = fn(r["presentation"], r["budget"], cyclically_reduce_after_moves=r["cyclical"])
        assert ok == r["solved"] and path == _as_tuples(r["path"]), (r["tag"], r["algo"], r["budget"], r["cyclical"])


def test_invalid_input(search):
    with pytest.raises(AssertionError):
        search.bfs([1, 0, 2, 0, 0, 0], 10)  # interior / empty relator: not a valid presentation


@pytest.mark.parametrize("algo", ["bfs", "greedy"])
def test_ms_pool_at_L25_vs_oracle(search, golden_json, algo):
    """BASELINE configs 3/4 shape: MS presentations re-embedded at max_relator_length = 25"""
    from oracle import ac_oracle as O

    pool = ms_pool_generator_order(golden_json("ms_pool.json"))
    rng = np.random.default_rng(3)
    for k in rng.choice(len(pool), size=24, replace=False):
        p = np.array(pool[k])
        half = len(p) // 2
        row = np.zeros(50, np.int8)
        for h in (0, 1):
            w = p[h * half:(h + 1) * half]
            w = w[w != 0]
            row[h * 25:h * 25 + len(w)] = w
        for budget in (1, 50, 2000, 60000):
            for cyc in (False, True):
                if algo == "bfs":
                    want = O.bfs(row, budget, cyclically_reduce_after_moves=cyc)
                    got = search.bfs(row, budget, cyclically_reduce_after_moves=cyc)
                else:
                    want = O.greedy_search(row, budget, cyclically_reduce_after_moves=cyc)
                    got = search.greedy_search(row, budget, cyclically_reduce_after_moves=cyc)
                assert got == want, (int(k), budget, cyc)


def test_stats_match_oracle_node_counts(search):
    from ac_solver import _acx
    from ac_solver.search._common import run_search
    from oracle import ac_oracle as O

    ak3 = np.zeros(50, np.int8)
    ak3[:7] = [1, 1, 1, -2, -2, -2, -2]
    ak3[25:31] = [1, 2, 1, -2, -1, -2]
    for kind, ofn in ((_acx.SEARCH_BFS, O.bfs), (_acx.SEARCH_GREEDY, O.greedy_search)):
        for budget in (1000, 100000):
            ok, path, st = run_search(kind, ak3, budget, False)
            wok, wpath, wst = ofn(ak3, budget, stats=True)
            assert (ok, path) == (wok, wpath)
            assert st["nodes"] == wst["nodes"] and st["expanded"] == wst["expanded"], (kind, budget, st, wst)


def test_bfs_budget_sweep_around_batch_and_tile_edges(search):
    """Every small budget, and budgets around the tile size of the one-pass commit (2048 candidates) and the batch sizes:
    node and expansion counts, result and path equal the oracle's (the budget cut is taken from the written nodes)."""
    from ac_solver import _acx
    from ac_solver.search._common import run_search
    from oracle import ac_oracle as O

    ak3 = np.zeros(50, np.int8)
    ak3[:7] = [1, 1, 1, -2, -2, -2, -2]
    ak3[25:31] = [1, 2, 1, -2, -1, -2]
    ak2 = np.array([1, 1, -2, -2, -2, 0, 0, 1, 2, 1, -2, -1, -2, 0], np.int8)
    budgets = list(range(1, 140)) + list(range(2040, 2058)) + list(range(4090, 4104)) + list(range(12280, 12300, 3)) + [24575, 24576, 24577, 99999]
    ak3w = np.zeros(72, np.int8)  # max_relator_length 36: 128-bit keys
    ak3w[:7] = ak3[:7]
    ak3w[36:42] = ak3[25:31]
    for pres, cyc in ((ak3, False), (ak3, True), (ak2, False), (ak3w, False)):
        for budget in budgets:
            ok, path, st = run_search(_acx.SEARCH_BFS, pres, budget, cyc)
            wok, wpath, wst = O.bfs(pres, budget, cyclically_reduce_after_moves=cyc, stats=True)
            assert (ok, path) == (wok, wpath), (cyc, budget)
            assert st["nodes"] == wst["nodes"] and st["expanded"] == wst["expanded"], (cyc, budget, st, wst)


@pytest.mark.timeout(300)
@pytest.mark.parametrize("algo", ["greedy", "bfs"])
def test_full_size_config3_ak3_budget_1e7_against_oracle(search, algo):
    """BASELINE config 3 at its full size: AK(3) at max_relator_length = 25, 1e7-node budget -- identical (solved, path)
    and identical node / expansion counts as the CPU oracle (which needs a few seconds for it)."""
    from ac_solver import _acx
    from ac_solver.search._common import run_search
    from oracle import ac_oracle as O

    ak3 = np.zeros(50, np.int8)
    ak3[:7] = [1, 1, 1, -2, -2, -2, -2]
    ak3[25:31] = [1, 2, 1, -2, -1, -2]
    kind, ofn = (_acx.SEARCH_GREEDY, O.greedy_search) if algo == "greedy" else (_acx.SEARCH_BFS, O.bfs)
    ok, path, st = run_search(kind, ak3, 10**7, False)
    wok, wpath, wst = ofn(ak3, 10**7, stats=True)
    assert (ok, path) == (wok, wpath) and not ok
    assert st["nodes"] == wst["nodes"] and st["expanded"] == wst["expanded"], (st, wst)


@pytest.mark.timeout(600)
def test_bfs_run_ahead_batches_equal_read_back_batches_and_the_oracle(search, golden_json):
    """Round 3: once the frontier holds a full batch the fused bfs enqueues its batches back to back against a device-resident
    cursor and reads it two batches late (acx_frontier.h: BfsCursor); the option ACX_OPT_BFS_NO_RUNAHEAD reads every batch's decision
    back, as small frontiers and verbose searches do.  Both must give the oracle's (solved, path) and counts: searches that END inside the run-ahead phase by success
    (Miller-Schupp presentations bfs solves late), by budget (at several offsets inside a batch) and by exhaustion of nothing
    (budget far beyond the last saturated batch), both cyclical values, both key widths."""
    from ac_solver import _acx
    from ac_solver.search._common import run_search
    from oracle import ac_oracle as O

    g = golden_json("ms_pool.json")
    pool = ms_pool_generator_order(g)
    ak3 = np.zeros(50, np.int8)
    ak3[:7] = [1, 1, 1, -2, -2, -2, -2]
    ak3[25:31] = [1, 2, 1, -2, -1, -2]
    # the Miller-Schupp presentations that bfs (cyclical, budget 1e6) solves LAST: 0.66-0.96e6 nodes with 0.4-0.5e6 of them queued,
    # i.e. well inside the run-ahead phase (a batch is 250 000 parents at this budget); 900, 630 and 730 have 128-bit keys
    late = [pool[k] for k in (325, 334, 494, 900, 49, 185, 235, 630, 25, 730)]
    cases = [(p, 10**6, True) for p in late] + [(p, 10**6, False) for p in late[:4]]
    cases += [(ak3, b, False) for b in (2 * 10**6, 2 * 10**6 + 1, 2999999, 3 * 10**6, 4194304, 4194305)] + [(ak3, 3 * 10**6, True)]
    n_solved = 0
    for p, budget, cyc in cases:
        ahead = run_search(_acx.SEARCH_BFS, p, budget, cyc)
        with _acx.options(OPT_BFS_NO_RUNAHEAD=1):
            back = run_search(_acx.SEARCH_BFS, p, budget, cyc)
        wok, wpath, wst = O.bfs(p, budget, cyclically_reduce_after_moves=cyc, stats=True)
        for ok, path, st in (ahead, back):
            assert (ok, path) == (wok, wpath), (budget, cyc)
            assert st["nodes"] == wst["nodes"] and st["expanded"] == wst["expanded"], (budget, cyc, st, wst)
        assert ahead[2]["levels"] == back[2]["levels"]  # the same batches
        n_solved += ok
    assert n_solved >= 10


def test_greedy_batch_per_launch_path_equals_device_frontier(search, golden_json):
    """greedy_search runs on the persistent one-workgroup frontier (acx_greedy.h); the batch-per-launch path it falls
    back to when a capacity is exceeded -- and that verbose searches take -- (the option ACX_OPT_GREEDY_HOST forces it) must
    return the same thing."""
    from ac_solver import _acx
    from ac_solver.search._common import run_search

    pool = ms_pool_generator_order(golden_json("ms_pool.json"))
    ak3 = np.zeros(50, np.int8)
    ak3[:7] = [1, 1, 1, -2, -2, -2, -2]
    ak3[25:31] = [1, 2, 1, -2, -1, -2]
    cases = [(ak3, 1, False), (ak3, 2, False), (ak3, 13, True), (ak3, 30000, False), (pool[1100], 20000, False), (pool[77], 5000, True), (pool[600], 10**5, False)]
    for p, budget, cyc in cases:
        a = run_search(_acx.SEARCH_GREEDY, p, budget, cyc)
        with _acx.options(OPT_GREEDY_HOST=1):
            b = run_search(_acx.SEARCH_GREEDY, p, budget, cyc)
        assert a[:2] == b[:2] and a[2]["nodes"] == b[2]["nodes"] and a[2]["expanded"] == b[2]["expanded"], (budget, cyc)
        assert a[2]["min_len"] >= b[2]["min_len"]  # the batch path also counts children of parents it speculated on


@pytest.mark.parametrize("hand_min", [1, 6, 100])
def test_greedy_whole_gpu_batches_equal_the_reference(search, golden_json, hand_min):
    """A single greedy_search hands buckets of >= 512 parents to the whole-GPU kernels (acx_greedy_mega.h).  With the
    threshold lowered (ACX_OPT_GREEDY_HAND_MIN) every bucket of the fixture searches takes that route: reference-generated
    paths (all widths incl. 128-bit words, both `cyclical`, solved / budget / raising rows) and the oracle's node counts."""
    from ac_solver import _acx
    from ac_solver.search._common import run_search
    from oracle import ac_oracle as O

    cap = 10**5 if hand_min == 1 else 10**6
    n = 0
    with _acx.options(OPT_GREEDY_HAND_MIN=hand_min):
        for r in golden_json("search.json"):
            if r["algo"] != "greedy" or r["budget"] > cap:
                continue
            ok, path = search.greedy_search(r["presentation"], r["budget"], cyclically_reduce_after_moves=r["cyclical"])
            assert ok == r["solved"] and path == _as_tuples(r["path"]), (r["tag"], r["budget"], r["cyclical"])
            n += 1
        assert n > 100
        pool = ms_pool_generator_order(golden_json("ms_pool.json"))
        rng = np.random.default_rng(hand_min)
        for k in rng.choice(len(pool), size=6, replace=False):
            for budget, cyc in ((40000, False), (7000, True)):
                got = run_search(_acx.SEARCH_GREEDY, np.array(pool[k], dtype=np.int8), budget, cyc)
                wok, wpath, wst = O.greedy_search(pool[k], budget, cyclically_reduce_after_moves=cyc, stats=True)
                assert got[:2] == (wok, wpath), (int(k), budget, cyc)
                assert got[2]["nodes"] == wst["nodes"] and got[2]["expanded"] == wst["expanded"], (int(k), budget, cyc)


@pytest.mark.timeout(600)
def test_greedy_buckets_larger_than_one_whole_gpu_batch(search):
    """AK(3) with a 3e7-node budget meets buckets of 36 000 parents: more than the 16 384 of one mega-batch (acx_greedy_mega.h),
    so a bucket is worked off in several of them; and the cyclically reducing search at 1e7.  Node for node as the oracle."""
    from ac_solver import _acx
    from ac_solver.search._common import run_search
    from oracle import ac_oracle as O

    ak3 = np.zeros(50, np.int8)
    ak3[:7] = [1, 1, 1, -2, -2, -2, -2]
    ak3[25:31] = [1, 2, 1, -2, -1, -2]
    for budget, cyc in ((3 * 10**7, False), (10**7, True)):
        ok, path, st = run_search(_acx.SEARCH_GREEDY, ak3, budget, cyc)
        wok, wpath, wst = O.greedy_search(ak3, budget, cyclically_reduce_after_moves=cyc, stats=True)
        assert (ok, path) == (wok, wpath), (budget, cyc)
        assert st["nodes"] == wst["nodes"] and st["expanded"] == wst["expanded"], (budget, cyc, st, wst)


@pytest.mark.parametrize("env", [{}, {"OPT_MEGA_RANK_MAX": 256}, {"OPT_MEGA_RANK_MAX": 256, "OPT_GREEDY_HAND_MIN": 64}, {"OPT_GREEDY_HAND_MIN": 0},
                                 {"OPT_MEGA_RANK_MAX": 700, "OPT_GREEDY_HAND_MIN": 2048}])
def test_greedy_hand_off_protocols_agree_with_the_oracle(search, env):
    """The hand-off cycle of a single greedy_search, chained on the stream (every kernel reads its work from device scalars, the host
    looks two cycles late): the handed-off bucket ordered by the whole-GPU counting sort, or -- above ACX_OPT_MEGA_RANK_MAX -- by the
    frontier kernel itself before the hand-off; hand-offs from 64 / 512 / 2048 queued parents, or never (the one-workgroup frontier
    alone).  AK(3), 64- and 128-bit keys: result, path and counts as the oracle's."""
    from ac_solver import _acx
    from ac_solver.search._common import run_search
    from oracle import ac_oracle as O

    ak3 = np.zeros(50, np.int8)
    ak3[:7] = [1, 1, 1, -2, -2, -2, -2]
    ak3[25:31] = [1, 2, 1, -2, -1, -2]
    wide = np.zeros(72, np.int8)
    wide[:7] = ak3[:7]
    wide[36:42] = ak3[25:31]
    for pres, budget, cyc in ((ak3, 10**6, False), (ak3, 3 * 10**5, True), (wide, 3 * 10**5, False)):
        with _acx.options(**env):
            ok, path, st = run_search(_acx.SEARCH_GREEDY, pres, budget, cyc)
        wok, wpath, wst = O.greedy_search(pres, budget, cyclically_reduce_after_moves=cyc, stats=True)
        assert (ok, path) == (wok, wpath), (env, budget, cyc)
        assert st["nodes"] == wst["nodes"] and st["expanded"] == wst["expanded"], (env, budget, cyc, st, wst)


def test_greedy_paths_file_sample(search, golden_json):
    """data/greedy_search_paths.txt (budget 1e6): a sample through the device frontier at native L (up to 36 -> 128-bit keys)"""
    pool = ms_pool_generator_order(golden_json("ms_pool.json"))
    g = golden_json("greedy_paths_1e6.json")
    for row in g["rows"][::9]:
        ok, path = search.greedy_search(pool[row["pool_index"]], g["budget"])
        assert ok and path == _as_tuples(row["path"]), row["pool_index"]


# ---- sharded frontier: several ranks of the HIP engine on one GPU (threads stand in for processes) ----
@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_sharded_bfs_hip_engine_matches_reference(search, golden_json, world):
    from ac_solver.search.sharded import SingleComm, bfs_sharded
    from oracle import ac_oracle as O
    from tests.shard_helpers import run_threads

    ak2 = [1, 1, -2, -2, -2, 0, 0, 1, 2, 1, -2, -1, -2, 0]
    ak3 = np.zeros(50, np.int8)
    ak3[:7] = [1, 1, 1, -2, -2, -2, -2]
    ak3[25:31] = [1, 2, 1, -2, -1, -2]
    pool = ms_pool_generator_order(golden_json("ms_pool.json"))
    cases = [(ak2, 10**6, False, 1 << 18), (ak2, 10, False, 1 << 18), (ak2, 1, False, 4), (ak2, 3000, True, 100),
             (ak3, 10**5, False, 1 << 14), (ak3, 5000, True, 333), (pool[1100], 20000, False, 1 << 12), (pool[600], 3000, False, 50)]

    def run(comm):
        # (the three schedules of the side stream in turn: whatever the world size defaults to, every rank count sees all of them)
        # replicate_below (round 6): every level exchanged (0), the frontier partitioned by owner after a few small levels (20, 700),
        # the default (2^18: these searches stay replicated to their end -- no collective but the closing all-reduce)
        out = []
        for repl in (0, 20, 700, None):
            out.append([bfs_sharded(p, b, cyclically_reduce_after_moves=c, comm=comm, batch_parents=bp, want_stats=True, overlap=("insert", "commit", False, None)[k % 4],
                                    replicate_below=repl) for k, (p, b, c, bp) in enumerate(cases)])
        return out

    results = [run(SingleComm())] if world == 1 else run_threads(world, run)
    want = [O.bfs(p, b, cyclically_reduce_after_moves=c, stats=True) for p, b, c, bp in cases]
    for per_repl in results:
        for repl, res in zip((0, 20, 700, None), per_repl):
            for (p, b, c, bp), (ok, path, st), (wok, wpath, wst) in zip(cases, res, want):
                assert (ok, path) == (wok, wpath), (world, b, c, repl)
                assert st["nodes"] == wst["nodes"] and st["expanded"] == wst["expanded"], (world, b, c, repl, st, wst)
                if world > 1 and repl is None:
                    assert st["replicated_levels"] == st["levels"], (world, repl, st)  # no level was exchanged
            if world > 1 and repl in (20, 700):  # some search was partitioned in mid-flight
                assert any(0 < st["replicated_levels"] < st["levels"] for _, _, st in res), (world, repl)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [4, 8])
def test_sharded_bfs_mid_size_thread_ranks_equal_the_fused_search(search, golden_json, world, monkeypatch):
    """Several ranks of the HIP engine on one GPU at sizes where every rank runs many full workgroups per chunk, several chunks per
    level and the two-stream pipeline: 3e6 nodes on AK(3) with 64-bit keys, 1e6 on a Miller-Schupp presentation with 128-bit keys
    (max_relator_length 36), 2e6 with cyclic reduction.  Every rank must return the fused search's node and expansion counts
    (itself checked against the oracle at 1e7, test_full_size_config3...) and the same (solved, path)."""
    from ac_solver import _acx
    from ac_solver.search import sharded
    from ac_solver.search._common import run_search
    from ac_solver.search.sharded import bfs_sharded
    from tests.shard_helpers import run_threads

    monkeypatch.setattr(sharded, "_CHECK_OWNERS", True)  # every node must live on the rank the owner function names (csrc/acx_owner.h)
    ak3 = np.zeros(50, np.int8)
    ak3[:7] = [1, 1, 1, -2, -2, -2, -2]
    ak3[25:31] = [1, 2, 1, -2, -1, -2]
    pool = ms_pool_generator_order(golden_json("ms_pool.json"))
    cases = [(ak3, 3 * 10**6, False, 1 << 16), (np.array(pool[1100], np.int8), 10**6, False, 1 << 15), (ak3, 2 * 10**6, True, 1 << 17),
             (np.array(pool[630], np.int8), 10**6, True, 1 << 14)]  # (630: solved after 0.78e6 nodes with cyclic reduction)
    want = [run_search(_acx.SEARCH_BFS, p, b, c) for p, b, c, _ in cases]

    def run(comm):
        # every level exchanged, then the small levels replicated (partition at the first level of >= 2^12 / the default 2^18 parents)
        return [[bfs_sharded(p, b, cyclically_reduce_after_moves=c, comm=comm, batch_parents=bp, want_stats=True, replicate_below=repl) for p, b, c, bp in cases]
                for repl in (0, 1 << 12, None)]

    all_res = run_threads(world, run)
    for per_repl in all_res:
        for repl, res in zip((0, 1 << 12, None), per_repl):
            for (ok, path, st), (wok, wpath, wst) in zip(res, want):
                assert (ok, path) == (wok, wpath), (world, repl)
                assert st["nodes"] == wst["nodes"] and st["expanded"] == wst["expanded"], (world, repl, st, wst)
                assert st["owner_mismatches"] == 0, (world, repl, st)
                if repl:
                    assert 0 < st["replicated_levels"] < st["levels"], (world, repl, st)
    all_res = [per_repl[0] for per_repl in all_res]  # (what follows looks at the searches with every level exchanged)
    res = all_res[-1]
    # the ranks' shares of the nodes: the owner function spreads them (1.03-1.10 x the mean at 8 ranks on these searches)
    for k in range(len(cases)):
        shares = [res[k][2]["local_nodes"] for res in all_res]
        assert max(shares) * world <= 1.35 * sum(shares), (world, k, shares)
    assert want[3][0] and not want[0][0]
    if world == 4:
        # the regions of the exchange: adaptive capacity really in use (chunks of 2^16 parents: an even share of 3072 records per
        # sub-region), and a capacity that cannot hold overflows, fails every rank at the same chunk and is rerun with the
        # default -- same counts
        fills = res[0][2]["region_fill_q8"]
        assert isinstance(fills, list) and fills and min(fills) < 320, fills

        # (only the few per cent of the children that another rank owns are sent, csrc/acx_owner.h: it takes a level of ~2.7e6 parents in
        # ONE chunk to push ~7e3 records at a sub-region of 3072 + 1/256 of the even share)
        big = run_search(_acx.SEARCH_BFS, ak3, 10**7, False)

        def forced(comm):
            return bfs_sharded(ak3, 10**7, comm=comm, batch_parents=1 << 22, want_stats=True, region_fill=1)

        for ok, path, st in run_threads(world, forced):
            assert (ok, path) == big[:2] and st["nodes"] == big[2]["nodes"] and st["expanded"] == big[2]["expanded"]
            assert st.get("region_overflow_reruns") == 1, st
        # ... and when the default capacity overflows as well (here: because the test shrinks it), the third attempt runs under the
        # hard bound (every workgroup could send a region all it has) with small chunks
        monkeypatch.setattr(sharded, "FILL_DEFAULT", 2)
        for ok, path, st in run_threads(world, forced):
            assert (ok, path) == big[:2] and st["nodes"] == big[2]["nodes"] and st["expanded"] == big[2]["expanded"]
            assert st.get("region_overflow_reruns") == 2, st


@pytest.mark.timeout(600)
@pytest.mark.parametrize("where", ["chunk_expand", "chunk_insert", "chunk_commit"])
def test_a_host_side_failure_of_the_hip_engine_ends_every_thread_rank_together(search, where):
    """An engine call of ONE rank raises on the host side (before the C call: what an exhausted allocation or a HIP error looks like
    to the orchestrator) in the middle of a level of many chunks, with the control block read two chunks late and the expansion
    running ahead on the side stream.  Every rank must issue the same collectives and raise; a rank left alone in a collective
    would hang the thread barrier (and this test).  The engine of the failing rank keeps its ring of open chunks in step
    (acx_shard_chunk_insert_dead), so a second search on fresh engines afterwards is exact."""
    from ac_solver.search import sharded
    from ac_solver.search.sharded import HipShardEngine, bfs_sharded
    from oracle import ac_oracle as O
    from tests.shard_helpers import run_threads

    ak3 = np.zeros(50, np.int8)
    ak3[:7] = [1, 1, 1, -2, -2, -2, -2]
    ak3[25:31] = [1, 2, 1, -2, -1, -2]
    for fail_at in (2, 9, 23):
        calls = {"n": 0}

        class Flaky(HipShardEngine):
            def _maybe(self, name):
                if name == where and self.rank == 1:
                    calls["n"] += 1
                    if calls["n"] == fail_at:
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
                bfs_sharded(ak3, 200000, comm=comm, engine_factory=Flaky, batch_parents=1 << 10)
            except RuntimeError as e:
                return str(e), dict(comm.stats)
            return "no error", dict(comm.stats)

        out = run_threads(2, run)
        assert calls["n"] >= fail_at, "the search ended before the failing call"
        assert all("sharded bfs failed" in m for m, _ in out) and "simulated" in out[1][0], [m for m, _ in out]
        assert out[0][1] == out[1][1], [st for _, st in out]

    def clean(comm):
        return bfs_sharded(ak3, 200000, comm=comm, batch_parents=1 << 10)

    assert all(r == O.bfs(ak3, 200000) for r in run_threads(2, clean))


def test_search_beyond_the_key_width_says_so(search):
    """the reference searches at any max_relator_length; the device frontier names a relator by one 128-bit key: 64 letters at most, and says so
    beyond (a ValueError naming the limit, not a generic library error); from 62 letters on the root has to be freely reduced"""
    from ac_solver import bfs, greedy_search
    from ac_solver.search.sharded import bfs_sharded, bfs_sharded_native

    p = np.zeros(130, np.int8)
    p[:2], p[65:67] = [1, 2], [2, 1]
    for fn in (bfs, greedy_search, bfs_sharded, bfs_sharded_native):
        with pytest.raises(ValueError, match="max_relator_length = 65"):
            fn(p, 100)
    q = np.zeros(124, np.int8)
    q[:3], q[62:64] = [1, -1, 2], [2, 1]  # x x^-1 y: not freely reduced
    for fn in (bfs, greedy_search, bfs_sharded, bfs_sharded_native):
        with pytest.raises(ValueError, match="freely reduced"):
            fn(q, 100)
    ok = np.zeros(122, np.int8)
    ok[:5], ok[61:67] = [1, 1, -2, -2, -2], [1, 2, 1, -2, -1, -2]  # AK(2) at the widest length of the plain 128-bit key
    assert greedy_search(ok, 10000)[1] is not None and bfs(ok, 1000) == (False, None)


def _ms_presentation(n, w, L):
    """<x, y | x^-1 y^n x = y^(n + 1), x = w> as the reference's generator writes it (miller_schupp.py:20-83) at max_relator_length L"""
    r1 = [-1] + [2] * n + [1] + [-2] * (n + 1)
    r2 = [-1] + list(w)
    p = np.zeros(2 * L, np.int8)
    p[:len(r1)], p[L:L + len(r2)] = r1, r2
    return p


@pytest.mark.timeout(900)
@pytest.mark.parametrize("L", [62, 63, 64])
def test_searches_at_max_relator_length_62_to_64_match_the_oracle(search, L):
    """The reference takes any max_relator_length (breadth_first.py:42-45) and its own Miller-Schupp generator reaches 64 at n = 14
    (miller_schupp.py:43).  Keys of these lengths (csrc/acx_keys.h): bfs, greedy_search, the sharded bfs (Python orchestration and
    acx_bfs_sharded, 1 and 3 thread ranks) and a batch through acx_search_many against the C oracle -- result, path, node and expansion
    counts -- on Miller-Schupp presentations at n = 14 / 13, on AK(3), and on a presentation whose relators reach the full 64 letters."""
    from ac_solver import _acx, bfs, greedy_search
    from ac_solver.search._common import run_search, run_search_many
    from ac_solver.search.sharded import NativeComm, bfs_sharded, bfs_sharded_native
    from oracle import ac_oracle as O
    from tests.shard_helpers import run_threads

    rng = np.random.default_rng(L)
    ak3 = np.zeros(2 * L, np.int8)
    ak3[:7], ak3[L:L + 6] = [1, 1, 1, -2, -2, -2, -2], [1, 2, 1, -2, -1, -2]
    long_rows = np.zeros(2 * L, np.int8)  # two long reduced relators: moves run into the length bound from the start
    for h in (0, 1):
        w = []
        while len(w) < L - 3 * h:
            c = int(rng.choice([-2, -1, 1, 2]))
            if not w or w[-1] != -c:
                w.append(c)
        long_rows[h * L:h * L + len(w)] = w
    cases = [(_ms_presentation(14 if L == 64 else 13, [2, 1, -2], L), 20000, False), (_ms_presentation(14 if L == 64 else 13, [2, 1, 2, -1], L), 6000, True),
             (ak3, 30000, False), (ak3, 5000, True), (long_rows, 3000, False), (long_rows, 2000, True)]
    for p, budget, cyc in cases:
        wb = O.bfs(p, budget, cyclically_reduce_after_moves=cyc, stats=True)
        wg = O.greedy_search(p, budget, cyclically_reduce_after_moves=cyc, stats=True)
        ok, path, st = run_search(_acx.SEARCH_BFS, p, budget, cyc)
        assert (ok, path) == wb[:2] and st["nodes"] == wb[2]["nodes"] and st["expanded"] == wb[2]["expanded"], (L, budget, cyc)
        ok, path, st = run_search(_acx.SEARCH_GREEDY, p, budget, cyc)
        assert (ok, path) == wg[:2] and st["nodes"] == wg[2]["nodes"] and st["expanded"] == wg[2]["expanded"], (L, budget, cyc)
        with _acx.options(OPT_GREEDY_HOST=1):  # the batch-per-launch greedy frontier on the same key type
            ok, path, st = run_search(_acx.SEARCH_GREEDY, p, budget, cyc)
        assert (ok, path) == wg[:2] and st["nodes"] == wg[2]["nodes"] and st["expanded"] == wg[2]["expanded"], (L, budget, cyc)
        assert bfs(p, budget, cyclically_reduce_after_moves=cyc) == (wb[0], wb[1] if wb[0] else None)
        assert greedy_search(p, budget, cyclically_reduce_after_moves=cyc) == wg[:2]
        ok, path, st = bfs_sharded(p, budget, cyclically_reduce_after_moves=cyc, want_stats=True)
        assert (ok, path) == wb[:2] and st["nodes"] == wb[2]["nodes"], (L, budget, cyc)

        def run(comm):
            nat = NativeComm.from_python(comm)
            a = bfs_sharded_native(p, budget, cyclically_reduce_after_moves=cyc, comm=nat, batch_parents=512, replicate_below=40, want_stats=True)
            b = bfs_sharded(p, budget, cyclically_reduce_after_moves=cyc, comm=comm, batch_parents=512, replicate_below=0, want_stats=True)
            assert not nat.errors, nat.errors[:1]
            return a, b

        for a, b in run_threads(3, run):
            for ok, path, st in (a, b):
                assert (ok, path) == wb[:2] and st["nodes"] == wb[2]["nodes"] and st["expanded"] == wb[2]["expanded"], (L, budget, cyc)
    if L == 64:  # the batch driver itself at n = 14 (miller_schupp.py:95-177: max_relator_length 64 there) against the oracle, search by search
        from ac_solver.search.miller_schupp.miller_schupp import generate_miller_schupp_presentations, trivialize_miller_schupp_through_search

        for fn, ofn in ((greedy_search, O.greedy_search), (bfs, O.bfs)):
            s_rels, u_rels, s_paths = trivialize_miller_schupp_through_search(min_n=14, max_n=14, min_w_len=1, max_w_len=3, max_nodes_to_explore=2000, search_fn=fn)
            gen14 = generate_miller_schupp_presentations(14, 3)
            todo = [p for lenw in range(1, 4) for p in gen14.get(lenw, [])]
            assert len(s_rels) + len(u_rels) == len(todo) > 0 and all(len(p) == 128 for p in todo)
            want = [ofn(np.array(p, np.int8), 2000) for p in todo]
            assert [list(p) for p in s_rels] == [list(p) for p, w in zip(todo, want) if w[0]]
            assert s_paths == [w[1] for w in want if w[0]]
    if L == 64:  # the package's own generator at n = 14 (the reference's reaches max_relator_length 64 there) through the single searches
        from ac_solver.search.miller_schupp.miller_schupp import generate_miller_schupp_presentations

        gen = generate_miller_schupp_presentations(14, 3)
        sample = [np.array(p, np.int8) for lenw in sorted(gen) for p in gen[lenw]][:8]
        assert sample and all(len(p) == 128 for p in sample)
        for p in sample:
            assert greedy_search(p, 3000) == O.greedy_search(p, 3000) and bfs(p, 3000) == (lambda r: (r[0], r[1] if r[0] else None))(O.bfs(p, 3000))
    rows = np.stack([c[0] for c in cases[:3]])
    for kind, fn in ((_acx.SEARCH_BFS, O.bfs), (_acx.SEARCH_GREEDY, O.greedy_search)):
        got = run_search_many(kind, rows, 4000, False)
        for r, (ok, path, st) in zip(rows, got):
            assert (ok, path) == fn(r, 4000)[:2], (L, kind)


def _key_words(state, L, KW):
    """the packed key of a state as the engine's int64 words (2-bit letters, length in the top six bits of each relator's word)"""
    code = {-2: 0, -1: 1, 1: 2, 2: 3}
    words = []
    for h in (0, 1):
        w = [int(a) for a in state[h * L:(h + 1) * L] if a != 0]
        k = 0
        for i, a in enumerate(w):
            k |= code[a] << (2 * i)
        if KW == 2:
            k |= len(w) << 58
            words.append(k - (1 << 64) if k >= (1 << 63) else k)
        else:
            k |= len(w) << 122
            for part in (k & ((1 << 64) - 1), k >> 64):
                words.append(part - (1 << 64) if part >= (1 << 63) else part)
    return words


@pytest.mark.timeout(600)
@pytest.mark.parametrize("L,cyclical", [(25, False), (36, False), (25, True)])
def test_device_routing_matches_owner_of(search, L, cyclical):
    """`world` engines on one GPU driven in lockstep by hand (the exchange is a copy per pair of ranks, the mask all-reduce a sum):
    acx_shard_chunk_expand routes every child to the region of the rank the owner function names (sharded.owner_of = csrc/acx_owner.h:
    a function of the conjugacy classes of the two relators and of the letter next to each one's cyclically reduced core), the
    children a rank owns itself are BORN in the expansion kernel (no record) -- in a normal-form search the kernel inherits a
    conjugation child's class hashes from its parent without computing them --, and exactly the children that should travel do: not the unchanged ones, not the ones that undo their
    parent's move (normal-form root, cyclical = False), and of the duplicates inside a workgroup tile (128 consecutive LOCAL parents)
    only the smallest tag.  Insert + commit on every rank reproduce a plain BFS level by level; every node lives on its owner."""
    import torch

    from ac_solver.search.sharded import CTL_NODES, CTL_NEXT_COUNT, HDR, HipShardEngine, owner_of
    from oracle import ac_oracle as O

    inverse = {0: 2, 2: 0, 1: 3, 3: 1, 4: 8, 8: 4, 5: 9, 9: 5, 6: 10, 10: 6, 7: 11, 11: 7}
    ak3 = np.zeros(2 * L, np.int8)
    ak3[:7] = [1, 1, 1, -2, -2, -2, -2]
    ak3[L:L + 6] = [1, 2, 1, -2, -1, -2]
    for world in (2, 3, 8):
        engs = [HipShardEngine(L, cyclical, 100000, 1 << 15, r, world, 50000) for r in range(world)]
        KW, RW = engs[0].KW, engs[0].RW

        def owner(st):
            return int(owner_of(np.array([_key_words(st, L, KW)], np.int64), world)[0])

        roots = [e.root_record(ak3) for e in engs]  # (every rank calls it: it selects the move code)
        o_root = engs[0].root_owner(roots[0])
        assert o_root == owner(ak3) == int(owner_of(roots[0][None, :KW], world)[0])
        for r, e in enumerate(engs):
            e.seed(roots[r] if r == o_root else None)
        level = [(ak3.copy(), 0xff, o_root, 0)]  # host mirror of the frontier in global FIFO order: state, action that made it, owner, local id there
        visited = {tuple(ak3.tolist())}
        n_local = [1 if r == o_root else 0 for r in range(world)]
        n_born = n_sent = 0
        for lvl in range(5 if not cyclical else 6):  # a few levels: 12, then up to 144, ... children
            F = len(level)
            bufs = [e.chunk_expand(0, F, True) for e in engs]
            S, cap, rw = engs[0].layout(F, 0)
            # what every rank should have sent where, and what it should have kept: tile by tile of its LOCAL parents
            want = {}                                    # tag -> (state, owner)
            want_sent = [dict() for _ in range(world)]   # per sender: tag -> (destination, parent's local id)
            local_idx = [0] * world
            tile_keys = {}
            for gp, (st, made_by, po, pid) in enumerate(level):
                out, lens, err = O.move_batch(np.repeat(st[None], 12, axis=0), np.arange(12, dtype=np.uint8), L, cyclical=cyclical)
                tile = (po, local_idx[po] // 128)
                local_idx[po] += 1
                for a in range(12):
                    if np.array_equal(out[a], st) or (not cyclical and made_by < 12 and a == inverse[made_by]):
                        continue
                    key = tuple(out[a].tolist())
                    if key in tile_keys.setdefault(tile, set()):
                        continue
                    tile_keys[tile].add(key)
                    co = owner(out[a])
                    want[12 * gp + a] = (out[a], co)
                    if co != po:
                        want_sent[po][12 * gp + a] = (co, pid)
                    else:
                        n_born += 1
            for r, (send, recv) in enumerate(bufs):
                regs = send.view(S * world, rw).cpu()
                got = {}
                for q in range(S * world):
                    n = int(regs[q, 0])
                    assert n <= cap and int(regs[q, 1]) == 1 << 62 and int(regs[q, 2]) == 1 << 62 and int(regs[q, 3]) == 0
                    rows = regs[q, HDR:HDR + n * RW].view(n, RW)
                    if n:
                        assert (owner_of(rows[:, :KW], world) == q // S).all(), (L, world, r, q)
                    for row in rows.tolist():
                        got[row[KW] >> 32] = (q // S, row[KW] & 0xFFFFFFFF)
                assert got == want_sent[r], (L, world, lvl, r)
                n_sent += len(got)
            # the exchange: region block d of rank r's send buffer -> block r of rank d's receive area
            blk = S * rw
            for d in range(world):
                for r in range(world):
                    bufs[d][1].view(world, blk)[r].copy_(bufs[r][0].view(world, blk)[d])
            masks = [e.chunk_insert(F).clone() for e in engs]
            total = torch.stack(masks).sum(0).to(torch.int32)  # the all-reduce (sum == or: a child has one owner)
            lmasks = []
            for e, m in zip(engs, masks):
                packed = m.to(torch.int64)
                lmasks.append(torch.stack([packed & 0xFFF, (packed >> 16) & 0xFFF], dim=1).reshape(-1)[:F].cpu())
                e.gmask_view(F).copy_(total)
                e.chunk_commit(1 << 40)
            nxt = []
            new_local = [0] * world
            for tag in sorted(want):
                st, co = want[tag]
                key = tuple(st.tolist())
                if key in visited:
                    continue
                visited.add(key)
                nxt.append((st, tag % 12, co, n_local[co] + new_local[co]))
                new_local[co] += 1
                for r in range(world):  # the bit is set on the owner and nowhere else
                    assert ((int(lmasks[r][tag // 12]) >> (tag % 12)) & 1) == (1 if r == co else 0), (tag, r, co)
            for r, e in enumerate(engs):
                e.ctl_snapshot(0)
                ctl = e.ctl_wait(0)
                assert int(ctl[CTL_NEXT_COUNT]) == len(nxt) and int(ctl[CTL_NODES]) == n_local[r] + new_local[r], (L, world, lvl, r)
                assert e.check_owners() == 0
            for k in (0, len(nxt) // 2, len(nxt) - 1):  # committed in tag order: a rank's node ids follow the global FIFO order
                st, a, co, lid = nxt[k]
                got_a, tl, pref = engs[co].node_info(lid)
                assert got_a == a and tl == int(np.count_nonzero(st))
            n_local = [n_local[r] + new_local[r] for r in range(world)]
            level = nxt
        assert n_sent > 0 and n_born > 0  # (a few levels deep the conjugators are short and about half of the children still travel; at depth a quarter)
        for e in engs:
            e.close()


def test_many_searches_overlapped_equal_single(search, golden_json):
    from ac_solver import _acx
    from ac_solver.search._common import run_search, run_search_many

    pool = ms_pool_generator_order(golden_json("ms_pool.json"))
    rows = np.array(pool[340:510], dtype=np.int8)  # the 170 presentations of n = 3 (L = 20)
    for kind in (_acx.SEARCH_GREEDY, _acx.SEARCH_BFS):
        many = run_search_many(kind, rows, 3000, False, n_threads=16)
        for k in range(0, len(rows), 7):
            ok, path, st = run_search(kind, rows[k], 3000, False)
            assert (ok, path) == many[k][:2] and st["nodes"] == many[k][2]["nodes"]


def _pad(rel0, rel1, L):
    row = np.zeros(2 * L, np.int8)
    row[: len(rel0)] = rel0
    row[L: L + len(rel1)] = rel1
    return row


@pytest.mark.parametrize("L", [9, 33])
def test_many_bfs_searches_share_their_launches_and_equal_single_ones(search, L):
    """acx_search_many(bfs) = acx_bfs_many.h: one round of launches advances every search of the group by a batch.  Groups that mix
    roots in and out of normal form (two move codes), searches that end by success, by budget at every small budget and by an empty
    queue, batches of 128 / 1024 / 32768 parents: (solved, path, nodes, expanded) equal the single search's, which the other
    tests hold against the oracle."""
    from ac_solver import _acx
    from ac_solver.search._common import run_search, run_search_many

    rows = np.stack([
        _pad([1, 1, -2, -2, -2], [1, 2, 1, -2, -1, -2], L),          # AK(2): solved after 7e4 nodes
        _pad([1, 1, 1, -2, -2, -2, -2], [1, 2, 1, -2, -1, -2], L),   # AK(3): never solved here
        _pad([1, 2, -2, 1], [2, 1, -1, 2, 2], L),                    # not freely reduced: the general move code
        _pad([2, 1, -2], [1, 2, -1, -1], L),                         # not cyclically reduced
        _pad([1], [2], L),                                           # trivial already: the first child ends the search
        _pad([1, 1], [2, 2], L),
        _pad([1, 2], [2, 1, 1], L),
    ])
    for cyc in (False, True):
        single = {}
        for bmax in (128, 1024, 32768):
            for budget in list(range(0, 30)) + [157, 1537, 1538, 20000, 120000]:
                if bmax != 128 and budget < 30 and budget % 7:
                    continue
                with _acx.options(OPT_BFS_MANY_BMAX=bmax):
                    many = run_search_many(_acx.SEARCH_BFS, rows, budget, cyc)
                for k, (ok, path, st) in enumerate(many):
                    if (k, budget) not in single:
                        single[(k, budget)] = run_search(_acx.SEARCH_BFS, rows[k], budget, cyc)
                    wok, wpath, wst = single[(k, budget)]
                    assert (ok, path) == (wok, wpath), (cyc, bmax, budget, k)
                    # (min_len also counts the children of the last batch behind the one that ended the search: it depends on the batch size)
                    assert [st[f] for f in ("nodes", "expanded", "children")] == [wst[f] for f in ("nodes", "expanded", "children")], (cyc, bmax, budget, k, st, wst)
                    assert st["min_len"] == wst["min_len"] or not ok
    # a finite state space: the queue runs empty (breadth_first.py:61) in every search of the group, after different numbers of batches
    tiny = np.stack([_pad([1, 1], [2, 2], 3), _pad([1, 1], [2, 1], 3), _pad([1, 1], [2], 3), _pad([1, 2, 1], [2, 2], 3), _pad([2, 2], [1, -2], 3)])
    with _acx.options(OPT_BFS_MANY_BMAX=128):
        got = run_search_many(_acx.SEARCH_BFS, tiny, 10**5, False)
    assert [(ok, path, st["nodes"], st["expanded"]) for ok, path, st in got] == [(False, None, n, n) for n in (1, 48, 108, 60, 48)]  # (the C oracle's counts)
    for k, (ok, path, st) in enumerate(got):
        wok, wpath, wst = run_search(_acx.SEARCH_BFS, tiny[k], 10**5, False)
        assert (ok, path, st["nodes"], st["expanded"]) == (wok, wpath, wst["nodes"], wst["expanded"])
    # a move on which the reference's ACMove raises, in ONE search of a group: the call raises as the single search does
    bad = np.stack([_pad([1, 1], [2, 2], 2), _pad([1, 2], [2, 1], 2)])
    with pytest.raises(AssertionError):
        run_search(_acx.SEARCH_BFS, bad[1], 100, False)
    with pytest.raises(AssertionError):
        run_search_many(_acx.SEARCH_BFS, bad, 100, False)


@pytest.fixture
def greedy_slots(request):
    from ac_solver import _acx

    with _acx.options(OPT_GREEDY_SLOTS=request.param):
        yield request.param


@pytest.mark.parametrize("greedy_slots", [2, 5, 512], indirect=True)
def test_greedy_searches_as_jobs_on_a_few_workgroup_slots(search, golden_json, greedy_slots):
    """acx_search_groups / acx_search_many(greedy) = k_greedy_sched: a fixed set of workgroups, each with the memory of ONE search,
    takes the searches from a counter and cleans its slot (visited table, bucket rows, the vector L1) between two of them.  With 2 or 5
    slots every workgroup runs dozens of searches one after the other -- of different max_relator_length, solved after a handful
    of nodes or cut off by the budget at depth > 100, 64- and 128-bit keys, roots in and out of normal form: each (solved, path,
    nodes, expanded) as the single search's, which the other tests hold against the oracle."""
    from ac_solver import _acx
    from ac_solver.search._common import run_search, run_search_groups, run_search_many

    slots = greedy_slots
    pool = ms_pool_generator_order(golden_json("ms_pool.json"))
    rng = np.random.default_rng(17)
    groups = []
    for n0 in (0, 2, 4, 5, 6):  # max_relator_length 18, 20, 28 (64-bit keys), 32, 36 (128-bit keys)
        pick = np.sort(rng.choice(170, size=14, replace=False)) + 170 * n0
        groups.append(np.array([pool[int(k)] for k in pick], dtype=np.int8))
    extra = np.stack([_pad([1, 2, -2, 1], [2, 1, -1, 2, 2], 12), _pad([2, 1, -2], [1, 2, -1, -1], 12), _pad([1], [2], 12), _pad([1, 1, 1, -2, -2, -2, -2], [1, 2, 1, -2, -1, -2], 12)])
    groups.append(extra)  # not in normal form: the general move code, a launch of its own
    for budget, cyc in ((3000, False), (40000, True)):
        got = run_search_groups(_acx.SEARCH_GREEDY, groups, budget, cyc)
        assert [len(r) for r in got] == [len(g) for g in groups]
        for g, res in zip(groups, got):
            for row, (ok, path, st) in zip(g, res):
                wok, wpath, wst = run_search(_acx.SEARCH_GREEDY, row, budget, cyc)
                assert (ok, path) == (wok, wpath), (slots, budget, cyc, row.tolist())
                assert st["nodes"] == wst["nodes"] and st["expanded"] == wst["expanded"], (slots, budget, cyc, row.tolist(), st, wst)
        one = run_search_many(_acx.SEARCH_GREEDY, groups[1], budget, cyc)  # a single batch takes the same route
        assert [(ok, path, st["nodes"]) for ok, path, st in one] == [(ok, path, st["nodes"]) for ok, path, st in got[1]]
    # bfs through the same entry: the batches one after the other
    got = run_search_groups(_acx.SEARCH_BFS, groups[:3], 2000, True)
    for g, res in zip(groups[:3], got):
        for row, (ok, path, st) in zip(g, res):
            wok, wpath, wst = run_search(_acx.SEARCH_BFS, row, 2000, True)
            assert (ok, path, st["nodes"], st["expanded"]) == (wok, wpath, wst["nodes"], wst["expanded"])
    # a search in which the reference raises makes the call raise
    bad = np.stack([_pad([1, 1], [2, 2], 2), _pad([1, 2], [2, 1], 2)])
    with pytest.raises(AssertionError):
        run_search_groups(_acx.SEARCH_GREEDY, [groups[0], bad], 100, False)


def test_greedy_slot_blocks_come_back_clean(search, golden_json):
    """The blocks that hold the slots of a call go back to the library's block pool tagged "every slot clean" (every workgroup hands its
    slot back that way), and the next call with the same layout fills nothing: the same groups three times over, another budget (another
    layout: the tag must not match) in between -- the same results every time, and as the single search's."""
    from ac_solver import _acx
    from ac_solver.search._common import run_search, run_search_groups

    pool = ms_pool_generator_order(golden_json("ms_pool.json"))
    rng = np.random.default_rng(23)
    groups = []
    for n0 in (1, 3, 6):  # max_relator_length 18, 24 (64-bit keys), 36 (128-bit keys)
        pick = np.sort(rng.choice(170, size=24, replace=False)) + 170 * n0
        groups.append(np.array([pool[int(k)] for k in pick], dtype=np.int8))

    def key(res):
        return [[(ok, path, st["nodes"], st["expanded"]) for ok, path, st in r] for r in res]

    first = key(run_search_groups(_acx.SEARCH_GREEDY, groups, 20000, False))
    other = key(run_search_groups(_acx.SEARCH_GREEDY, groups, 5000, False))
    assert key(run_search_groups(_acx.SEARCH_GREEDY, groups, 20000, False)) == first
    assert key(run_search_groups(_acx.SEARCH_GREEDY, groups[::-1], 20000, False)) == first[::-1]
    assert key(run_search_groups(_acx.SEARCH_GREEDY, groups, 5000, False)) == other
    for g, res in zip(groups, first):
        for row, got in zip(g[::5], res[::5]):
            wok, wpath, wst = run_search(_acx.SEARCH_GREEDY, row, 20000, False)
            assert got == (wok, wpath, wst["nodes"], wst["expanded"])


def test_greedy_jobs_borrow_the_shared_sort_scratch(search, golden_json):
    """k_greedy_sched: a slot's own sort scratch is small, a bucket that outgrows it is ordered in one of a few full-size regions that all
    slots of the call share (GreedyDev::big_lock).  With the smallest scratch (2048 entries) and 24 searches on 12 slots whose buckets
    reach thousands of entries the regions are contended: every search as the single search's."""
    from ac_solver import _acx
    from ac_solver.search._common import run_search, run_search_many

    pool = ms_pool_generator_order(golden_json("ms_pool.json"))
    rows = [np.asarray(pool[k], dtype=np.int8) for k in range(170, 170 + 340, 15)]  # max_relator_length 18 / 20: the tightest bounds, the largest buckets
    budget = 150000
    with _acx.options(OPT_GREEDY_SLOTS=12, OPT_GREEDY_SCRATCH=2048):
        got = {}
        for L in sorted({len(r) // 2 for r in rows}):
            grp = np.stack([r for r in rows if len(r) // 2 == L])
            for row, res in zip(grp, run_search_many(_acx.SEARCH_GREEDY, grp, budget, False)):
                got[row.tobytes()] = res
    for row in rows:
        ok, path, st = got[row.tobytes()]
        wok, wpath, wst = run_search(_acx.SEARCH_GREEDY, row, budget, False)
        assert (ok, path, st["nodes"], st["expanded"]) == (wok, wpath, wst["nodes"], wst["expanded"]), row.tolist()


def test_miller_schupp_driver_matches_reference_test_ranges(search, golden_json):
    """tests/search/miller_schupp/test_miller_schupp.py of the reference: n, w in {1, 2} (greedy 1e6, bfs 1e4) and {3, 4} (greedy 1e4)"""
    from ac_solver.search.miller_schupp.miller_schupp import trivialize_miller_schupp_through_search

    rows = golden_json("search.json")

    def expected(algo, budget, tags):
        sel = [r for r in rows if r["algo"] == algo and r["budget"] == budget and r["tag"] in tags and not r["cyclical"]]
        solved = [r["presentation"] for r in sel if r["solved"]]
        unsolved = [r["presentation"] for r in sel if not r["solved"]]
        paths = [_as_tuples(r["path"]) for r in sel if r["solved"]]
        return solved, unsolved, paths

    small = [f"ms_n{n}_w{w}" for n in (1, 2) for w in (1, 2)]
    big = [f"ms_n{n}_w{w}" for n in (3, 4) for w in (3, 4)]
    for algo, fn, budget, tags, rng in (("greedy", search.greedy_search, 10**6, small, (1, 2, 1, 2)), ("bfs", search.bfs, 10**4, small, (1, 2, 1, 2)),
                                        ("greedy", search.greedy_search, 10**4, big, (3, 4, 3, 4))):
        s, u, p = trivialize_miller_schupp_through_search(min_n=rng[0], max_n=rng[1], min_w_len=rng[2], max_w_len=rng[3],
                                                          max_nodes_to_explore=budget, search_fn=fn)
        ws, wu, wp = expected(algo, budget, tags)
        assert [list(x) for x in s] == ws and [list(x) for x in u] == wu and p == wp


# ---- the whole sharded search as ONE C call per rank (acx_bfs_sharded, csrc/acx_shard_run.hip) ----
@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_native_sharded_bfs_equals_the_reference_and_the_python_orchestrator(search, golden_json, world):
    """acx_bfs_sharded on thread ranks of one GPU (the collectives through ctypes callbacks into the tests' ThreadComm): result, path
    and counts of the oracle (breadth_first.py:61-95), and the same levels / chunks / replicated levels as sharded.py's orchestration of
    the same engine -- every level exchanged, the frontier partitioned after a few small levels, the default."""
    from ac_solver.search.sharded import NativeComm, SingleComm, bfs_sharded, bfs_sharded_native
    from oracle import ac_oracle as O
    from tests.shard_helpers import run_threads

    ak2 = [1, 1, -2, -2, -2, 0, 0, 1, 2, 1, -2, -1, -2, 0]
    ak3 = np.zeros(50, np.int8)
    ak3[:7] = [1, 1, 1, -2, -2, -2, -2]
    ak3[25:31] = [1, 2, 1, -2, -1, -2]
    pool = ms_pool_generator_order(golden_json("ms_pool.json"))
    cases = [(ak2, 10**6, False, 1 << 18), (ak2, 10, False, 1 << 18), (ak2, 1, False, 4), (ak2, 3000, True, 100),
             (ak3, 10**5, False, 1 << 14), (ak3, 5000, True, 333), (pool[1100], 20000, False, 1 << 12), (pool[600], 3000, False, 50)]
    repls = (0, 20, 700, None)

    def run(comm):
        nat = None if world == 1 else NativeComm.from_python(comm)
        out = []
        for repl in repls:
            for p, b, c, bp in cases:
                a = bfs_sharded_native(p, b, cyclically_reduce_after_moves=c, comm=nat, batch_parents=bp, want_stats=True, replicate_below=repl)
                ref = bfs_sharded(p, b, cyclically_reduce_after_moves=c, comm=comm, batch_parents=bp, want_stats=True, replicate_below=repl)
                out.append((a, ref))
        assert nat is None or not nat.errors, nat.errors
        return out

    results = [run(SingleComm())] if world == 1 else run_threads(world, run)
    want = [O.bfs(p, b, cyclically_reduce_after_moves=c, stats=True) for p, b, c, bp in cases]
    for res in results:
        for k, ((ok, path, st), (rok, rpath, rst)) in enumerate(res):
            wok, wpath, wst = want[k % len(cases)]
            assert (ok, path) == (wok, wpath) == (rok, rpath), (world, k)
            assert st["nodes"] == wst["nodes"] and st["expanded"] == wst["expanded"], (world, k, st, wst)
            assert (st["levels"], st["chunks"], st["replicated_levels"]) == (rst["levels"], rst["chunks"], rst["replicated_levels"]), (world, k, st, rst)
            assert st["min_len"] == rst["min_len"], (world, k, st, rst)


@pytest.mark.timeout(600)
def test_native_sharded_bfs_a_failing_rank_ends_every_rank(search):
    """acx_shard_opts.fail_at_call: the n-th engine call of rank 1 fails instead of being made -- in the replicated phase, at the
    partition, in the middle of exchanged levels.  Every rank must come back with an error (thread ranks meet at barriers: a rank left
    alone in a collective would hang this test), and the next search on the same communicator must run."""
    from ac_solver.search.sharded import NativeComm, bfs_sharded_native
    from oracle import ac_oracle as O
    from tests.shard_helpers import run_threads

    ak3 = np.zeros(50, np.int8)
    ak3[:7] = [1, 1, 1, -2, -2, -2, -2]
    ak3[25:31] = [1, 2, 1, -2, -1, -2]
    want = O.bfs(ak3, 30000)

    def run(comm):
        nat = NativeComm.from_python(comm)
        out = []
        for fail_at in (2, 7, 15, 16, 17, 25, 40, 61, 90):
            try:
                bfs_sharded_native(ak3, 30000, comm=nat, batch_parents=256, replicate_below=40, _fail_at_call=fail_at, _fail_rank=1)
                out.append("no error")
            except RuntimeError as e:
                out.append(str(e))
        out.append(bfs_sharded_native(ak3, 30000, comm=nat, batch_parents=256, replicate_below=40))
        # a record log sized for 1/256 of the budget: it has to grow (by doubling) several times in mid-search
        assert bfs_sharded_native(ak3, 30000, comm=nat, batch_parents=256, replicate_below=40, log_fraction=1 / 256) == out[-1]
        assert not nat.errors, nat.errors
        return out

    got = run_threads(3, run)
    for r, res in enumerate(got):
        assert all("sharded bfs failed" in m for m in res[:-1]), (r, res[:-1])
        assert res[-1] == want
    assert all("simulated" in m for m in got[1][:-1]), got[1][:-1]


def test_rccl_communicator_of_the_library(search):
    """acx_comm_rccl: librccl resolved at run time, a communicator of the library's own (acx_rccl_unique_id / acx_rccl_comm_create, world 1 on
    the one GPU of a box) -- its two collectives called through the acx_comm function pointers on device buffers, then a search on it."""
    import ctypes as C

    import torch

    from ac_solver import _acx
    from ac_solver.search.sharded import NativeComm, bfs_sharded_native
    from oracle import ac_oracle as O

    assert _acx.lib.acx_rccl_available() == 1
    comm = NativeComm.create_rccl(0, 1, lambda b: b)
    try:
        assert (comm.rank, comm.world) == (0, 1)
        st = torch.cuda.current_stream().cuda_stream
        a = torch.arange(1000, dtype=torch.int64, device="cuda")
        b = torch.zeros_like(a)
        assert comm.c.all_to_all(comm.c.ctx, a.data_ptr(), b.data_ptr(), a.numel(), st) == 0
        torch.cuda.synchronize()
        assert torch.equal(a, b)
        m = torch.arange(77, dtype=torch.int32, device="cuda")
        assert comm.c.all_reduce(comm.c.ctx, m.data_ptr(), m.numel(), _acx.I32, _acx.RED_SUM, st) == 0
        assert comm.c.all_reduce(comm.c.ctx, a.data_ptr(), a.numel(), _acx.I64, _acx.RED_MAX, st) == 0
        torch.cuda.synchronize()
        assert torch.equal(m.cpu(), torch.arange(77, dtype=torch.int32)) and torch.equal(a.cpu(), torch.arange(1000))
        ak2 = [1, 1, -2, -2, -2, 0, 0, 1, 2, 1, -2, -1, -2, 0]
        assert bfs_sharded_native(ak2, 10**6, comm=comm) == O.bfs(ak2, 10**6)
    finally:
        comm.close()


def test_sharded_bfs_over_rccl_process_group(search):
    """The production communicator (torch.distributed, backend nccl == RCCL) with device tensors, world size 1:
    exercises the equal-split all_to_all_single and the all_reduce on the HIP engine's buffers."""
    import socket

    import torch
    import torch.distributed as dist

    from ac_solver.search.sharded import TorchDistComm, bfs_sharded
    from oracle import ac_oracle as O

    if dist.is_initialized():
        pytest.skip("a process group already exists")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        ak2 = [1, 1, -2, -2, -2, 0, 0, 1, 2, 1, -2, -1, -2, 0]
        # the mask all-reduce on the communicator everything else uses, and on one of its own (dist.new_group)
        for comm in (TorchDistComm(torch.device("cuda", 0)), TorchDistComm(torch.device("cuda", 0), mask_group="own")):
            for budget in (10, 5000, 10**6):
                got = _bfs_through_comm(bfs_sharded, ak2, budget, comm)
                assert got == O.bfs(ak2, budget)
            assert comm.stats["mask_all_reduce_calls"] > 0 and comm.stats["all_to_all_calls"] > 0
        # the process group's own ncclComm_t handed to the library (ProcessGroupNCCL._comm_ptr) and `bfs` switched to the collective form
        from ac_solver.search.breadth_first import bfs, shard_over_process_group
        from ac_solver.search.sharded import NativeComm, bfs_sharded_native

        nat = NativeComm.from_process_group()
        assert (nat.rank, nat.world) == (0, 1)
        assert bfs_sharded_native(ak2, 10**6, comm=nat) == O.bfs(ak2, 10**6)
        old = shard_over_process_group(True, min_nodes=0)
        try:
            assert bfs(ak2, 10**6) == O.bfs(ak2, 10**6)
        finally:
            shard_over_process_group(old)
        import ac_solver.search.sharded as sh

        sh._FORCE_EXCHANGE = True
        try:  # the communicator itself copes with the aliased send / receive areas of world 1; the per-stage timeline comes back
            ok, path, st = bfs_sharded(ak2, 10**6, comm=comm, want_stats=True, timeline=True, batch_parents=1 << 14)
        finally:
            sh._FORCE_EXCHANGE = False
        assert (ok, path) == O.bfs(ak2, 10**6)
        tl = st["timeline"]
        assert tl["chunks_timed"] > 0 and tl["expand_us"] > 0 and tl["insert_us"] > 0 and tl["commit_us"] > 0 and tl["all_to_all_us"] > 0 and tl["mask_all_reduce_us"] > 0
    finally:
        dist.destroy_process_group()


def _bfs_through_comm(bfs_sharded, p, budget, comm):
    import torch

    import ac_solver.search.sharded as sh

    class Forced:
        """world-size-1 communicator that still routes every call through torch.distributed"""
        rank, world = 0, 1

        def all_to_all_single(self, recv, send):
            out = torch.empty_like(send)  # RCCL rejects aliased buffers: at world 1 the engine's send and receive areas are the same
            comm.all_to_all_single(out, send)
            recv.copy_(out)

        def all_reduce(self, t, op):
            return comm.all_reduce(t, op)

        def all_reduce_masks(self, t):
            return comm.all_reduce_masks(t) if hasattr(comm, "all_reduce_masks") else comm.all_reduce(t, "sum")

    saved = sh.bfs_sharded.__globals__.get("_FORCE_EXCHANGE")
    sh._FORCE_EXCHANGE = True
    try:
        return bfs_sharded(p, budget, comm=Forced())
    finally:
        sh._FORCE_EXCHANGE = saved


@pytest.mark.timeout(600)
@pytest.mark.parametrize("cyclical", [False, True])
def test_config4_all_1190_ms_presentations_bfs_1e4_vs_oracle(search, golden_json, cyclical):
    """BASELINE config 4 / SURVEY 8(d): bfs on ALL 1190 Miller-Schupp presentations (native max_relator_length 18..36, so both
    key widths) with budget 1e4 -- (solved, path) and the node / expansion counts against the C oracle, one by one."""
    from ac_solver import _acx
    from ac_solver.search._common import run_search_many
    from oracle import ac_oracle as O

    pool = ms_pool_generator_order(golden_json("ms_pool.json"))
    assert len(pool) == 1190
    bad = []
    for lo in range(0, 1190, 170):
        rows = np.array(pool[lo:lo + 170], dtype=np.int8)
        got = run_search_many(_acx.SEARCH_BFS, rows, 10**4, cyclical, n_threads=16)
        for k, (ok, path, st) in enumerate(got):
            wok, wpath, wst = O.bfs(rows[k], 10**4, cyclically_reduce_after_moves=cyclical, stats=True)
            if (ok, path) != (wok, wpath) or st["nodes"] != wst["nodes"] or st["expanded"] != wst["expanded"]:
                bad.append((lo + k, ok, wok, st["nodes"], wst["nodes"], st["expanded"], wst["expanded"]))
    assert not bad, bad[:10]


@pytest.mark.timeout(1800)
@pytest.mark.parametrize("world", [2, 8])
@pytest.mark.parametrize("cyclical", [False, True])
def test_config4_all_1190_ms_presentations_through_the_sharded_engine(search, golden_json, world, cyclical):
    """BASELINE config 4 as it is worded -- bfs over the Miller-Schupp presentations with the frontier SHARDED: all 1190 presentations
    (both key widths), budget 1e4, through acx_bfs_sharded on 2 and 8 thread ranks of the one GPU, the frontier partitioned by owner at
    the first level of >= 64 parents and exchanged in chunks of 1024 parents from there on; (solved, path) and the node / expansion
    counts of every search on every rank against the same oracle rows as the single-GPU test above (breadth_first.py:61-95)."""
    from ac_solver.search.sharded import NativeComm, bfs_sharded_native
    from oracle import ac_oracle as O
    from tests.shard_helpers import run_threads

    pool = ms_pool_generator_order(golden_json("ms_pool.json"))
    assert len(pool) == 1190
    rows = [np.array(p, dtype=np.int8) for p in pool]
    want = [O.bfs(r, 10**4, cyclically_reduce_after_moves=cyclical, stats=True) for r in rows]

    def run(comm):
        nat = NativeComm.from_python(comm)
        out = [bfs_sharded_native(r, 10**4, cyclically_reduce_after_moves=cyclical, comm=nat, batch_parents=1024, replicate_below=64, want_stats=True) for r in rows]
        assert not nat.errors, nat.errors[:1]
        return out

    got = run_threads(world, run)
    bad = []
    exchanged = 0
    for r, res in enumerate(got):
        for k, ((ok, path, st), (wok, wpath, wst)) in enumerate(zip(res, want)):
            if (ok, path) != (wok, wpath) or st["nodes"] != wst["nodes"] or st["expanded"] != wst["expanded"]:
                bad.append((r, k, ok, wok, st["nodes"], wst["nodes"], st["expanded"], wst["expanded"]))
            exchanged += r == 0 and st["levels"] > st["replicated_levels"]
    assert not bad, bad[:10]
    assert exchanged > 1000, exchanged  # (nearly every search reaches a level of 64 parents within its 1e4 nodes: the exchange really ran)


def _two_process_worker(rank, world, port, q):
    """child process of test_sharded_bfs_two_processes_share_one_gpu: its own HIP context on cuda:0, gloo between the processes"""
    import os

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ac_solver.search.sharded import TorchDistComm, bfs_sharded

        class HostStaged(TorchDistComm):
            """gloo moves host tensors: device tensors are staged through the host around every collective"""

            def __init__(self):
                super().__init__(torch.device("cpu"))

            def all_to_all_single(self, recv, send):
                h = torch.empty(send.shape, dtype=send.dtype)
                super().all_to_all_single(h, send.cpu())
                recv.copy_(h)

            def all_reduce(self, t, op):
                h = t.cpu()
                super().all_reduce(h, op)
                t.copy_(h)
                return t

        torch.cuda.set_device(0)
        comm = HostStaged()
        ak2 = [1, 1, -2, -2, -2, 0, 0, 1, 2, 1, -2, -1, -2, 0]
        ak3 = np.zeros(50, np.int8)
        ak3[:7] = [1, 1, 1, -2, -2, -2, -2]
        ak3[25:31] = [1, 2, 1, -2, -1, -2]
        res = [bfs_sharded(p, b, cyclically_reduce_after_moves=c, comm=comm, batch_parents=bp, want_stats=True)
               for p, b, c, bp in ((ak2, 10**6, False, 1 << 16), (ak2, 500, True, 7), (ak3, 300000, False, 1 << 14), (ak3, 20000, True, 333))]
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_sharded_bfs_two_processes_share_one_gpu():
    """The first multi-PROCESS run of acx_shard_*: two freshly spawned processes (the parent touches no GPU API before the
    spawn), each with its own engine on cuda:0, exchanging records over gloo with host-staged tensors.  Every rank must
    return the oracle's answer."""
    import socket

    import torch.multiprocessing as mp

    from oracle import ac_oracle as O

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_two_process_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    got = dict(q.get(timeout=500) for _ in range(2))
    [p.join(120) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    ak2 = [1, 1, -2, -2, -2, 0, 0, 1, 2, 1, -2, -1, -2, 0]
    ak3 = np.zeros(50, np.int8)
    ak3[:7] = [1, 1, 1, -2, -2, -2, -2]
    ak3[25:31] = [1, 2, 1, -2, -1, -2]
    cases = ((ak2, 10**6, False), (ak2, 500, True), (ak3, 300000, False), (ak3, 20000, True))
    for r in (0, 1):
        for (p, b, c), (ok, path, st) in zip(cases, got[r]):
            wok, wpath, wst = O.bfs(p, b, cyclically_reduce_after_moves=c, stats=True)
            assert (ok, path) == (wok, wpath), (r, b, c)
            assert st["nodes"] == wst["nodes"] and st["expanded"] == wst["expanded"], (r, b, c, st, wst)


def test_verbose_prints_the_references_lines(search, golden_json, capsys):
    """verbose=True: the same lines, in the same order, as the reference prints (tests/golden/verbose_lines.json holds its
    captured stdout): every "New minimal length found", the greedy success report, the budget message."""
    from ac_solver import bfs, greedy_search

    for row in golden_json("verbose_lines.json"):
        fn = bfs if row["algo"] == "bfs" else greedy_search
        capsys.readouterr()
        ok, _ = fn(np.array(row["presentation"], dtype=np.int8), max_nodes_to_explore=row["budget"], verbose=True,
                   cyclically_reduce_after_moves=row["cyclical"])
        out = capsys.readouterr().out.splitlines()
        assert ok == row["solved"] and out == row["lines"], (row["algo"], row["budget"], row["cyclical"], out, row["lines"])
