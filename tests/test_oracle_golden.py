"""Pins the CPU oracle (oracle/ac_oracle.c) against golden vectors produced by the Python
reference (oracle/tools/make_golden.py) and against the data the reference's own tests hold.
CPU only."""
import numpy as np
import pytest

from tests.conftest import ms_pool_generator_order
from oracle import ac_oracle as O

EXC = {1: AssertionError, 2: IndexError, 3: ValueError}


def test_simplify_relator_reference_table(golden_json):
    for r in golden_json("unit_tables.json")["simplify_relator"]:
        out, n = O.simplify_relator(np.array(r["relator"]), r["L"], cyclical=r["cyclical"], padded=r["padded"])
        assert out.tolist() == r["out"] and n == r["length"], r


def test_simplify_relator_fuzz(golden_json):
    for r in golden_json("simplify_fuzz.json"):
        if r["err"]:
            with pytest.raises(EXC[r["err"]]):
                O.simplify_relator(np.array(r["relator"]), r["L"], cyclical=r["cyclical"], padded=r["padded"])
        else:
            out, n = O.simplify_relator(np.array(r["relator"]), r["L"], cyclical=r["cyclical"], padded=r["padded"])
            assert out.tolist() == r["out"] and n == r["length"], r


def test_validity_and_triviality_tables(golden_json):
    t = golden_json("unit_tables.json")
    for r in t["is_array_valid_presentation"]:
        assert O.is_array_valid_presentation(r["array"]) == r["valid"], r
    for r in t["is_presentation_trivial"]:
        assert O.is_presentation_trivial(r["array"]) == r["trivial"], r


def test_simplify_presentation_table(golden_json):
    for r in golden_json("unit_tables.json")["simplify_presentation"]:
        out, lens = O.simplify_presentation(r["presentation"], r["L"], r["lengths"])
        assert out.tolist() == r["out"] and lens == r["out_lengths"]


@pytest.mark.parametrize("name", ["concatenate_relators", "conjugate"])
def test_move_tables(golden_json, name):
    fn = getattr(O, name)
    for r in golden_json("unit_tables.json")[name]:
        out, lens = fn(r["presentation"], r["L"], r["i"], r["j"], r["sign"], r["lengths"])
        assert out.tolist() == r["out"] and lens == r["out_lengths"], r


def test_raw_moves_fuzz(golden_json):
    for r in golden_json("moves_raw_fuzz.json"):
        fn = O.concatenate_relators if r["fn"] == "cat" else O.conjugate
        if r["err"]:
            with pytest.raises(EXC[r["err"]]):
                fn(r["p"], r["L"], r["i"], r["j"], r["sign"], r["lengths"])
        else:
            out, lens = fn(r["p"], r["L"], r["i"], r["j"], r["sign"], r["lengths"])
            assert out.tolist() == r["out"] and lens == r["out_lengths"], r


def test_acmove_table(golden_json):
    for r in golden_json("unit_tables.json")["ACMove"]:
        out, lens = O.ACMove(r["move"], r["presentation"], r["L"], [4, 4], cyclical=r["cyclical"])
        assert out.tolist() == r["out"] and lens == r["out_lengths"]


@pytest.mark.parametrize("L", [2, 3, 4, 5, 7, 12, 25, 36])
def test_acmove_fuzz(golden_npz, L):
    z = golden_npz("acmove_fuzz.npz")
    st, mv, cy = z[f"L{L}_state"], z[f"L{L}_move"], z[f"L{L}_cyclical"]
    for c in (0, 1):
        m = cy == c
        out, lens, err = O.move_batch(st[m], mv[m], L, cyclical=bool(c))
        assert np.array_equal(err, z[f"L{L}_err"][m])
        assert np.array_equal(out, z[f"L{L}_out"][m])
        assert np.array_equal(lens, z[f"L{L}_lens"][m])


def test_stable_ak3_notebook_sequence(golden_json):
    g = golden_json("stable_ak3.json")
    L = g["L"]
    state = np.array(g["relator1"] + [0] * (L - len(g["relator1"])) + g["relator2"] + [0] * (L - len(g["relator2"])), dtype=np.int8)
    lens = None
    for m in g["sequence_one_based"]:
        state, lens = O.ACMove(m - 1, state, L, lens, cyclical=False)
    assert state.tolist() == g["end_state"] and lens == g["end_lengths"]


def test_env_trajectories(golden_npz):
    z = golden_npz("env_traj.npz")
    T, N = z["tape"].shape
    states = z["init"].copy()
    counts = np.zeros(N, np.int32)
    horizon = int(z["horizon"])
    t_prev = 0
    rew, done, trunc = [], [], []
    for t in (0, 1, 7, 31, 127):
        r, d, tr, err = O.env_rollout(states, counts, horizon, np.ascontiguousarray(z["tape"][t_prev:t + 1]))
        assert not err.any()
        assert np.array_equal(states, z[f"state_t{t}"]), t
        rew.append(r), done.append(d), trunc.append(tr)
        t_prev = t + 1
    assert np.array_equal(np.concatenate(rew), z["reward"])
    assert np.array_equal(np.concatenate(done), z["done"])
    assert np.array_equal(np.concatenate(trunc), z["truncated"])


def test_env_episodes(golden_json):
    for ep in golden_json("env_episodes.json")["episodes"]:
        s = np.array(ep["initial_state"], dtype=np.int8)[None].copy()
        c = np.zeros(1, np.int32)
        for st in ep["steps"]:
            r, d, tr, err = O.env_rollout(s, c, ep["horizon"], np.array([[st["action"]]], dtype=np.uint8))
            assert s[0].tolist() == st["state"]
            assert (int(r[0, 0]), bool(d[0, 0]), bool(tr[0, 0])) == (st["reward"], st["done"], st["truncated"]), (ep["name"], st)


def test_search_golden(golden_json):
    for r in golden_json("search.json"):
        fn = O.bfs if r["algo"] == "bfs" else O.greedy_search
        ok, path = fn(r["presentation"], r["budget"], cyclically_reduce_after_moves=r["cyclical"])
        want = [tuple(x) for x in r["path"]] if r["path"] is not None else None
        assert ok == r["solved"] and path == want, (r["tag"], r["algo"], r["budget"])


def test_greedy_paths_file_all_533(golden_json):
    """data/greedy_search_paths.txt (budget 1e6): every path is reproduced exactly."""
    pool = ms_pool_generator_order(golden_json("ms_pool.json"))
    g = golden_json("greedy_paths_1e6.json")
    assert len(g["rows"]) == 533
    for row in g["rows"]:
        ok, path = O.greedy_search(pool[row["pool_index"]], g["budget"])
        assert ok and path == [tuple(x) for x in row["path"]], row["pool_index"]


# ---- neighbourhood sizes: the C restatement against the reference's own program ----------------
def test_ball_oracle_matches_reference_program(golden_json):
    """oracle/ac_ball_oracle.c vs the sizes printed by the reference's neibourhoods.cpp (tests/golden/ball_sizes.json, made by
    oracle/tools/make_ball_golden.py from oracle/_ref/ball_ref), including the five known answers of its README."""
    from oracle import ac_oracle as O

    cases = golden_json("ball_sizes.json")["cases"]
    readme = [c["size"] for c in cases if c["tag"] == "readme" and not c["classic"]]
    assert readme == [28631, 49668, 72392, 28631, 28631]  # README.txt:38-44
    for c in cases:
        if c["radius"] == 5 and c["tag"] not in ("readme", "ms_0", "ms_1100"):
            continue  # the full radius on a few, the smaller radii on all (keeps the CPU suite short)
        assert O.ball_size(c["presentation"], c["radius"], c["classic"]) == c["size"], (c["tag"], c["radius"], c["classic"])


def test_simplex_oracle_matches_reference_program(golden_npz):
    """oracle/ac_ball_oracle.c:ac_simplex_graph vs the files written by the reference's ac_bfs.cpp (prime and classic),
    tests/golden/simplex_data.npz made by oracle/tools/make_simplex_golden.py from oracle/_ref/simplex_*."""
    from oracle import ac_oracle as O

    g = golden_npz("simplex_data.npz")
    for tag, classic in (("prime", False), ("classic", True)):
        for n in (4, 6, 8):
            sizes, edges, filt = O.simplex_graph(n, classic)
            assert np.array_equal(sizes, g[f"{tag}_{n}_node_size"]) and np.array_equal(edges, g[f"{tag}_{n}_edges"])
            assert np.array_equal(filt, g[f"{tag}_{n}_edge_filt"])
    assert len(g["prime_8_node_size"]) == 9172 and len(g["prime_8_edges"]) == 17416 and len(g["classic_8_edges"]) == 26363


# ---- the pure Python / NumPy restatement (bench.py's third CPU-baseline leg) against the same reference fixtures ----
def test_numpy_restatement_tables(golden_json):
    from oracle import ac_numpy as P

    t = golden_json("unit_tables.json")
    for r in t["simplify_relator"]:
        out, n = P.simplify_relator(np.array(r["relator"]), r["L"], cyclical=r["cyclical"], padded=r["padded"])
        assert out.tolist() == r["out"] and n == r["length"], r
    for r in t["is_array_valid_presentation"]:
        assert P.is_array_valid_presentation(np.array(r["array"])) == r["valid"], r
    for name in ("concatenate_relators", "conjugate"):
        for r in t[name]:
            out, lens = getattr(P, name)(np.array(r["presentation"]), r["L"], r["i"], r["j"], r["sign"], r["lengths"])
            assert out.tolist() == r["out"] and lens == r["out_lengths"], (name, r)
    for r in t["ACMove"]:
        out, lens = P.ACMove(r["move"], np.array(r["presentation"]), r["L"], [4, 4], cyclical=r["cyclical"])
        assert out.tolist() == r["out"] and lens == r["out_lengths"]


@pytest.mark.parametrize("L", [3, 7, 25, 36])
def test_numpy_restatement_acmove_fuzz_sample(golden_npz, L):
    from oracle import ac_numpy as P

    z = golden_npz("acmove_fuzz.npz")
    st, mv, cy, err = z[f"L{L}_state"], z[f"L{L}_move"], z[f"L{L}_cyclical"], z[f"L{L}_err"]
    def padded_words(row):  # both halves non-empty words padded on the right: the domain the searches and the env live in
        return all(0 < np.count_nonzero(h) and not h[np.count_nonzero(h):].any() and h[:np.count_nonzero(h)].all() for h in (row[:L], row[L:]))

    # the cases where the reference returns; its raises and its behaviour on malformed rows are pinned on the C oracle
    pick = [k for k in np.flatnonzero(err == 0)[::5] if padded_words(st[k])][:500]
    assert len(pick) > 100
    for k in pick:
        row = st[k]
        lens = [int(np.count_nonzero(row[:L])), int(np.count_nonzero(row[L:]))]
        out, ol = P.ACMove(int(mv[k]), row.copy(), L, lens, cyclical=bool(cy[k]))
        assert out.tolist() == z[f"L{L}_out"][k].tolist() and ol == z[f"L{L}_lens"][k].tolist(), k


def test_numpy_restatement_env_and_searches(golden_npz, golden_json):
    from oracle import ac_numpy as P

    z = golden_npz("env_traj.npz")
    for e in range(0, 1024, 97):  # a few environments of the BASELINE config-2 recipe, first 32 steps
        env = P.Env(z["init"][e], horizon_length=int(z["horizon"]))
        for t in range(32):
            _, r, d, tr = env.step(int(z["tape"][t, e]))
            assert (int(r), bool(d), bool(tr)) == (int(z["reward"][t, e]), bool(z["done"][t, e]), bool(z["truncated"][t, e])), (e, t)
        assert env.state.tolist() == z["state_t31"][e].tolist()
    n = 0
    for r in golden_json("search.json"):
        if r["budget"] > 2000:
            continue
        fn = P.bfs if r["algo"] == "bfs" else P.greedy_search
        ok, path = fn(r["presentation"], r["budget"], cyclically_reduce_after_moves=r["cyclical"])
        want = [tuple(x) for x in r["path"]] if r["path"] is not None else None
        assert ok == r["solved"] and path == want, (r["tag"], r["algo"], r["budget"])
        n += 1
    assert n >= 20
