#!/usr/bin/env python3
"""Generate tests/golden/* from the Python reference (build container only).

Run from the repo root:

    python oracle/tools/make_golden.py [--skip-slow]

It puts oracle/tools/stubs (a stand-in for the absent `gymnasium`/`wandb` imports) and
/root/reference on sys.path, imports the reference's ac_solver package, runs the
reference functions on seeded inputs and writes ONLY inputs + expected outputs (data)
into tests/golden/.  Nothing of the reference's source or bytecode is copied.  The GPU
box never runs this script and never sees /root/reference.

Fixtures written
  unit_tables.json      argument/expectation tables the reference's tests/test_ac_env.py
                        parametrizes over (read from the pytest marks) + what the reference
                        returns for them; helper-function outputs
  acmove_fuzz.npz       ACMove on seeded random / adversarial states, all 12 moves, both
                        `cyclical` values, L in {2,3,4,5,7,12,25,36}; includes the reference's
                        AssertionError / IndexError cases
  simplify_fuzz.json    simplify_relator on seeded random words (generic letters, all flags)
  env_traj.npz          ACEnv.step trajectories, BASELINE config 2 recipe (first 1024 envs x
                        128 steps) + scripted done / truncated episodes
  search.json           bfs / greedy_search (solved, path) on AK(2), AK(3), MS presentations,
                        several budgets, both cyclical values
  ms_pool.json          generate_miller_schupp_presentations for n=1..7, max_w_len=7 and the
                        small cases; the data/*.txt orderings expressed as index lists
  greedy_paths_1e6.json data/greedy_search_paths.txt re-encoded to today's (action, length)
                        convention (file actions are 1-based, root (0, len))
  stable_ak3.json       notebooks/Stable-AK3.ipynb cell 1: move list and printed end state
"""
import argparse
import ast
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(HERE, "stubs"))

from ac_solver.envs import ac_moves as R_moves  # noqa: E402
from ac_solver.envs import utils as R_utils  # noqa: E402
from ac_solver.envs.ac_env import ACEnv, ACEnvConfig  # noqa: E402
from ac_solver.search.breadth_first import bfs as R_bfs  # noqa: E402
from ac_solver.search.greedy import greedy_search as R_greedy  # noqa: E402
from ac_solver.search.miller_schupp import miller_schupp as R_ms  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")
ERR = {AssertionError: 1, IndexError: 2, ValueError: 3}


def jl(x):
    """numpy -> plain lists for json"""
    if isinstance(x, np.ndarray):
        return x.tolist()
    if isinstance(x, (list, tuple)):
        return [jl(v) for v in x]
    if isinstance(x, dict):
        return {str(k): jl(v) for k, v in x.items()}
    if isinstance(x, (np.integer,)):
        return int(x)
    if isinstance(x, (np.bool_,)):
        return bool(x)
    return x


def dump(name, obj):
    path = os.path.join(OUT, name)
    with open(path, "w") as f:
        json.dump(jl(obj), f, separators=(",", ":"))
    print(f"wrote {name}: {os.path.getsize(path) / 1024:.1f} KiB")


def call(fn, *a, **k):
    """-> (err_code, result)"""
    try:
        return 0, fn(*a, **k)
    except tuple(ERR) as e:  # noqa: B030
        return ERR[type(e)], None


# ---------------------------------------------------------------------------------------------
def unit_tables():
    import importlib

    sys.path.insert(0, REF)
    T = importlib.import_module("tests.test_ac_env")

    def table(fn):
        mark = [m for m in fn.pytestmark if m.name == "parametrize"][0]
        return mark.args[1]

    out = {}
    rows = []
    for rel, L, cyc, padded, exp_rel, exp_len in table(T.test_simplify_relator):
        err, res = call(R_utils.simplify_relator, np.array(rel), L, cyclical=cyc, padded=padded)
        assert err == 0 and np.array_equal(res[0], exp_rel) and res[1] == exp_len
        rows.append(dict(relator=jl(rel), L=L, cyclical=cyc, padded=padded, out=jl(exp_rel), length=exp_len))
    out["simplify_relator"] = rows
    rows = []
    for p, exp in table(T.test_is_array_valid_presentation):
        assert R_utils.is_array_valid_presentation(p) == exp
        rows.append(dict(array=jl(p), valid=exp))
    # a few more shapes the reference accepts (lists, empties)
    for p in ([], [1, 0, 0], [1, 2, 0, 0, -2, -1, 0, 0], [0, 1, 2, 0], [1, -1, 2, 2]):
        rows.append(dict(array=p, valid=bool(R_utils.is_array_valid_presentation(list(p)))))
    out["is_array_valid_presentation"] = rows
    rows = []
    for p, exp in table(T.test_is_presentation_trivial):
        assert R_utils.is_presentation_trivial(p) == exp
        rows.append(dict(array=jl(p), trivial=exp))
    out["is_presentation_trivial"] = rows
    rows = []
    for p, L, lens, exp, exp_lens in table(T.test_simplify_presentation):
        res = R_utils.simplify_presentation(p, L, lens)
        assert np.array_equal(res[0], exp) and res[1] == exp_lens
        rows.append(dict(presentation=jl(p), L=L, lengths=lens, out=jl(exp), out_lengths=exp_lens))
    out["simplify_presentation"] = rows
    for name, fn, ref in (("concatenate_relators", T.test_concatenate_relators, R_moves.concatenate_relators),
                          ("conjugate", T.test_conjugate, R_moves.conjugate)):
        rows = []
        for rels, L, i, j, sign, lens, exp, exp_lens in table(fn):
            res = ref(rels, L, i, j, sign, list(lens))
            assert np.array_equal(res[0], exp) and res[1] == exp_lens
            rows.append(dict(presentation=jl(rels), L=L, i=i, j=j, sign=sign, lengths=lens, out=jl(exp), out_lengths=exp_lens))
        out[name] = rows
    # test_ACMove's single state: all 12 moves, both cyclical values (the reference test covers 5 of them)
    rows = []
    init = np.array([1, 2, 0, 0, -2, 0, 0, 0])
    for cyc in (True, False):
        for m in range(12):
            err, res = call(R_moves.ACMove, m, init, 4, [4, 4], cyclical=cyc)
            rows.append(dict(presentation=jl(init), L=4, move=m, cyclical=cyc, err=err,
                             out=jl(res[0]) if res else None, out_lengths=jl(res[1]) if res else None))
    out["ACMove"] = rows
    # helpers
    out["generate_trivial_states"] = {str(L): jl(R_utils.generate_trivial_states(L)) for L in (1, 2, 3, 4, 25)}
    ak2 = R_utils.convert_relators_to_presentation([1, 1, -2, -2, -2], [1, 2, 1, -2, -1, -2], 7)
    out["convert_relators_to_presentation"] = [
        dict(r1=[1, 1, -2, -2, -2], r2=[1, 2, 1, -2, -1, -2], L=7, out=jl(ak2), dtype=str(ak2.dtype)),
        dict(r1=[1], r2=[2], L=1, out=jl(R_utils.convert_relators_to_presentation([1], [2], 1)), dtype="int8"),
    ]
    p36 = R_utils.change_max_relator_length_of_presentation(list(ak2.tolist()), 36)
    out["change_max_relator_length_of_presentation"] = [dict(presentation=jl(ak2), new_L=36, out=jl(p36))]
    dump("unit_tables.json", out)


# ---------------------------------------------------------------------------------------------
def rand_word(rng, n, letters=(1, -1, 2, -2), reduced=True):
    w = []
    while len(w) < n:
        c = int(rng.choice(letters))
        if reduced and w and w[-1] == -c:
            continue
        w.append(c)
    return w


def embed(r0, r1, L):
    return np.array(list(r0) + [0] * (L - len(r0)) + list(r1) + [0] * (L - len(r1)), dtype=np.int8)


def fuzz_states(rng, L, count):
    """Seeded mix of reduced, unreduced, related, over-long-product and invalid presentations."""
    S = []
    for k in range(count):
        kind = k % 10
        n0 = int(rng.integers(1, L + 1))
        n1 = int(rng.integers(1, L + 1))
        if kind <= 3:  # freely reduced (not necessarily cyclically)
            r0, r1 = rand_word(rng, n0), rand_word(rng, n1)
        elif kind == 4:  # unreduced
            r0, r1 = rand_word(rng, n0, reduced=False), rand_word(rng, n1, reduced=False)
        elif kind == 5:  # r1 = r0 or r0^-1 (concat cancels everything -> AssertionError in the reference)
            r0 = rand_word(rng, n0)
            r1 = list(r0) if rng.integers(2) else [-c for c in reversed(r0)]
        elif kind == 6:  # long shared junction: r1 = inv(suffix of r0) + tail
            r0 = rand_word(rng, n0)
            s = int(rng.integers(0, n0 + 1))
            r1 = [-c for c in reversed(r0[n0 - s:])] + rand_word(rng, int(rng.integers(0, max(1, L - s))))
            r1 = r1[:L] or [1]
        elif kind == 7:  # short words (conjugation end cases, length 1/2)
            r0, r1 = rand_word(rng, int(rng.integers(1, min(3, L) + 1))), rand_word(rng, int(rng.integers(1, min(3, L) + 1)))
        elif kind == 8:  # generic letters up to 6 (the reference's own tests use them)
            lt = (1, -1, 2, -2, 3, -3, 4, -5, 6)
            r0, r1 = rand_word(rng, n0, lt), rand_word(rng, n1, lt)
        else:  # invalid: interior zeros / empty relator
            r0, r1 = rand_word(rng, n0), rand_word(rng, n1)
            p = embed(r0, r1, L)
            z = rng.integers(0, 2 * L, size=int(rng.integers(1, 4)))
            p[z] = 0
            if rng.integers(4) == 0:
                h = int(rng.integers(2))
                p[h * L:(h + 1) * L] = 0
            S.append(p)
            continue
        S.append(embed(r0, r1, L))
    return np.stack(S)


def acmove_fuzz():
    rng = np.random.default_rng(20241022)
    out = {}
    total = errs = 0
    for L, count in ((2, 120), (3, 160), (4, 200), (5, 200), (7, 300), (12, 300), (25, 500), (36, 200)):
        states = fuzz_states(rng, L, count)
        S, M, Cy, O, Ln, E = [], [], [], [], [], []
        for s in states:
            for cyc in (0, 1):
                for m in range(12):
                    lens = [int(np.count_nonzero(s[:L])), int(np.count_nonzero(s[L:]))]
                    err, res = call(R_moves.ACMove, m, s.copy(), L, lens, cyclical=bool(cyc))
                    S.append(s); M.append(m); Cy.append(cyc); E.append(err)
                    if err:
                        O.append(s); Ln.append([int(np.count_nonzero(s[:L])), int(np.count_nonzero(s[L:]))]); errs += 1  # the reference mutates `lens` before raising
                    else:
                        O.append(np.asarray(res[0], dtype=np.int8)); Ln.append([int(v) for v in res[1]])
                    total += 1
        out[f"L{L}_state"] = np.stack(S).astype(np.int8)
        out[f"L{L}_move"] = np.array(M, dtype=np.uint8)
        out[f"L{L}_cyclical"] = np.array(Cy, dtype=np.uint8)
        out[f"L{L}_out"] = np.stack(O).astype(np.int8)
        out[f"L{L}_lens"] = np.array(Ln, dtype=np.int32)
        out[f"L{L}_err"] = np.array(E, dtype=np.uint8)
    path = os.path.join(OUT, "acmove_fuzz.npz")
    np.savez_compressed(path, **out)
    print(f"wrote acmove_fuzz.npz: {total} cases ({errs} reference exceptions), {os.path.getsize(path) / 1024:.1f} KiB")

    # concatenate_relators / conjugate alone (no simplify), incl. generic letters and invalid rows
    rows = []
    rng = np.random.default_rng(7)
    for L in (3, 4, 7, 25):
        for s in fuzz_states(rng, L, 60):
            lens = [int(np.count_nonzero(s[:L])), int(np.count_nonzero(s[L:]))]
            for i, j, sign in ((0, 1, 1), (0, 1, -1), (1, 0, 1), (1, 0, -1)):
                err, res = call(R_moves.concatenate_relators, s.copy(), L, i, j, sign, list(lens))
                rows.append(dict(fn="cat", p=jl(s), L=L, i=i, j=j, sign=sign, lengths=lens, err=err,
                                 out=jl(res[0]) if res else None, out_lengths=jl(res[1]) if res else None))
            for i in (0, 1):
                for j in (1, 2):
                    for sign in (1, -1):
                        err, res = call(R_moves.conjugate, s.copy(), L, i, j, sign, list(lens))
                        rows.append(dict(fn="conj", p=jl(s), L=L, i=i, j=j, sign=sign, lengths=lens, err=err,
                                         out=jl(res[0]) if res else None, out_lengths=jl(res[1]) if res else None))
    dump("moves_raw_fuzz.json", rows)


def simplify_fuzz():
    rng = np.random.default_rng(11)
    rows = []
    for k in range(400):
        n = int(rng.integers(0, 14))
        lt = (1, -1, 2, -2) if k % 3 else (1, -1, 2, -2, 3, -3)
        w = rand_word(rng, n, lt, reduced=False)
        pad = int(rng.integers(0, 4))
        arr = np.array(w + [0] * pad, dtype=np.int64)
        if k % 17 == 0 and len(arr) > 2:  # interior zero -> AssertionError
            arr[int(rng.integers(0, max(1, n)))] = 0
        L = int(rng.integers(max(1, n - 3), n + pad + 3))
        for cyc in (False, True):
            for padded in (False, True):
                try:
                    res = R_utils.simplify_relator(arr.copy(), L, cyclical=cyc, padded=padded)
                    rows.append(dict(relator=jl(arr), L=L, cyclical=cyc, padded=padded, err=0, out=jl(res[0]), length=int(res[1])))
                except (AssertionError, ValueError, IndexError) as e:
                    rows.append(dict(relator=jl(arr), L=L, cyclical=cyc, padded=padded, err=ERR[type(e)], out=None, length=None))
    dump("simplify_fuzz.json", rows)


# ---------------------------------------------------------------------------------------------
def ms_pool_generator_order(max_n=7, max_w_len=7):
    pool = []
    for n in range(1, max_n + 1):
        d = R_ms.generate_miller_schupp_presentations(n, max_w_len)
        for lenw in range(1, max_w_len + 1):
            pool += d[lenw]
    return pool


def reembed(p, L):
    """copy the non-zero prefix of each half into width L (what change_max_relator_length does)"""
    return R_utils.change_max_relator_length_of_presentation(list(p), L)


def ms_pool():
    out = {"by_n": {}}
    for n in range(1, 8):
        d = R_ms.generate_miller_schupp_presentations(n, 7)
        out["by_n"][str(n)] = {str(k): v for k, v in d.items()}
    out["small"] = {}
    for n in (1, 2, 3):
        for mw in (1, 2, 3):
            d = R_ms.generate_miller_schupp_presentations(n, mw)
            out["small"][f"{n},{mw}"] = {str(k): v for k, v in d.items()}
    pool = ms_pool_generator_order()
    key = {tuple(p): k for k, p in enumerate(pool)}
    assert len(key) == 1190

    def order(fname):
        with open(os.path.join(REF, "ac_solver/search/miller_schupp/data", fname)) as f:
            return [key[tuple(ast.literal_eval(line))] for line in f if line.strip()]

    # data/*.txt orderings as indices into the generator-order pool (data, not source)
    out["all_presentations_order"] = order("all_presentations.txt")
    out["greedy_solved_order"] = order("greedy_solved_presentations.txt")
    out["bfs_solved_order"] = order("bfs_solved_presentations.txt")
    dump("ms_pool.json", out)


def greedy_paths():
    base = os.path.join(REF, "ac_solver/search/miller_schupp/data")
    pool = ms_pool_generator_order()
    key = {tuple(p): k for k, p in enumerate(pool)}
    rows = []
    with open(os.path.join(base, "greedy_solved_presentations.txt")) as fp, open(os.path.join(base, "greedy_search_paths.txt")) as fq:
        for lp, lq in zip(fp, fq):
            pres, path = ast.literal_eval(lp), ast.literal_eval(lq)
            # file convention: 1-based actions, root (0, len)  ->  today's (action, len) with root (-1, len)
            rows.append(dict(pool_index=key[tuple(pres)], path=[[a - 1, l] for a, l in path]))
    dump("greedy_paths_1e6.json", dict(budget=10**6, cyclical=False, note="paths re-encoded to 0-based actions, root -1", rows=rows))


def stable_ak3():
    nb = json.load(open(os.path.join(REF, "notebooks/Stable-AK3.ipynb")))
    src = "".join(nb["cells"][1]["source"])
    tree = ast.parse(src)
    vals = {}
    for node in tree.body:
        if isinstance(node, ast.Assign) and isinstance(node.targets[0], ast.Name) and node.targets[0].id in ("relator1", "relator2", "max_length", "sequence"):
            vals[node.targets[0].id] = ast.literal_eval(node.value)
    printed = "".join(nb["cells"][1]["outputs"][0]["text"])
    end = [int(t) for t in printed.split("[", 1)[1].replace("]", " ").split()]
    state = R_utils.convert_relators_to_presentation(vals["relator1"], vals["relator2"], vals["max_length"])
    lens = [13, 12]
    for m in vals["sequence"]:
        state, lens = R_moves.ACMove(m - 1, state, vals["max_length"], lens, cyclical=False)
    assert state.tolist() == end
    dump("stable_ak3.json", dict(relator1=vals["relator1"], relator2=vals["relator2"], L=vals["max_length"],
                                 sequence_one_based=vals["sequence"], cyclical=False, end_state=end, end_lengths=jl(lens)))


# ---------------------------------------------------------------------------------------------
def env_traj():
    L, N, T = 25, 1024, 128
    pool = [reembed(p, L) for p in ms_pool_generator_order()]
    tape = np.random.default_rng(0).integers(0, 12, size=(1000, 65536), dtype=np.uint8)[:T, :N]
    rew = np.zeros((T, N), np.int32); done = np.zeros((T, N), np.uint8); trunc = np.zeros((T, N), np.uint8)
    keep_t = (0, 1, 7, 31, 127)
    snaps = {t: np.zeros((N, 2 * L), np.int8) for t in keep_t}
    t0 = time.time()
    for e in range(N):
        env = ACEnv(ACEnvConfig(initial_state=np.array(pool[e % len(pool)]), horizon_length=100))
        for t in range(T):
            s, r, d, tr, _ = env.step(int(tape[t, e]))
            rew[t, e], done[t, e], trunc[t, e] = r, d, tr
            if t in snaps:
                snaps[t][e] = s
    print(f"env_traj main: {time.time() - t0:.1f}s")
    out = dict(L=L, horizon=100, tape=tape, reward=rew, done=done, truncated=trunc,
               init=np.stack([np.asarray(pool[e % len(pool)], np.int8) for e in range(N)]))
    for t, s in snaps.items():
        out[f"state_t{t}"] = s
    np.savez_compressed(os.path.join(OUT, "env_traj.npz"), **out)
    print(f"wrote env_traj.npz: {os.path.getsize(os.path.join(OUT, 'env_traj.npz')) / 1024:.1f} KiB")

    # scripted episodes: a done episode (info['actions']), a truncated one, stepping past done, reset(options)
    ak2 = [1, 1, -2, -2, -2, 0, 0, 1, 2, 1, -2, -1, -2, 0]
    ok, path = R_greedy(ak2, 10**5, cyclically_reduce_after_moves=True)
    assert ok
    acts = [a for a, _ in path[1:]]
    eps = []
    env = ACEnv(ACEnvConfig(initial_state=ak2, horizon_length=len(acts) + 3))
    steps = []
    for a in acts + [0, 5, 7, 1]:  # keeps stepping after done; crosses the horizon
        s, r, d, tr, info = env.step(a)
        steps.append(dict(action=a, state=jl(s), reward=int(r), done=bool(d), truncated=bool(tr), info=jl(info), lengths=jl(env.lengths)))
    eps.append(dict(name="ak2_done_then_continue", initial_state=ak2, horizon=len(acts) + 3, max_reward=env.max_reward, steps=steps))
    s0, info0 = env.reset(options={"starting_state": np.array([1, 2, 0, 0, 0, 0, 0, -2, 1, 0, 0, 0, 0, 0])})
    steps = []
    for a in (3, 3, 0, 9):
        s, r, d, tr, info = env.step(a)
        steps.append(dict(action=a, state=jl(s), reward=int(r), done=bool(d), truncated=bool(tr), info=jl(info), lengths=jl(env.lengths)))
    eps.append(dict(name="after_reset_with_starting_state", initial_state=jl(s0), horizon=len(acts) + 3, max_reward=env.max_reward,
                    count_steps_after_reset=0, steps=steps))
    env = ACEnv(ACEnvConfig(initial_state=ak2, horizon_length=4))
    steps = []
    for a in (0, 1, 2, 3, 4, 5):
        s, r, d, tr, info = env.step(a)
        steps.append(dict(action=a, state=jl(s), reward=int(r), done=bool(d), truncated=bool(tr), info=jl(info), lengths=jl(env.lengths)))
    eps.append(dict(name="truncation_h4", initial_state=ak2, horizon=4, max_reward=env.max_reward, steps=steps))
    # unreduced initial state is kept as is by ACEnvConfig
    unred = [1, -1, 2, 2, 0, 0, 2, 1, -1, -2, 1, 0]
    env = ACEnv(ACEnvConfig(initial_state=unred, horizon_length=10))
    steps = []
    for a in (4, 0, 7):
        s, r, d, tr, info = env.step(a)
        steps.append(dict(action=a, state=jl(s), reward=int(r), done=bool(d), truncated=bool(tr), info=jl(info), lengths=jl(env.lengths)))
    eps.append(dict(name="unreduced_initial", initial_state=unred, horizon=10, max_reward=env.max_reward,
                    initial_lengths=[4, 5], steps=steps))
    cfg = dict(default_initial_state=jl(ACEnvConfig().initial_state), default_horizon=ACEnvConfig().horizon_length,
               max_reward_h1000_L25=ACEnv(ACEnvConfig(initial_state=pool[0])).max_reward,
               obs_low=-2, obs_high=2, n_actions=ACEnv().action_space.n)
    dump("env_episodes.json", dict(config=cfg, episodes=eps))


# ---------------------------------------------------------------------------------------------
def search(skip_slow):
    ak2 = [1, 1, -2, -2, -2, 0, 0, 1, 2, 1, -2, -1, -2, 0]
    ak3_25 = R_utils.convert_relators_to_presentation([1, 1, 1, -2, -2, -2, -2], [1, 2, 1, -2, -1, -2], 25).tolist()
    pool = ms_pool_generator_order()
    rows = []

    def run(tag, algo, pres, budget, cyc):
        fn = R_bfs if algo == "bfs" else R_greedy
        t0 = time.time()
        ok, path = fn(presentation=np.array(pres), max_nodes_to_explore=budget, verbose=False, cyclically_reduce_after_moves=cyc)
        rows.append(dict(tag=tag, algo=algo, presentation=jl(pres), budget=budget, cyclical=cyc, solved=bool(ok),
                         path=jl(path) if path is not None else None))
        print(f"  {tag} {algo} budget={budget} cyc={cyc}: solved={ok} len={len(path) if path else None} {time.time() - t0:.1f}s", flush=True)

    # the reference's own golden cases (tests/search/test_bfs.py, test_gs.py)
    run("ak2", "bfs", ak2, 10**6, False)
    run("ak2", "bfs", ak2, 10, False)
    run("ak2", "greedy", ak2, 10**6, False)
    run("ak2", "greedy", ak2, 10000, False)  # BASELINE config 1
    for b in (1, 2, 13, 14, 100, 1000):
        run("ak2", "bfs", ak2, b, False)
        run("ak2", "greedy", ak2, b, False)
    run("ak2", "bfs", ak2, 10**5, True)
    run("ak2", "greedy", ak2, 10**5, True)
    # tests/search/miller_schupp/test_miller_schupp.py ranges: n,w in {1,2} and {3,4}
    for n in (1, 2):
        d = R_ms.generate_miller_schupp_presentations(n, 2)
        for w in (1, 2):
            for p in d[w]:
                run(f"ms_n{n}_w{w}", "greedy", p, 10**6, False)
                run(f"ms_n{n}_w{w}", "bfs", p, 10**4, False)
    for n in (3, 4):
        d = R_ms.generate_miller_schupp_presentations(n, 4)
        for w in (3, 4):
            for p in d[w]:
                run(f"ms_n{n}_w{w}", "greedy", p, 10**4, False)
    # MS pool sample at native L and re-embedded at L=25, both algorithms, both cyclical values
    idx = list(range(0, 1190, 29))
    for k in idx:
        for b in (300, 3000):
            run(f"pool{k}", "bfs", pool[k], b, False)
            run(f"pool{k}", "greedy", pool[k], b, False)
        p25 = reembed(pool[k], 25).tolist()
        run(f"pool{k}@25", "bfs", p25, 1000, bool(k % 2))
        run(f"pool{k}@25", "greedy", p25, 1000, bool(k % 2))
    # BASELINE config 3: AK(3) @ L=25
    for b in (10**3, 10**4) + (() if skip_slow else (10**5,)):
        run("ak3@25", "greedy", ak3_25, b, False)
    run("ak3@25", "bfs", ak3_25, 10**4, False)
    # inputs the searches accept but that are not reduced (bfs keeps the raw initial tuple)
    run("unreduced", "bfs", [1, -1, 2, 2, 0, 0, 2, 1, -1, -2, 1, 0], 500, False)
    run("unreduced", "greedy", [1, -1, 2, 2, 0, 0, 2, 1, -1, -2, 1, 0], 500, False)
    dump("search.json", rows)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-slow", action="store_true")
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    jobs = dict(unit=unit_tables, fuzz=acmove_fuzz, simplify=simplify_fuzz, ms=ms_pool, paths=greedy_paths, ak3=stable_ak3,
                env=env_traj, search=lambda: search(a.skip_slow))
    for name, fn in jobs.items():
        if a.only and name not in a.only.split(","):
            continue
        t0 = time.time()
        fn()
        print(f"[{name}] {time.time() - t0:.1f}s", flush=True)
