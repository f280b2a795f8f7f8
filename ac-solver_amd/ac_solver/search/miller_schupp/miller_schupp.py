"""Miller-Schupp presentations and the batch search driver -- drop-in for
ac_solver/search/miller_schupp/miller_schupp.py.

MS(n, w) = <x, y | x^-1 y^n x = y^(n+1), x = w>, n >= 1, w a word with zero exponent sum on x.

`generate_miller_schupp_presentations` enumerates candidate words on the host, reduces all of them in
ONE batched launch of libacx's simplify kernel per word length, and then applies the reference's
first-seen-wins filter over cyclic rotations (reference: miller_schupp.py:20-83).

Example:
    python -u -m ac_solver.search.miller_schupp.miller_schupp --min-n=3 --max-n=5 --min-w-len=2 --max-w-len=3 --search-fn=greedy
"""
import argparse
import os

import numpy as np

from ac_solver import _acx

_LETTERS = np.array([1, 2, -1, -2], dtype=np.int8)  # itertools.product order used by the reference


def _words_in_product_order(length):
    """All 4**length words over (1, 2, -1, -2), rows ordered like itertools.product(..., repeat=length)."""
    idx = np.arange(4 ** length, dtype=np.int64)
    digits = (idx[:, None] // (4 ** np.arange(length - 1, -1, -1, dtype=np.int64))[None, :]) % 4
    return _LETTERS[digits]


def generate_miller_schupp_presentations(n, max_w_len):
    """dict: length(w) -> list of presentations (each a list of 2L ints), for fixed n and all words up to
    max_w_len, L = 2 * max(2n + 3, max_w_len + 1) + 2.  Two presentations whose second relator differs
    only by free/cyclic reduction of x^-1 w or by a cyclic rotation are kept once (the first met)."""
    assert n >= 1 and max_w_len >= 1, f"expect n >= 1 and max_w_len >=1 ; got n = {n}, max_w_len = {max_w_len}"
    L = 2 * max(2 * n + 3, max_w_len + 1) + 2
    relator1 = [-1] + [2] * n + [1] + [-2] * (n + 1) + [0] * (L - 2 * n - 3)

    seen = set()
    by_lenw = {}
    for length in range(1, max_w_len + 1):
        words = _words_in_product_order(length)
        x_sum = np.where(np.abs(words) == 1, words, 0).sum(axis=1)
        words = words[x_sum == 0]  # zero exponent sum on x
        if len(words) == 0:
            continue
        rows = np.concatenate([np.full((len(words), 1), -1, np.int8), words], axis=1)  # x^-1 w
        reduced, lens, err = _acx.simplify_rows(rows, cyclical=True)
        assert not err.any()
        for row, (k, _) in zip(reduced, lens):
            relator2 = [int(v) for v in row[:k]]
            if relator2 == [-1]:  # w reduced to the empty word
                continue
            key = tuple(relator2)
            if key in seen:
                continue
            for r in range(len(relator2)):
                seen.add(tuple(relator2[r:] + relator2[:r]))
            by_lenw.setdefault(len(relator2) - 1, []).append(relator1 + relator2 + [0] * (L - len(relator2)))
    return by_lenw


def write_list_to_text_file(list, filepath):
    if not filepath.endswith(".txt"):
        filepath = filepath + ".txt"
    with open(filepath, "w") as f:
        for element in list:
            f.write(f"{element}\n")


def trivialize_miller_schupp_through_search(min_n, max_n, min_w_len, max_w_len, max_nodes_to_explore, search_fn,
                                            write_output_to_file=False):
    """Run `search_fn` (greedy_search or bfs) on every MS(n, w) with n in [min_n, max_n] and reduced
    length(w) in [min_w_len, max_w_len].  Returns (solved_rels, unsolved_rels, solved_paths).
    Reference: miller_schupp.py:95-177."""
    assert search_fn.__name__ in ["greedy_search", "bfs"], f"expect search_fn to be greedy or bfs; got {search_fn.__name__}"
    rels = {n: generate_miller_schupp_presentations(n, max_w_len) for n in range(min_n, max_n + 1)}
    solved_rels, unsolved_rels, solved_paths = [], [], []

    from ac_solver.search import breadth_first, greedy
    from ac_solver.search._common import run_search_groups

    ours = {breadth_first.bfs: _acx.SEARCH_BFS, greedy.greedy_search: _acx.SEARCH_GREEDY}.get(search_fn)
    groups = {n: [(lenw, pres) for lenw in range(min_w_len, max_w_len + 1) for pres in rels[n].get(lenw, [])] for n in range(min_n, max_n + 1)}
    by_n = {}
    if ours is not None:
        # the searches of one n share max_relator_length and go to the GPU as one batch (acx_search_many), the batches of all n
        # are in flight together; results come back in the reference's order
        ns = [n for n in groups if groups[n]]
        res = run_search_groups(ours, [np.array([p for _, p in groups[n]], dtype=np.int8) for n in ns], max_nodes_to_explore, False)
        for n, r in zip(ns, res):
            by_n[n] = [(ok, path if (ok or ours == _acx.SEARCH_GREEDY) else None, st) for ok, path, st in r]
    for n in range(min_n, max_n + 1):
        results = by_n.get(n)
        k = 0
        for lenw in range(min_w_len, max_w_len + 1):
            print(f"Applying {search_fn.__name__} to presentations of n = {n}, lenw = {lenw}")
            for pres in rels[n].get(lenw, []):
                if results is not None:
                    solved, path, st = results[k]
                    if not solved and st["nodes"] >= max_nodes_to_explore:
                        print(f"Exiting search as number of explored nodes = {st['nodes']} has exceeded the limit {max_nodes_to_explore}")
                else:
                    solved, path = search_fn(presentation=pres, max_nodes_to_explore=max_nodes_to_explore, verbose=False,
                                             cyclically_reduce_after_moves=False)
                k += 1
                if solved:
                    solved_rels.append(pres)
                    solved_paths.append(path)
                else:
                    unsolved_rels.append(pres)

    if write_output_to_file:
        dirname = os.path.join(os.path.dirname(os.path.realpath(__file__)), "data")
        os.makedirs(dirname, exist_ok=True)
        base = f"n-{min_n}-to-{max_n}_lenw-{min_w_len}-to-{max_w_len}-max-nodes-{max_nodes_to_explore}-{search_fn.__name__}"
        write_list_to_text_file(list=solved_rels, filepath=os.path.join(dirname, base + "_solved"))
        write_list_to_text_file(list=unsolved_rels, filepath=os.path.join(dirname, base + "_unsolved"))
        write_list_to_text_file(list=solved_paths, filepath=os.path.join(dirname, base + "_paths"))
        print(f"saved output in {dirname} with filenames: {base}_solved, {base}_unsolved, {base}_paths")
    return solved_rels, unsolved_rels, solved_paths


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--min-n", type=int, default=1, help="Minimum value of label n of Miller-Schupp series")
    ap.add_argument("--max-n", type=int, default=7, help="Maximum value of label n of Miller-Schupp series")
    ap.add_argument("--min-w-len", type=int, default=1, help="Minimum word-length of w of Miller-Schupp series")
    ap.add_argument("--max-w-len", type=int, default=7, help="Maximum word-length of w of Miller-Schupp series")
    ap.add_argument("--max-nodes-to-explore", type=int, default=int(1e6), help="Maximum number of nodes to explore during tree search.")
    ap.add_argument("--search-fn", type=str, default="greedy", help="the name of the search function; greedy or bfs")
    args = ap.parse_args()
    assert args.search_fn in ["greedy", "bfs"], f"expect search-algorithm to be greedy or bfs; got {args.search_fn}"
    assert args.min_n <= args.max_n, "min_n cannot be greater than max_n"
    assert args.min_w_len <= args.max_w_len, "min_w_len cannot be greater than max_w_len"
    if args.search_fn == "greedy":
        from ac_solver.search.greedy import greedy_search as search_fn
    else:
        from ac_solver.search.breadth_first import bfs as search_fn
    trivialize_miller_schupp_through_search(min_n=args.min_n, max_n=args.max_n, min_w_len=args.min_w_len, max_w_len=args.max_w_len,
                                            max_nodes_to_explore=args.max_nodes_to_explore, search_fn=search_fn, write_output_to_file=True)
