#!/usr/bin/env python3
"""BASELINE config 4 shape: bfs / greedy_search over all 1190 Miller-Schupp presentations (native max_relator_length
of each n), overlapped on one GPU, checked against the reference's published results (data/*.txt as index fixtures)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "ac-solver_amd"), ROOT]
import numpy as np

from ac_solver import _acx
from ac_solver.search._common import run_search_groups, run_search_many

g = json.load(open(os.path.join(ROOT, "tests/golden/ms_pool.json")))
pool = []
for n in range(1, 8):
    for w in range(1, 8):
        pool += g["by_n"][str(n)][str(w)]
algo = sys.argv[1] if len(sys.argv) > 1 else "bfs"
budget = int(float(sys.argv[2])) if len(sys.argv) > 2 else 10**6
threads = int(sys.argv[3]) if len(sys.argv) > 3 else 16
cyc = bool(int(sys.argv[4])) if len(sys.argv) > 4 else False
kind = _acx.SEARCH_BFS if algo == "bfs" else _acx.SEARCH_GREEDY
together = len(sys.argv) > 5 and sys.argv[5] == "together"  # all seven widths at once (one host thread per n)
t0 = time.perf_counter()
solved_idx, nodes, paths = [], 0, {}
groups = [np.array(pool[(n - 1) * 170:n * 170], dtype=np.int8) for n in range(1, 8)]
if together:
    all_res = run_search_groups(kind, groups, budget, cyc, n_threads=threads)
for n in range(1, 8):
    rows = groups[n - 1]
    t1 = time.perf_counter()
    res = all_res[n - 1] if together else run_search_many(kind, rows, budget, cyc, n_threads=threads)
    for k, (ok, path, st) in enumerate(res):
        nodes += st["nodes"]
        if ok:
            solved_idx.append((n - 1) * 170 + k)
            paths[(n - 1) * 170 + k] = path
    print(f"n={n} L={rows.shape[1] // 2}: {sum(r[0] for r in res)} solved of 170 in {time.perf_counter() - t1:.2f}s", flush=True)
dt = time.perf_counter() - t0
print(f"{algo} budget={budget} cyclical={cyc}: {len(solved_idx)} / 1190 solved, {nodes} nodes in {dt:.2f}s = {nodes / dt:.3e} nodes/s, {1190 / dt:.1f} searches/s")
want = sorted(g["bfs_solved_order"] if algo == "bfs" else g["greedy_solved_order"])
print("solved set equals the reference's published set:", sorted(solved_idx) == want, f"(published {len(want)})")
if algo == "greedy":
    gp = json.load(open(os.path.join(ROOT, "tests/golden/greedy_paths_1e6.json")))
    same = sum(paths.get(r["pool_index"]) == [tuple(x) for x in r["path"]] for r in gp["rows"])
    print(f"paths identical to data/greedy_search_paths.txt: {same} / {len(gp['rows'])}")
