#!/usr/bin/env python3
"""tests/golden/verbose_lines.json: what the reference PRINTS with verbose=True (build container only).

    python oracle/tools/make_verbose_golden.py

Imports the reference's bfs / greedy_search (same bootstrap as make_golden.py), runs a few searches with verbose=True and
stores the captured stdout lines next to the inputs.  Data only."""
import contextlib
import io
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
sys.path.insert(0, os.path.join(HERE, "stubs"))

from ac_solver.search.breadth_first import bfs as R_bfs  # noqa: E402
from ac_solver.search.greedy import greedy_search as R_greedy  # noqa: E402

AK2 = [1, 1, -2, -2, -2, 0, 0, 1, 2, 1, -2, -1, -2, 0]
pool = json.load(open(os.path.join(REPO, "tests", "golden", "ms_pool.json")))
ms = [pool["by_n"]["2"]["3"][0], pool["by_n"]["3"]["4"][1], pool["by_n"]["1"]["5"][0]]
AK3 = [1, 1, 1, -2, -2, -2, -2] + [0] * 6 + [1, 2, 1, -2, -1, -2] + [0] * 7
cases = [("bfs", AK2, 4000, False), ("bfs", AK2, 300, True), ("greedy", AK2, 10000, False), ("greedy", AK2, 50, False), ("greedy", AK3, 3000, False),
         ("bfs", AK3, 3000, True)] + [(a, p, 3000, c) for p in ms for a, c in (("bfs", False), ("greedy", True))]
rows = []
for algo, p, budget, cyc in cases:
    fn = R_bfs if algo == "bfs" else R_greedy
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        ok, path = fn(np.array(p, dtype=np.int8), max_nodes_to_explore=budget, verbose=True, cyclically_reduce_after_moves=cyc)
    rows.append({"algo": algo, "presentation": list(map(int, p)), "budget": budget, "cyclical": cyc, "solved": bool(ok), "lines": buf.getvalue().splitlines()})
    print(algo, budget, cyc, ok, len(rows[-1]["lines"]), "lines")
json.dump(rows, open(os.path.join(REPO, "tests", "golden", "verbose_lines.json"), "w"), indent=0)
