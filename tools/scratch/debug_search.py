#!/usr/bin/env python3
"""Runs the search fixtures one by one with progress output (debug aid)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "ac-solver_amd"), ROOT]
import faulthandler

faulthandler.dump_traceback_later(60, exit=True)
import ac_solver  # noqa: E402

rows = json.load(open(os.path.join(ROOT, "tests/golden/search.json")))
bad = 0
for i, r in enumerate(rows):
    fn = ac_solver.bfs if r["algo"] == "bfs" else ac_solver.greedy_search
    print(i, r["tag"], r["algo"], r["budget"], r["cyclical"], "L=", len(r["presentation"]) // 2, end=" ... ", flush=True)
    t0 = time.time()
    faulthandler.cancel_dump_traceback_later()
    faulthandler.dump_traceback_later(40, exit=True)
    ok, path = fn(r["presentation"], r["budget"], cyclically_reduce_after_moves=r["cyclical"])
    want = None if r["path"] is None else [tuple(x) for x in r["path"]]
    good = ok == r["solved"] and path == want
    bad += not good
    print("OK" if good else f"MISMATCH got {ok} {path and path[-3:]} want {r['solved']} {want and want[-3:]}", f"{time.time() - t0:.2f}s", flush=True)
print("mismatches:", bad)
