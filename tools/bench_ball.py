#!/usr/bin/env python3
"""Radius-5 neighbourhood sizes of all 1190 Miller-Schupp presentations (the reference's barcode_analysis/5_steps_neibourhoods
run), both move sets, on the GPU; the C oracle timed on a sample for context."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "ac-solver_amd"), ROOT]
from ac_solver.barcode import neighbourhood_sizes
from oracle import ac_oracle as O

g = json.load(open(os.path.join(ROOT, "tests/golden/ms_pool.json")))
radius = int(sys.argv[1]) if len(sys.argv) > 1 else 5
neighbourhood_sizes(g["by_n"]["1"]["1"][:2], 2)  # warm-up
for classic in (False, True):
    t0 = time.perf_counter()
    total = 0
    for n in range(1, 8):
        rows = [p for w in range(1, 8) for p in g["by_n"][str(n)][str(w)]]
        sizes = neighbourhood_sizes(rows, radius, classic)
        total += sum(sizes)
    dt = time.perf_counter() - t0
    t1 = time.perf_counter()
    sample = [g["by_n"][str(n)]["7"][0] for n in range(1, 8)]
    osum = sum(O.ball_size(r, radius, classic) for r in sample)
    dto = time.perf_counter() - t1
    print(f"{'classic' if classic else 'prime'} moves, radius {radius}: 1190 balls, {total} nodes in {dt:.2f}s = {total / dt:.3e} nodes/s "
          f"({12 * total / dt if not classic else 14 * total / dt:.3e} moves/s); C oracle on 7 of them: {osum / dto:.3e} nodes/s on one core")
