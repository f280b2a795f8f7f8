#!/usr/bin/env python3
"""experiment: the bfs sweep over the 1190 Miller-Schupp presentations (seven max_relator_lengths) with K width groups in flight at a time
(host threads, one acx_search_many each) instead of one after the other"""
import os, sys, time, threading, queue
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "ac-solver_amd"), ROOT]
import numpy as np
import torch
from ac_solver import _acx
from ac_solver.search._common import run_search_many, run_search_groups
from ac_solver.search.miller_schupp.miller_schupp import generate_miller_schupp_presentations

groups = []
for n in range(1, 8):
    d = generate_miller_schupp_presentations(n, 7)
    groups.append(np.array([q for w in range(1, 8) for q in d[w]], dtype=np.int8))
kind = _acx.SEARCH_BFS
for order_name, order in (("n = 1..7", list(range(7))), ("n = 7..1", list(range(6, -1, -1)))):
    for K in (1, 2, 3):
        times = []
        solved = 0
        for rep in range(4):
            res = [None] * 7
            q = queue.Queue()
            for g in order:
                q.put(g)
            def work():
                while True:
                    try:
                        g = q.get_nowait()
                    except queue.Empty:
                        return
                    res[g] = run_search_many(kind, groups[g], 10**6, True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            th = [threading.Thread(target=work) for _ in range(K)]
            [t.start() for t in th]
            [t.join() for t in th]
            times.append(time.perf_counter() - t0)
            solved = sum(ok for r in res for ok, _, _ in r)
        print(f"order {order_name}, {K} group(s) in flight: {sorted(times)[1]:.4f} s (samples {[round(t, 4) for t in times]}), solved {solved}", flush=True)
t = []
for rep in range(4):
    t0 = time.perf_counter()
    run_search_groups(kind, groups, 10**6, True)
    t.append(time.perf_counter() - t0)
print("run_search_groups:", [round(x, 4) for x in t])
