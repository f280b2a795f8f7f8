#!/usr/bin/env python3
"""where the host time of the greedy sweep goes: the C call (acx_search_groups) against the conversion of its outputs (_collect)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "ac-solver_amd"), ROOT]
import numpy as np
from ac_solver import _acx
from ac_solver.search import _common
from ac_solver.search.miller_schupp.miller_schupp import generate_miller_schupp_presentations
groups = []
for n in range(1, 8):
    d = generate_miller_schupp_presentations(n, 7)
    groups.append(np.array([q for w in range(1, 8) for q in d[w]], dtype=np.int8))
orig_collect = _common._collect
orig_call = _acx.lib.acx_search_groups
T = {}
def collect(*a):
    t0 = time.perf_counter(); r = orig_collect(*a); T["collect"] = time.perf_counter() - t0; return r
_common._collect = collect
class LibProxy:
    def __init__(self, lib): self._lib = lib
    def __getattr__(self, k):
        f = getattr(self._lib, k)
        if k != "acx_search_groups": return f
        def g(*a):
            t0 = time.perf_counter(); r = f(*a); T["c_call"] = time.perf_counter() - t0; return r
        return g
_acx.lib = LibProxy(_acx.lib)
kind = _acx.SEARCH_GREEDY if (len(sys.argv) < 2 or sys.argv[1] == "greedy") else _acx.SEARCH_BFS
for rep in range(4):
    t0 = time.perf_counter()
    res = _common.run_search_groups(kind, groups, 10**6, False)
    tot = time.perf_counter() - t0
    lens = [len(p) for r in res for ok, p, st in r if p]
    print(f"call {rep}: {tot*1e3:.1f} ms, of which the C call {T['c_call']*1e3:.1f} ms, _collect {T['collect']*1e3:.1f} ms; {len(lens)} paths, {sum(lens)} entries, longest {max(lens)}", flush=True)
