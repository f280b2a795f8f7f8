"""each Miller-Schupp width alone (warm), then all together: where the sweep's time goes"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "ac-solver_amd"), ROOT]
import numpy as np
from ac_solver import _acx
from ac_solver.search._common import run_search_groups, run_search_many, run_search
g = json.load(open(os.path.join(ROOT, "tests/golden/ms_pool.json")))
budget = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10**6
groups = [np.array([p for w in range(1, 8) for p in g["by_n"][str(n)][str(w)]], dtype=np.int8) for n in range(1, 8)]
run_search_groups(_acx.SEARCH_BFS, groups, budget, True)
tot = 0
for n, grp in enumerate(groups, 1):
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        res = run_search_many(_acx.SEARCH_BFS, grp, budget, True)
        best = min(best, time.perf_counter() - t0)
    nodes = sum(st["nodes"] for _, _, st in res)
    ch = sum(st["children"] for _, _, st in res)
    lv = max(st["levels"] for _, _, st in res)
    tot += best
    print(f"n={n} L={grp.shape[1]//2}: {best*1e3:.1f} ms alone; nodes {nodes:.3e} children {ch:.3e} max batches {lv}; {nodes/best:.3e} nodes/s, {ch/best:.3e} children/s", flush=True)
print(f"sum of groups alone {tot*1e3:.1f} ms")
for rep in range(3):
    t0 = time.perf_counter()
    run_search_groups(_acx.SEARCH_BFS, groups, budget, True)
    print(f"together {1e3*(time.perf_counter()-t0):.1f} ms")
# one unsolved presentation of each width as a single big search
for n, grp in enumerate(groups, 1):
    ok, _, st = run_search(_acx.SEARCH_BFS, grp[-1], 3 * 10**7, True)
    ok, _, st = run_search(_acx.SEARCH_BFS, grp[-1], 3 * 10**7, True)
    print(f"single n={n}: solved {ok} nodes {st['nodes']:.3e} children {st['children']:.3e} in {st['seconds']*1e3:.2f} ms: {st['children']/st['seconds']:.3e} children/s")
