"""single greedy_search (chained hand-off cycle) on random Miller-Schupp presentations at budgets with many hand-offs, against the C oracle"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "ac-solver_amd"), ROOT]
import numpy as np
from ac_solver import _acx
from ac_solver.search._common import run_search
from oracle import ac_oracle as O
g = json.load(open(os.path.join(ROOT, "tests/golden/ms_pool.json")))
pool = [p for n in range(1, 8) for w in range(1, 8) for p in g["by_n"][str(n)][str(w)]]
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
for k in rng.choice(len(pool), size=int(sys.argv[2]) if len(sys.argv) > 2 else 24, replace=False):
    budget = int(rng.choice([200000, 500000, 1000000])); cyc = bool(rng.integers(0, 2))
    hm = str(int(rng.choice([64, 512, 512, 1024])))
    os.environ["ACX_GREEDY_HAND_MIN"] = hm
    ok, path, st = run_search(_acx.SEARCH_GREEDY, np.array(pool[k], dtype=np.int8), budget, cyc)
    wok, wpath, wst = O.greedy_search(pool[k], budget, cyclically_reduce_after_moves=cyc, stats=True)
    same = (ok, path) == (wok, wpath) and st["nodes"] == wst["nodes"] and st["expanded"] == wst["expanded"]
    bad += not same
    print(int(k), budget, cyc, hm, ok, st["nodes"], st["levels"], "ok" if same else "MISMATCH", flush=True)
print("mismatches:", bad)
