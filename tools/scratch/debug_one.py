import json, os, sys, time, faulthandler
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "ac-solver_amd"), ROOT]
faulthandler.dump_traceback_later(25, exit=True)
import ac_solver
rows = json.load(open(os.path.join(ROOT, "tests/golden/search.json")))
i = int(sys.argv[1])
r = rows[i]
fn = ac_solver.bfs if r["algo"] == "bfs" else ac_solver.greedy_search
print(r["tag"], r["algo"], r["budget"], r["presentation"], flush=True)
ok, path = fn(r["presentation"], r["budget"], cyclically_reduce_after_moves=r["cyclical"])
print(ok, path == (None if r["path"] is None else [tuple(x) for x in r["path"]]))
