"""Full-size GPU checks against the reference's PUBLISHED results (ac_solver/search/miller_schupp/data/*.txt,
carried here as index lists in tests/golden/ms_pool.json and as re-encoded paths in greedy_paths_1e6.json):

* greedy_search, budget 1e6, on all 1190 Miller-Schupp presentations: exactly the 533 published ones are
  solved and all 533 published paths are reproduced;
* bfs, budget 1e6, cyclically_reduce_after_moves=True: exactly the 278 presentations of
  bfs_solved_presentations.txt are solved (with cyclical=False the reference algorithm solves a strict subset;
  the published file can only have been produced with cyclic reduction on -- established here, SURVEY left it open).
"""
import numpy as np
import pytest

from tests.conftest import ms_pool_generator_order

pytestmark = pytest.mark.gpu


def _sweep(kind, pool, budget, cyclical):
    from ac_solver.search._common import run_search_many

    out = {}
    for n in range(7):
        rows = np.array(pool[n * 170:(n + 1) * 170], dtype=np.int8)
        for k, (ok, path, st) in enumerate(run_search_many(kind, rows, budget, cyclical, n_threads=16)):
            if ok:
                out[n * 170 + k] = path
    return out


@pytest.mark.timeout(600)
def test_greedy_sweep_reproduces_published_533(golden_json):
    from ac_solver import _acx

    _acx.require_device()
    g = golden_json("ms_pool.json")
    paths = _sweep(_acx.SEARCH_GREEDY, ms_pool_generator_order(g), 10**6, False)
    assert sorted(paths) == sorted(g["greedy_solved_order"]) and len(paths) == 533
    gp = golden_json("greedy_paths_1e6.json")
    for row in gp["rows"]:
        assert paths[row["pool_index"]] == [tuple(x) for x in row["path"]], row["pool_index"]


@pytest.mark.timeout(600)
def test_bfs_sweep_reproduces_published_278(golden_json):
    from ac_solver import _acx

    _acx.require_device()
    g = golden_json("ms_pool.json")
    paths = _sweep(_acx.SEARCH_BFS, ms_pool_generator_order(g), 10**6, True)
    assert sorted(paths) == sorted(g["bfs_solved_order"]) and len(paths) == 278
