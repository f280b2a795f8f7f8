"""The package the reference reads its published Miller-Schupp data files from (`ac_solver/search/miller_schupp/data/*.txt`, through
`importlib.resources`: `agents/utils.py:28`, `tests/search/miller_schupp/data/test_do_files_exist.py`).

The files are not shipped here: importing this package produces them once, with this build's own searches on the GPU (both sweeps over
the 1190 presentations: well under a second of device time), into this directory -- `all_presentations.txt`,
`greedy_solved_presentations.txt`, `greedy_search_paths.txt`, `bfs_solved_presentations.txt`, identical to the published ones
(tests/test_gpu_ppo.py).  Without a GPU the import succeeds and the directory stays empty."""
import os

from ac_solver.search.miller_schupp.data_files import DATA_DIR, FILES


def _ensure():
    if all(os.path.exists(os.path.join(DATA_DIR, name)) for name in FILES):
        return
    from ac_solver import _acx

    if _acx.device_count() > 0:
        from ac_solver.search.miller_schupp.data_files import make_data_files

        make_data_files(verbose=False)


_ensure()
