"""The package the reference reads its published Miller-Schupp data files from (`ac_solver/search/miller_schupp/data/*.txt`, through
`importlib.resources`: `agents/utils.py:28`, `tests/search/miller_schupp/data/test_do_files_exist.py`).

The files are not shipped here, and importing this package does NOT touch the GPU (an import under `torchrun` happens in every rank,
before any launcher decision): `ensure()` -- or the trainer's own `load_initial_states_from_text_file` -- produces them on first use
with this build's searches (both sweeps over the 1190 presentations: well under a second of device time), once per directory however
many ranks ask at the same time (`data_files.ensure_data_files`: directory lock + rename into place).  The results are identical to
the published files (tests/test_gpu_ppo.py)."""
from ac_solver.search.miller_schupp.data_files import DATA_DIR, FILES, ensure_data_files  # noqa: F401


def ensure():
    """`all_presentations.txt`, `greedy_solved_presentations.txt`, `greedy_search_paths.txt`, `bfs_solved_presentations.txt` exist in
    this package's directory after the call (needs a GPU the first time)."""
    return ensure_data_files()
