"""Breadth-first search on the AC graph -- drop-in for ac_solver/search/breadth_first.py.

Same signature and result as the reference `bfs` (breadth_first.py:15-97): FIFO order over parents,
actions 0..11 per parent, success as soon as a child of total length 2 is generated (checked before
deduplication), exact visited set, budget tested once per expanded parent, `(False, None)` when the
budget runs out.  The frontier, the hash set and the moves run on the GPU (libacx `acx_search`).
"""
import numpy as np

from ac_solver import _acx
from ac_solver.envs.utils import is_array_valid_presentation
from ac_solver.search._common import run_search


def bfs(presentation, max_nodes_to_explore=10000, verbose=False, cyclically_reduce_after_moves=False):
    """Returns (is_search_successful, path); path = [(-1, len0), (action, total_length), ...] or None."""
    assert is_array_valid_presentation(presentation), f"{presentation} is not a valid presentation"
    solved, path, stats = run_search(_acx.SEARCH_BFS, np.array(presentation, dtype=np.int8), max_nodes_to_explore,
                                     cyclically_reduce_after_moves, verbose=verbose)  # verbose: the per-improvement lines (:79-82)
    if not solved:
        if stats["nodes"] >= max_nodes_to_explore:
            print(f"Exiting search as number of explored nodes = {stats['nodes']} has exceeded the limit {max_nodes_to_explore}")
        return False, None
    return True, path



if __name__ == "__main__":  # the reference module's self-check (breadth_first.py:100-126)
    from ac_solver.search._common import self_check

    self_check(bfs)
