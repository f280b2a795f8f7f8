"""Greedy (best-first) search on the AC graph -- drop-in for ac_solver/search/greedy.py.

Same signature and result as the reference `greedy_search` (greedy.py:15-121): nodes are expanded in
increasing (total length, depth, state tuple) order, success is checked before deduplication, the
budget once per expanded node, and an unsuccessful search returns
`(False, path_of_the_last_expanded_node + [(11, length_of_its_last_child)])` exactly as the reference does.
"""
import numpy as np

from ac_solver import _acx
from ac_solver.search._common import run_search


def greedy_search(presentation, max_nodes_to_explore=10000, verbose=False, cyclically_reduce_after_moves=False):
    """Returns (is_search_successful, path); path = [(-1, len0), (action, total_length), ...]."""
    presentation = np.array(presentation, dtype=np.int8)
    solved, path, stats = run_search(_acx.SEARCH_GREEDY, presentation, max_nodes_to_explore, cyclically_reduce_after_moves,
                                     verbose=verbose)  # verbose: the per-improvement lines (greedy.py:85-89)
    if solved:
        if verbose:  # greedy.py:92-101
            from ac_solver.envs.ac_moves import ACMove

            L = len(presentation) // 2
            state, lengths = presentation, [int(np.count_nonzero(presentation[:L])), int(np.count_nonzero(presentation[L:]))]
            for action, _ in path[1:]:
                state, lengths = ACMove(action, state, L, lengths, cyclical=cyclically_reduce_after_moves)
            print(f"Found {state[0:1], state[L:L + 1]} after exploring {stats['expanded']} nodes")
            print(f"Path to a trivial state: (tuples are of form (action, length of a state)) {path}")
            print(f"Total path length: {len(path)}")
        return True, path
    if stats["nodes"] >= max_nodes_to_explore:
        print(f"Exiting search as number of explored nodes = {stats['nodes']} has exceeded the limit {max_nodes_to_explore}")
    return False, path



if __name__ == "__main__":  # the reference module's self-check (greedy.py:124-150)
    from ac_solver.search._common import self_check

    self_check(greedy_search)
