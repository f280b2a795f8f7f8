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


_SHARDED = {"comm": None, "min_nodes": 0}


def shard_over_process_group(group=True, min_nodes=10**6):
    """Make `bfs` a COLLECTIVE call: from now on every `bfs(...)` with a budget of at least `min_nodes` runs with its frontier sharded over
    the GPUs of a torch.distributed process group (backend "nccl" == RCCL over xGMI) -- acx_bfs_sharded, one C call per rank on the
    group's own RCCL communicator --, same signature, same result on every rank as the single-GPU search.  `group`: True = the default
    group, a ProcessGroup, a sharded.NativeComm, or None / False to switch it off again.

    Deliberately a switch and not automatic: the reference's `bfs` (breadth_first.py:15) is an ordinary function, and scripts that run
    under torchrun call it with DIFFERENT presentations on different ranks (trivialize_miller_schupp_through_search dealt `rank::world`).
    A collective needs every rank to make the same call; whoever turns this on says that they do."""
    from ac_solver.search.sharded import NativeComm

    old = _SHARDED["comm"]
    if not group:
        _SHARDED["comm"] = None
    elif isinstance(group, NativeComm):
        _SHARDED["comm"] = group
    else:
        _SHARDED["comm"] = NativeComm.from_process_group(None if group is True else group)
    _SHARDED["min_nodes"] = int(min_nodes)
    return old


def bfs(presentation, max_nodes_to_explore=10000, verbose=False, cyclically_reduce_after_moves=False):
    """Returns (is_search_successful, path); path = [(-1, len0), (action, total_length), ...] or None."""
    assert is_array_valid_presentation(presentation), f"{presentation} is not a valid presentation"
    comm = _SHARDED["comm"]
    if comm is not None and comm.world > 1 and not verbose and max_nodes_to_explore >= _SHARDED["min_nodes"]:
        # (verbose searches print the reference's "New minimal length found" lines, which only the single-GPU frontier records)
        from ac_solver.search.sharded import bfs_sharded_native

        solved, path, stats = bfs_sharded_native(presentation, max_nodes_to_explore, False, cyclically_reduce_after_moves, comm=comm, want_stats=True)
        if not solved and stats["nodes"] >= max_nodes_to_explore:
            print(f"Exiting search as number of explored nodes = {stats['nodes']} has exceeded the limit {max_nodes_to_explore}")
        return (True, path) if solved else (False, None)
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
