"""Shared driver for bfs / greedy_search: hands the presentation to libacx's device frontier."""
import ctypes as C
import os
import time

import numpy as np

from ac_solver import _acx


def run_search(kind, presentation, max_nodes_to_explore, cyclical, want_stats=False, verbose=False):
    """`verbose`: print "New minimal length found: l" for every child that is shorter than everything generated before it, as
    the reference does while it searches (breadth_first.py:79-82, greedy.py:85-89); here the lines come out when the
    search returns, in the reference's order."""
    _acx.require_device()
    if verbose:
        _acx.check(_acx.lib.acx_search_minima_enable(1))
        try:
            out = run_search(kind, presentation, max_nodes_to_explore, cyclical, want_stats)
            n = C.c_int64(0)
            buf = np.zeros(256, np.int32)
            _acx.check(_acx.lib.acx_search_last_minima(_acx.ptr(buf, C.c_int32), len(buf), C.byref(n)))
            for length in buf[: n.value].tolist():
                print(f"New minimal length found: {length}")
            return out
        finally:
            _acx.check(_acx.lib.acx_search_minima_enable(0))
    p = _acx.as_i8_rows(np.array(presentation))
    L = p.size // 2
    cap = 1 << 12
    while True:
        pa = np.empty(cap, np.int32)
        pl = np.empty(cap, np.int32)
        solved, n, st = C.c_int32(), C.c_int64(), _acx.SearchStats()
        rc = _acx.lib.acx_search(kind, _acx.ptr(p, C.c_int8), L, int(max_nodes_to_explore), int(bool(cyclical)), C.byref(solved),
                                 _acx.ptr(pa, C.c_int32), _acx.ptr(pl, C.c_int32), cap, C.byref(n), C.byref(st))
        if rc == _acx.E_CAPACITY and n.value > cap:
            cap = int(n.value)
            continue
        if rc == _acx.E_ROWERR:
            raise AssertionError(_acx.last_error())
        _acx.check(rc, "acx_search")
        break
    path = [(int(a), int(l)) for a, l in zip(pa[: n.value], pl[: n.value])] if n.value else None
    stats = dict(nodes=st.nodes, expanded=st.expanded, children=st.children, levels=st.levels, min_len=st.min_len, seconds=st.seconds)
    return bool(solved.value), path, stats


def run_search_many(kind, presentations, max_nodes_to_explore, cyclical, n_threads=16, path_cap=4096):
    """Independent searches on one GPU, overlapped (acx_search_many).  `presentations` [n, 2L].
    -> list of (solved, path, stats) in input order, each identical to what run_search returns."""
    _acx.require_device()
    rows = _acx.as_i8_rows(np.asarray(presentations))
    n, width = rows.shape
    L = width // 2
    solved = np.zeros(n, np.int32)
    pa = np.empty((n, path_cap), np.int32)
    pl = np.empty((n, path_cap), np.int32)
    pn = np.zeros(n, np.int64)
    rcs = np.zeros(n, np.int32)
    stats = (_acx.SearchStats * n)()
    rc = _acx.lib.acx_search_many(kind, _acx.ptr(rows, C.c_int8), n, L, int(max_nodes_to_explore), int(bool(cyclical)), int(n_threads),
                                  _acx.ptr(solved, C.c_int32), _acx.ptr(pa, C.c_int32), _acx.ptr(pl, C.c_int32), path_cap, _acx.ptr(pn, C.c_int64),
                                  stats, _acx.ptr(rcs, C.c_int32))
    if rc == _acx.E_ROWERR and (rcs == _acx.E_ROWERR).any():
        raise AssertionError(_acx.last_error())
    _acx.check(rc, "acx_search_many")
    out = []
    for k in range(n):
        if rcs[k] == _acx.E_CAPACITY:  # rare: a path longer than path_cap -> redo that search alone
            out.append(run_search(kind, rows[k], max_nodes_to_explore, cyclical))
            continue
        st = stats[k]
        path = [(int(a), int(l)) for a, l in zip(pa[k, : pn[k]], pl[k, : pn[k]])] if pn[k] else None
        out.append((bool(solved[k]), path, dict(nodes=st.nodes, expanded=st.expanded, children=st.children, levels=st.levels, min_len=st.min_len,
                                                seconds=st.seconds)))
    return out


def run_search_groups(kind, groups, max_nodes_to_explore, cyclical, n_threads=16, path_cap=4096):
    """`run_search_many` on several batches of presentations of DIFFERENT widths (the Miller-Schupp presentations of each n have
    their own max_relator_length).  greedy_search: one host thread per batch, so that their launches share the GPU -- a batch of
    170 one-workgroup searches fills 170 of the 256 compute units.  bfs: one batch after the other (see below).
    -> list (per batch) of lists of (solved, path, stats)."""
    from concurrent.futures import ThreadPoolExecutor

    if len(groups) <= 1:
        return [run_search_many(kind, g, max_nodes_to_explore, cyclical, n_threads, path_cap) for g in groups]
    # Longest first: the widest presentations (128-bit keys from max_relator_length 30) are the slowest searches, and a search that
    # starts late runs its tail alone on an otherwise idle GPU.  Their launches go out first (the other batches follow a moment
    # later), so the hardware hands their workgroups compute units before the shorter ones.
    order = sorted(range(len(groups)), key=lambda k: -np.asarray(groups[k]).shape[-1])
    stagger = os.environ.get("ACX_SWEEP_STAGGER_MS", "0")  # (a pause between the wide and the narrow batches: measured no better than none)
    wide = [k for k in order if np.asarray(groups[k]).shape[-1] // 2 > 29]
    # bfs: a batch of searches fills the GPU on its own (acx_bfs_many.h: a tile of every search's frontier per workgroup), and batches in
    # flight together only evict each other's tables from the caches -- measured: the seven widths one after another 144 ms, all at once
    # 190-250 ms.  greedy: one workgroup per search, so the batches must overlap to fill the compute units.
    workers = int(os.environ.get("ACX_SWEEP_WORKERS", "0")) or (1 if kind == _acx.SEARCH_BFS and os.environ.get("ACX_BFS_MANY") != "multi" else len(groups))
    with ThreadPoolExecutor(max_workers=workers) as ex:
        futs = {}
        for pos, k in enumerate(order):
            if wide and pos == len(wide) and float(stagger) > 0:
                time.sleep(float(stagger) * 1e-3)
            futs[k] = ex.submit(run_search_many, kind, groups[k], max_nodes_to_explore, cyclical, n_threads, path_cap)
        return [futs[k].result() for k in range(len(groups))]


def self_check(search_fn, budget=10**6):
    """Solve AK(2) with `search_fn`, then replay the returned path move by move and report whether it ends at a trivial
    presentation (what the reference's search modules do when run as scripts)."""
    from ac_solver.envs.ac_moves import ACMove
    from ac_solver.envs.utils import is_presentation_trivial

    state = np.array([1, 1, -2, -2, -2, 0, 0, 1, 2, 1, -2, -1, -2, 0])  # AK(2) at max_relator_length = 7
    solved, path = search_fn(presentation=state, max_nodes_to_explore=budget)
    print(f"{search_fn.__name__}: solved = {solved}, path of {len(path) if path else 0} entries")
    if solved:
        lengths = [5, 6]
        for action, _ in path[1:]:
            state, lengths = ACMove(move_id=action, presentation=state, max_relator_length=7, lengths=lengths, cyclical=False)
        print(f"replayed path ends at {state.tolist()}; trivial: {is_presentation_trivial(state)}")
        return bool(is_presentation_trivial(state))
    return False
