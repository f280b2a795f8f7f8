"""Shared driver for bfs / greedy_search: hands the presentation to libacx's device frontier."""
import ctypes as C
import gc
import threading

import numpy as np

from ac_solver import _acx


MAX_SEARCH_RELATOR_LENGTH = 64  # one 128-bit key word per relator (csrc/acx_keys.h: 2 bits per letter; up to 61 the length rides in the word,
#                                 62 .. 64 use a key that only FREELY REDUCED words have -- which every state of a search is)


def _check_width(L, presentation=None):
    """The reference takes any max_relator_length; the device frontier names a relator by one 128-bit key: 64 letters at most, and from 62
    letters on the root has to be freely reduced (its own Miller-Schupp presentations are: miller_schupp.py:43 reaches 64 at n = 14)."""
    if L > MAX_SEARCH_RELATOR_LENGTH:
        raise ValueError(f"max_relator_length = {L} is not supported by the device search: at most {MAX_SEARCH_RELATOR_LENGTH} "
                         "(a relator is named by one 128-bit key word: 64 letters of 2 bits)")
    if L > 61 and presentation is not None:
        p = np.asarray(presentation)
        for h in (0, 1):
            w = p[h * L:(h + 1) * L]
            w = w[w != 0]
            if len(w) > 1 and bool((w[:-1] == -w[1:]).any()):
                raise ValueError(f"max_relator_length = {L} (> 61): the presentation must be freely reduced (csrc/acx_keys.h: the 128-bit key of these "
                                 "lengths names reduced words only)")


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
    _check_width(L, p)
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
    path = list(zip(pa[: n.value].tolist(), pl[: n.value].tolist())) if n.value else None  # (tolist: Python ints, as the reference's tuples hold)
    stats = dict(nodes=st.nodes, expanded=st.expanded, children=st.children, levels=st.levels, min_len=st.min_len, seconds=st.seconds)
    return bool(solved.value), path, stats


_path_io = threading.local()


def _path_buffers(n, path_cap):
    """the (n, path_cap) int32 arrays the library writes the paths into, kept per thread between calls of the same shape: a search writes a
    few hundred entries at the start of its 16 KB row, so fresh arrays cost a page fault per row and array (2 400 for a sweep: 3-4 ms)"""
    held = getattr(_path_io, "held", None)
    if held is None or held[0] != (n, path_cap):
        held = _path_io.held = ((n, path_cap), np.empty((n, path_cap), np.int32), np.empty((n, path_cap), np.int32))
    return held[1], held[2]


def run_search_many(kind, presentations, max_nodes_to_explore, cyclical, n_threads=16, path_cap=4096):
    """Independent searches on one GPU, overlapped (acx_search_many).  `presentations` [n, 2L].
    -> list of (solved, path, stats) in input order, each identical to what run_search returns."""
    _acx.require_device()
    rows = _acx.as_i8_rows(np.asarray(presentations))
    n, width = rows.shape
    L = width // 2
    _check_width(L)
    solved = np.zeros(n, np.int32)
    pa, pl = _path_buffers(n, path_cap)
    pn = np.zeros(n, np.int64)
    rcs = np.zeros(n, np.int32)
    stats = (_acx.SearchStats * n)()
    rc = _acx.lib.acx_search_many(kind, _acx.ptr(rows, C.c_int8), n, L, int(max_nodes_to_explore), int(bool(cyclical)), int(n_threads),
                                  _acx.ptr(solved, C.c_int32), _acx.ptr(pa, C.c_int32), _acx.ptr(pl, C.c_int32), path_cap, _acx.ptr(pn, C.c_int64),
                                  stats, _acx.ptr(rcs, C.c_int32))
    if rc == _acx.E_ROWERR and (rcs == _acx.E_ROWERR).any():
        raise AssertionError(_acx.last_error())
    _acx.check(rc, "acx_search_many")
    return _collect(n, solved, pa, pl, pn, rcs, stats, lambda k: run_search(kind, rows[k], max_nodes_to_explore, cyclical))


_STATS_DTYPE = np.dtype([("nodes", np.int64), ("expanded", np.int64), ("children", np.int64), ("levels", np.int64), ("min_len", np.int32), ("seconds", np.float64)],
                        align=True)  # = _acx.SearchStats (include/acx.h: acx_search_stats)


def _collect(n, solved, pa, pl, pn, rcs, stats, redo):
    """the output arrays of acx_search_many / acx_search_groups -> [(solved, path, stats)] (plain Python ints, as the reference's tuples hold); a search
    whose path outgrew the buffer (rare) is redone alone by `redo(k)`.  Whole-array conversions: a sweep is 1190 paths."""
    assert _STATS_DTYPE.itemsize == C.sizeof(_acx.SearchStats)
    st = np.frombuffer(stats, dtype=_STATS_DTYPE, count=n) if n else np.zeros(0, _STATS_DTYPE)
    cols = [st[f].tolist() for f in ("nodes", "expanded", "children", "levels", "min_len", "seconds")]
    ok, cnt, codes = solved.tolist(), pn.tolist(), rcs.tolist()
    out = []
    # (tens of thousands of tuples and a dict per search, none of them part of a cycle: the collector's generation counts would trigger
    # several full passes over the caller's heap in the middle of this loop -- 3 ms of conversion became 8)
    gc_was_on = gc.isenabled()
    gc.disable()
    try:
        _collect_rows(n, ok, cnt, codes, pa, pl, cols, redo, out)
    finally:
        if gc_was_on:
            gc.enable()
    return out


def _collect_rows(n, ok, cnt, codes, pa, pl, cols, redo, out):
    for k in range(n):
        if codes[k] == _acx.E_CAPACITY:
            out.append(redo(k))
            continue
        m = cnt[k]
        path = list(zip(pa[k, :m].tolist(), pl[k, :m].tolist())) if m else None
        out.append((bool(ok[k]), path, dict(nodes=cols[0][k], expanded=cols[1][k], children=cols[2][k], levels=cols[3][k], min_len=cols[4][k], seconds=cols[5][k])))


def run_search_groups(kind, groups, max_nodes_to_explore, cyclical, n_threads=16, path_cap=4096):
    """Searches on several batches of presentations of DIFFERENT widths (the Miller-Schupp presentations of each n have their own
    max_relator_length) in one call of acx_search_groups.  greedy_search: all searches of all batches are jobs of one launch per
    key width, taken by a fixed set of workgroups one after the other.  bfs: one batch after the other -- a batch fills the GPU on
    its own (acx_bfs_many.h: a tile of every search's frontier per workgroup), and batches in flight together only evict each
    other's tables from the caches (measured: the seven widths one after another 144 ms, all at once 190-250 ms).
    -> list (per batch) of lists of (solved, path, stats), each identical to what run_search returns."""
    _acx.require_device()
    rows = [_acx.as_i8_rows(np.asarray(g)) for g in groups]
    ng = len(rows)
    if ng == 0:
        return []
    counts = np.array([r.shape[0] for r in rows], np.int64)
    widths = np.array([r.shape[1] // 2 for r in rows], np.int32)
    _check_width(int(widths.max()))
    n = int(counts.sum())
    ptrs = (C.POINTER(C.c_int8) * ng)(*[_acx.ptr(r, C.c_int8) for r in rows])
    solved = np.zeros(n, np.int32)
    pa, pl = _path_buffers(n, path_cap)
    pn = np.zeros(n, np.int64)
    rcs = np.zeros(n, np.int32)
    stats = (_acx.SearchStats * n)()
    rc = _acx.lib.acx_search_groups(kind, ng, ptrs, _acx.ptr(counts, C.c_int64), _acx.ptr(widths, C.c_int32), int(max_nodes_to_explore), int(bool(cyclical)),
                                    _acx.ptr(solved, C.c_int32), _acx.ptr(pa, C.c_int32), _acx.ptr(pl, C.c_int32), path_cap, _acx.ptr(pn, C.c_int64), stats,
                                    _acx.ptr(rcs, C.c_int32))
    if rc == _acx.E_ROWERR and (rcs == _acx.E_ROWERR).any():
        raise AssertionError(_acx.last_error())
    _acx.check(rc, "acx_search_groups")
    first = np.concatenate([[0], np.cumsum(counts)]).tolist()

    def redo(k):
        g = int(np.searchsorted(first, k, side="right")) - 1
        return run_search(kind, rows[g][k - first[g]], max_nodes_to_explore, cyclical)

    flat = _collect(n, solved, pa, pl, pn, rcs, stats, redo)
    return [flat[first[g]:first[g + 1]] for g in range(ng)]


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
