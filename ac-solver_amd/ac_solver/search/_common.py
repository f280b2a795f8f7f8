"""Shared driver for bfs / greedy_search: hands the presentation to libacx's device frontier."""
import ctypes as C

import numpy as np

from ac_solver import _acx


def run_search(kind, presentation, max_nodes_to_explore, cyclical, want_stats=False):
    _acx.require_device()
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
