"""ctypes binding of oracle/libac_oracle.so (the CPU restatement of the reference).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package never imports this module.

The Python-level helpers return what the reference returns (arrays, lists of tuples) and
raise the exception class the reference raises, so parity tests can compare directly.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libac_oracle.so")

_EXC = {1: AssertionError, 2: IndexError, 3: ValueError, 4: MemoryError}


class Stats(C.Structure):
    _fields_ = [("nodes", C.c_int64), ("expanded", C.c_int64), ("moves", C.c_int64), ("min_len", C.c_int32)]


def build():
    """Compile the oracle with gcc (seconds)."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "libac_oracle.so", "libac_ball_oracle.so"])


def _load():
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "ac_oracle.c")):
        build()
    lib = C.CDLL(_SO)
    i8p, u8p, i32p, i64p = (C.POINTER(t) for t in (C.c_int8, C.c_uint8, C.c_int32, C.c_int64))
    lib.ac_is_valid_presentation.argtypes = [i8p, C.c_int]
    lib.ac_is_trivial.argtypes = [i8p, C.c_int]
    lib.ac_simplify_relator.argtypes = [i8p, C.c_int, C.c_int, C.c_int, C.c_int, i8p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.ac_simplify_presentation.argtypes = [i8p, C.c_int, C.c_int, C.POINTER(C.c_int)]
    lib.ac_concatenate_relators.argtypes = [i8p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]
    lib.ac_conjugate.argtypes = [i8p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]
    lib.ac_move.argtypes = [C.c_int, i8p, C.c_int, C.c_int, i8p, C.POINTER(C.c_int)]
    lib.ac_move_batch.argtypes = [i8p, u8p, C.c_int64, C.c_int, C.c_int, i8p, i32p, u8p]
    lib.ac_env_rollout.argtypes = [i8p, i32p, C.c_int64, C.c_int, C.c_int64, u8p, C.c_int64, i32p, u8p, u8p, u8p]
    for f in (lib.ac_bfs, lib.ac_greedy):
        f.argtypes = [i8p, C.c_int, C.c_int64, C.c_int, i32p, i32p, i32p, C.c_int64, i64p, C.POINTER(Stats)]
    return lib


_lib = _load()


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _i8(a):
    return np.ascontiguousarray(np.asarray(a), dtype=np.int8)


def _check(rc):
    if rc != 0:
        raise _EXC.get(-rc, RuntimeError)(f"oracle error code {rc}")


def is_array_valid_presentation(array):
    a = _i8(array)
    return bool(_lib.ac_is_valid_presentation(_p(a, C.c_int8), a.size))


def is_presentation_trivial(array):
    a = _i8(array)
    return bool(_lib.ac_is_trivial(_p(a, C.c_int8), a.size))


def simplify_relator(relator, max_relator_length, cyclical=False, padded=True):
    a = _i8(relator)
    out = np.zeros(max(a.size, max_relator_length, 1), dtype=np.int8)
    n, ln = C.c_int(), C.c_int()
    _check(_lib.ac_simplify_relator(_p(a, C.c_int8), a.size, max_relator_length, int(cyclical), int(padded),
                                    _p(out, C.c_int8), C.byref(n), C.byref(ln)))
    return out[: n.value].copy(), ln.value


def simplify_presentation(presentation, max_relator_length, lengths_of_words=None, cyclical=True):
    p = _i8(presentation).copy()
    lens = (C.c_int * 2)()
    _check(_lib.ac_simplify_presentation(_p(p, C.c_int8), max_relator_length, int(cyclical), lens))
    return p, [lens[0], lens[1]]


def concatenate_relators(presentation, max_relator_length, i, j, sign, lengths):
    p = _i8(presentation).copy()
    lens = (C.c_int * 2)(*lengths)
    _check(_lib.ac_concatenate_relators(_p(p, C.c_int8), max_relator_length, i, j, sign, lens))
    return p, [lens[0], lens[1]]


def conjugate(presentation, max_relator_length, i, j, sign, lengths):
    p = _i8(presentation).copy()
    lens = (C.c_int * 2)(*lengths)
    _check(_lib.ac_conjugate(_p(p, C.c_int8), max_relator_length, i, j, sign, lens))
    return p, [lens[0], lens[1]]


def ACMove(move_id, presentation, max_relator_length, lengths=None, cyclical=True):
    p = _i8(presentation)
    out = np.empty_like(p)
    lens = (C.c_int * 2)()
    _check(_lib.ac_move(int(move_id), _p(p, C.c_int8), max_relator_length, int(cyclical), _p(out, C.c_int8), lens))
    return out, [lens[0], lens[1]]


def move_batch(states, actions, max_relator_length, cyclical=True):
    """states [n, 2L] int8, actions [n] -> (out [n, 2L], lens [n, 2] int32, err [n] uint8)."""
    s = _i8(states)
    n = s.shape[0]
    a = np.ascontiguousarray(actions, dtype=np.uint8)
    out = np.empty_like(s)
    lens = np.empty((n, 2), dtype=np.int32)
    err = np.empty(n, dtype=np.uint8)
    _lib.ac_move_batch(_p(s, C.c_int8), _p(a, C.c_uint8), n, max_relator_length, int(cyclical), _p(out, C.c_int8),
                       _p(lens, C.c_int32), _p(err, C.c_uint8))
    return out, lens, err


def env_rollout(states, counts, horizon, tape, want_outputs=True, out=None):
    """ACEnv.step over an action tape [T, n]; `states` [n, 2L] and `counts` [n] are updated in place.

    Returns (reward [T, n] int32, done [T, n] u8, truncated [T, n] u8, err [n] u8) or None.  `out` = (reward, done,
    truncated) arrays to fill instead of fresh ones (timing loops)."""
    n, W = states.shape
    T = tape.shape[0]
    assert states.dtype == np.int8 and counts.dtype == np.int32 and tape.dtype == np.uint8
    assert states.flags.c_contiguous and tape.flags.c_contiguous
    err = np.zeros(n, dtype=np.uint8)
    if want_outputs:
        if out is not None:
            rew, done, trunc = out
            assert rew.shape == (T, n) and rew.dtype == np.int32 and done.dtype == np.uint8 and trunc.dtype == np.uint8
        else:
            rew = np.empty((T, n), dtype=np.int32)
            done = np.empty((T, n), dtype=np.uint8)
            trunc = np.empty((T, n), dtype=np.uint8)
        args = (_p(rew, C.c_int32), _p(done, C.c_uint8), _p(trunc, C.c_uint8))
    else:
        rew = done = trunc = None
        args = (None, None, None)
    _lib.ac_env_rollout(_p(states, C.c_int8), _p(counts, C.c_int32), n, W // 2, horizon, _p(tape, C.c_uint8), T,
                        *args, _p(err, C.c_uint8))
    return rew, done, trunc, err


def _search(fn, presentation, max_nodes_to_explore, cyclical, want_stats):
    p = _i8(presentation)
    L = p.size // 2
    cap = 1 << 16
    while True:
        pa = np.empty(cap, dtype=np.int32)
        pl = np.empty(cap, dtype=np.int32)
        solved, n, st = C.c_int32(), C.c_int64(), Stats()
        _check(fn(_p(p, C.c_int8), L, int(max_nodes_to_explore), int(cyclical), C.byref(solved), _p(pa, C.c_int32),
                  _p(pl, C.c_int32), cap, C.byref(n), C.byref(st)))
        if n.value <= cap:
            break
        cap = n.value
    path = [(int(a), int(l)) for a, l in zip(pa[: n.value], pl[: n.value])] if n.value else None
    res = (bool(solved.value), path)
    if want_stats:
        return res + ({"nodes": st.nodes, "expanded": st.expanded, "moves": st.moves, "min_len": st.min_len},)
    return res


def bfs(presentation, max_nodes_to_explore=10000, verbose=False, cyclically_reduce_after_moves=False, stats=False):
    return _search(_lib.ac_bfs, presentation, max_nodes_to_explore, cyclically_reduce_after_moves, stats)


def greedy_search(presentation, max_nodes_to_explore=10000, verbose=False, cyclically_reduce_after_moves=False,
                  stats=False):
    return _search(_lib.ac_greedy, presentation, max_nodes_to_explore, cyclically_reduce_after_moves, stats)


# ---- neighbourhood sizes (ac_ball_oracle.c: restatement of barcode_analysis/5_steps_neibourhoods) --------------
_ball = None


def ball_size(presentation, radius=5, classic=False, return_max_length=False):
    """Size of the radius-`radius` ball around `presentation` (zero-padded row [r1 | r2]) under the reference's prime /
    classic moves on sorted pairs of freely reduced relators (neibourhoods.cpp:18-54)."""
    global _ball
    if _ball is None:
        path = os.path.join(_HERE, "libac_ball_oracle.so")
        if not os.path.exists(path):
            build()
        _ball = C.CDLL(path)
        _ball.ac_ball_size.restype = C.c_longlong
        _ball.ac_ball_size.argtypes = [C.POINTER(C.c_int8), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]
    row = np.ascontiguousarray(presentation, dtype=np.int8)
    ml = C.c_int(0)
    size = _ball.ac_ball_size(row.ctypes.data_as(C.POINTER(C.c_int8)), len(row) // 2, int(radius), int(bool(classic)), C.byref(ml))
    if size < 0:
        raise OverflowError("a relator outgrew the oracle's buffer")
    return (int(size), ml.value) if return_max_length else int(size)


def simplex_graph(n, classic=False, cap_nodes=4_000_000, cap_edges=12_000_000):
    """(node_size, edges [E, 2], edge_filtration) of the graph of presentations of total length <= n around <a, b>, in the
    order the reference's ac_bfs.cpp writes them (ac_ball_oracle.c: ac_simplex_graph)."""
    ball_size([1, 2], 0)  # loads the library
    f = _ball.ac_simplex_graph
    f.restype = C.c_int
    f.argtypes = [C.c_int, C.c_int, C.c_longlong, C.c_longlong, C.POINTER(C.c_longlong), C.POINTER(C.c_uint8), C.POINTER(C.c_longlong),
                  C.POINTER(C.c_uint32), C.POINTER(C.c_uint8)]
    ns, ed, ef = np.zeros(cap_nodes, np.uint8), np.zeros(2 * cap_edges, np.uint32), np.zeros(cap_edges, np.uint8)
    nn, ne = C.c_longlong(), C.c_longlong()
    rc = f(int(n), int(bool(classic)), cap_nodes, cap_edges, C.byref(nn), ns.ctypes.data_as(C.POINTER(C.c_uint8)), C.byref(ne),
           ed.ctypes.data_as(C.POINTER(C.c_uint32)), ef.ctypes.data_as(C.POINTER(C.c_uint8)))
    if rc:
        raise OverflowError(f"ac_simplex_graph returned {rc}")
    return ns[:nn.value].copy(), ed[:2 * ne.value].reshape(-1, 2).copy(), ef[:ne.value].copy()
