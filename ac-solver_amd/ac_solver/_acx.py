"""ctypes binding of libacx.so (include/acx.h) -- the only way the Python surface reaches the
HIP kernels.  There is no CPU fallback: if the library is missing the import fails loudly, and
if no GPU is visible every compute call raises `AcxError`.

Reference-side note: the reference is pure Python, so this module is exactly the
stub a maintainer of shehper/AC-Solver would add to route ac_solver/envs and ac_solver/search
through the accelerator.
"""
import ctypes as C
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ACX_LIB", os.path.join(os.path.dirname(_HERE), "lib", "libacx.so"))

# return codes / flags (mirror include/acx.h)
OK, E_INVAL, E_NODEVICE, E_NOMEM, E_ROWERR, E_CAPACITY = 0, -1, -2, -3, -4, -5
F_CYCLICAL, F_NO_SIMPLIFY, F_NO_MOVE, F_BYTES = 1, 2, 4, 8
U8, I32, I64, I8, F32 = 0, 1, 2, 3, 4
ENV_RECORD_ACTIONS = 1
SEARCH_BFS, SEARCH_GREEDY = 0, 1
ERR_ASSERT, ERR_INDEX, ERR_VALUE, ERR_UNPACKABLE = 1, 2, 3, 250
# acx_set_option: tuning / test knobs of the search entry points (process wide; nothing on a call path reads the environment)
OPT_BFS_NO_RUNAHEAD, OPT_GREEDY_HOST, OPT_GREEDY_HAND_MIN, OPT_MEGA_RANK_MAX, OPT_BFS_MANY_BMAX, OPT_GREEDY_SLOTS, OPT_GENERAL_MOVE, OPT_GREEDY_SCRATCH, OPT_GREEDY_KEEP_ORDER = range(9)


class AcxError(RuntimeError):
    pass


class SearchStats(C.Structure):
    _fields_ = [("nodes", C.c_int64), ("expanded", C.c_int64), ("children", C.c_int64), ("levels", C.c_int64),
                ("min_len", C.c_int32), ("seconds", C.c_double)]


# the communicator of acx_bfs_sharded (include/acx.h: acx_comm): two collectives as plain function pointers
ALL_TO_ALL_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)
ALL_REDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p)
RED_SUM, RED_MAX = 0, 1


class Comm(C.Structure):
    _fields_ = [("rank", C.c_int32), ("world", C.c_int32), ("ctx", C.c_void_p), ("all_to_all", ALL_TO_ALL_FN), ("all_reduce", ALL_REDUCE_FN)]


class ShardOpts(C.Structure):
    _fields_ = [("batch_parents", C.c_int64), ("replicate_below", C.c_int64), ("region_fill", C.c_int32), ("overlap", C.c_int32),
                ("mask_comm", C.POINTER(Comm)), ("fail_at_call", C.c_int32), ("fail_rank", C.c_int32), ("log_fraction_q8", C.c_int32)]


class ShardRunStats(C.Structure):
    _fields_ = [("nodes", C.c_int64), ("expanded", C.c_int64), ("levels", C.c_int64), ("chunks", C.c_int64), ("replicated_levels", C.c_int64),
                ("local_nodes", C.c_int64), ("all_to_all_calls", C.c_int64), ("all_to_all_bytes", C.c_int64), ("all_reduce_calls", C.c_int64),
                ("all_reduce_bytes", C.c_int64), ("min_len", C.c_int32), ("reruns", C.c_int32), ("setup_seconds", C.c_double), ("loop_seconds", C.c_double)]


if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: build the HIP extension first (python __graft_entry__.py, or "
        "make -C ac-solver_amd/csrc).  ac_solver has no CPU fallback.")



def _preload_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm bundles its own libamdhip64.so (SONAME libamdhip64.so.7).
    If libacx.so were loaded first it would pull in the system copy, and a later `import torch` would load
    the bundled one as a SECOND runtime (its NEEDED entry is the bare file name, which does not match the
    loaded SONAME) -- the GPU then disappears for torch.  So when torch is installed but not yet imported,
    load its runtime first; libacx's NEEDED libamdhip64.so.7 then binds to it."""
    import importlib.util
    import sys

    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec and spec.submodule_search_locations:
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(cand):
            try:
                C.CDLL(cand, mode=C.RTLD_GLOBAL)
            except OSError:
                pass


_preload_hip_runtime()
lib = C.CDLL(LIB_PATH)

_i8p, _u8p, _i32p, _i64p, _f32p = (C.POINTER(t) for t in (C.c_int8, C.c_uint8, C.c_int32, C.c_int64, C.c_float))
_vp = C.c_void_p

SIGNATURES = {
    "acx_version": (C.c_int, []),
    "acx_last_error": (C.c_char_p, []),
    "acx_device_count": (C.c_int, []),
    "acx_move_batch_device": (C.c_int, [_vp, _vp, C.c_int, C.c_int64, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp]),
    "acx_move_batch": (C.c_int, [_i8p, _u8p, C.c_int64, C.c_int, C.c_int, _i8p, _i32p, _u8p, _i32p]),
    "acx_simplify_relators": (C.c_int, [_i8p, C.c_int64, C.c_int, C.c_int, _i8p, _i32p, _u8p]),
    "acx_replay_paths": (C.c_int, [_i8p, C.c_int64, C.c_int, C.c_int, _i32p, _i64p, _i32p, _u8p, _i8p]),
    "acx_env_create": (_vp, [C.c_int64, C.c_int, C.c_int64, C.c_int]),
    "acx_env_destroy": (None, [_vp]),
    "acx_env_set_initial": (C.c_int, [_vp, _i8p, _i64p, C.c_int64, _vp]),
    "acx_env_reset": (C.c_int, [_vp, _i8p, _i64p, C.c_int64, _vp]),
    "acx_env_reset_device": (C.c_int, [_vp, _vp, _vp, C.c_int64, _vp, _vp]),
    "acx_env_set_supermoves": (C.c_int, [_vp, _u8p, _i32p, C.c_int, _vp]),
    "acx_env_step": (C.c_int, [_vp, _vp, C.c_int, _vp, C.c_int, _vp, C.c_float, C.c_float, _vp, _vp, _vp, C.c_int, _vp]),
    "acx_env_step_host": (C.c_int, [_vp, _i64p, _i8p, _f32p, _u8p, _u8p, _i8p, C.c_int, _u8p, _vp]),
    "acx_env_rollout": (C.c_int, [_vp, _vp, C.c_int64, _vp, C.c_float, C.c_float, _vp, _vp, C.c_int, _vp]),
    "acx_env_observe": (C.c_int, [_vp, _vp, C.c_int, _vp]),
    "acx_env_get": (C.c_int, [_vp, _i64p, C.c_int64, _i8p, _i32p, _i32p, _vp]),
    "acx_env_get_actions": (C.c_int, [_vp, C.c_int64, C.c_int, _i32p, C.c_int64, _i64p, _vp]),
    "acx_env_get_errors": (C.c_int, [_vp, _u8p, C.c_int, _vp]),
    "acx_env_max_reward": (C.c_int64, [_vp]),
    "acx_search_many": (C.c_int, [C.c_int, _i8p, C.c_int64, C.c_int, C.c_int64, C.c_int, C.c_int, _i32p, _i32p, _i32p, C.c_int64, _i64p,
                               C.POINTER(SearchStats), _i32p]),
    "acx_search_groups": (C.c_int, [C.c_int, C.c_int, C.POINTER(_i8p), _i64p, _i32p, C.c_int64, C.c_int, _i32p, _i32p, _i32p, C.c_int64, _i64p,
                                 C.POINTER(SearchStats), _i32p]),
    "acx_shard_key_words": (C.c_int, [C.c_int]),
    "acx_shard_layout": (C.c_int, [C.c_int64, C.c_int, C.c_int, C.c_int, _i64p, _i64p, _i64p]),
    "acx_shard_create": (_vp, [C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int, C.c_int]),
    "acx_shard_destroy": (None, [_vp]),
    "acx_shard_attach": (C.c_int, [_vp, _vp, C.c_int64, _vp, C.c_int64, _vp]),
    "acx_shard_root_record": (C.c_int, [_vp, _i8p, _i64p]),
    "acx_shard_seed": (C.c_int, [_vp, _i64p, _vp]),
    "acx_shard_owner": (C.c_int, [C.c_int, _i64p, C.c_int]),
    "acx_shard_check_owners": (C.c_int, [_vp, _i64p, _vp]),
    "acx_shard_chunk_expand": (C.c_int, [_vp, C.c_int64, C.c_int64, C.c_int, C.c_int, _i64p, _i64p, _vp]),
    "acx_shard_chunk_insert": (C.c_int, [_vp, _vp]),
    "acx_shard_chunk_insert_dead": (C.c_int, [_vp, _vp]),
    "acx_shard_chunk_commit": (C.c_int, [_vp, C.c_int64, _vp]),
    "acx_shard_ctl_snapshot": (C.c_int, [_vp, C.c_int, _vp]),
    "acx_shard_ctl_wait": (C.c_int, [_vp, C.c_int, _i64p]),
    "acx_shard_fail": (C.c_int, [_vp, _vp]),
    "acx_shard_find": (C.c_int, [_vp, C.c_int64, _i64p, _vp]),
    "acx_shard_node_info": (C.c_int, [_vp, C.c_int64, _i64p]),
    "acx_shard_set_replicated": (C.c_int, [_vp, C.c_int]),
    "acx_shard_partition": (C.c_int, [_vp, _vp]),
    "acx_shard_walk": (C.c_int, [_vp, C.c_int64, C.c_int64, _i64p, _vp]),
    "acx_rccl_available": (C.c_int, []),
    "acx_comm_rccl": (C.c_int, [_vp, C.POINTER(Comm)]),
    "acx_rccl_unique_id": (C.c_int, [_vp]),
    "acx_rccl_comm_create": (C.c_int, [_vp, C.c_int, C.c_int, C.POINTER(_vp)]),
    "acx_rccl_comm_destroy": (C.c_int, [_vp]),
    "acx_bfs_sharded": (C.c_int, [_i8p, C.c_int, C.c_int64, C.c_int, C.POINTER(Comm), C.POINTER(ShardOpts), _i32p, _i32p, _i32p, C.c_int64, _i64p,
                               C.POINTER(ShardRunStats), _vp]),
    "acx_release_cached_memory": (C.c_int, []),
    "acx_set_option": (C.c_int, [C.c_int, C.c_int64]),
    "acx_get_option": (C.c_int64, [C.c_int]),
    "acx_policy_sample": (C.c_int, [_vp, C.c_int, C.c_int64, C.c_int, _vp, _vp, C.c_int, C.c_uint64, _vp, _vp, _vp, _vp]),
    "acx_policy_packed_bytes": (C.c_int64, [C.c_int]),
    "acx_search_minima_enable": (C.c_int, [C.c_int]),
    "acx_search_last_minima": (C.c_int, [_i32p, C.c_int64, _i64p]),
    "acx_search_digest_enable": (C.c_int, [C.c_int]),
    "acx_search_last_digest": (C.c_int, [C.POINTER(C.c_uint64)]),
    "acx_simplex_graph": (C.c_int, [C.c_int, C.c_int, C.c_int64, C.c_int64, _i64p, _u8p, _i64p, C.POINTER(C.c_uint32), _u8p]),
    "acx_ball_sizes": (C.c_int, [_i8p, C.c_int64, C.c_int, C.c_int, C.c_int, _i64p, C.POINTER(C.c_int32)]),
    "acx_search": (C.c_int, [C.c_int, _i8p, C.c_int, C.c_int64, C.c_int, _i32p, _i32p, _i32p, C.c_int64, _i64p, C.POINTER(SearchStats)]),
}
for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)  # AttributeError here means the library is stale: rebuild it
    _fn.restype, _fn.argtypes = _res, _args


def last_error():
    return (lib.acx_last_error() or b"").decode()


def check(rc, what="acx"):
    if rc != OK:
        raise AcxError(f"{what} failed (code {rc}): {last_error()}")


def device_count():
    return lib.acx_device_count()


def require_device():
    if device_count() < 1:
        raise AcxError("no MI355X / HIP device visible: ac_solver runs its moves, environment and searches "
                       "in HIP kernels and has no CPU fallback")


def ptr(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def as_i8_rows(a, what="presentation"):
    """-> C-contiguous int8 copy; values must be representable (letters are small integers)"""
    arr = np.asarray(a)
    if arr.dtype != np.int8:
        if arr.size and (arr.min() < -127 or arr.max() > 127):
            raise ValueError(f"{what}: letters must fit in int8")
        arr = arr.astype(np.int8)
    return np.ascontiguousarray(arr)


def move_rows(rows, actions, L, flags):
    """Byte-exact / packed batched ACMove on host arrays. -> (out, lens, err, fit)"""
    require_device()
    rows = as_i8_rows(rows)
    n = rows.shape[0]
    out = np.empty_like(rows)
    lens = np.empty((n, 2), np.int32)
    err = np.empty(n, np.uint8)
    fit = np.empty(n, np.int32)
    act = None if actions is None else np.ascontiguousarray(actions, np.uint8)
    check(lib.acx_move_batch(ptr(rows, C.c_int8), None if act is None else ptr(act, C.c_uint8), n, L, flags, ptr(out, C.c_int8),
                             ptr(lens, C.c_int32), ptr(err, C.c_uint8), ptr(fit, C.c_int32) if flags & F_BYTES else None), "acx_move_batch")
    return out, lens, err, fit


_one_row_io = threading.local()


def move_one_row(arr, move_id, L, flags):
    """ACMove on ONE presentation through the packed kernel: the host arrays and their ctypes pointers are made once per thread and
    width (`ndarray.ctypes.data_as` costs more than the device call's own host work).  -> (out row, len0, len1, err)"""
    io = getattr(_one_row_io, "by_width", None)
    if io is None:
        require_device()
        io = _one_row_io.by_width = {}
    slot = io.get(L)
    if slot is None:
        rows, act, out = np.zeros((1, 2 * L), np.int8), np.zeros(1, np.uint8), np.zeros((1, 2 * L), np.int8)
        lens, err = np.zeros((1, 2), np.int32), np.zeros(1, np.uint8)
        slot = io[L] = (rows, act, out, lens, err, ptr(rows, C.c_int8), ptr(act, C.c_uint8), ptr(out, C.c_int8), ptr(lens, C.c_int32),
                        ptr(err, C.c_uint8))
    rows, act, out, lens, err, p_rows, p_act, p_out, p_lens, p_err = slot
    rows[0] = arr if arr.dtype == np.int8 else as_i8_rows(arr)
    act[0] = move_id
    check(lib.acx_move_batch(p_rows, p_act, 1, L, flags, p_out, p_lens, p_err, None), "acx_move_batch")
    return out[0], int(lens[0, 0]), int(lens[0, 1]), int(err[0])


def simplify_rows(rows, cyclical):
    require_device()
    rows = as_i8_rows(rows, "relator")
    n, width = rows.shape
    out = np.empty_like(rows)
    lens = np.empty((n, 2), np.int32)
    err = np.empty(n, np.uint8)
    check(lib.acx_simplify_relators(ptr(rows, C.c_int8), n, width, int(bool(cyclical)), ptr(out, C.c_int8), ptr(lens, C.c_int32),
                                    ptr(err, C.c_uint8)), "acx_simplify_relators")
    return out, lens, err


class options:
    """`with _acx.options(OPT_GREEDY_HAND_MIN=16, ...):` -- library options (include/acx.h: ACX_OPT_*) for the duration of a block, the
    previous values restored on the way out.  Process wide: meant for tests and tuning runs, not for concurrent callers."""

    def __init__(self, **values):
        self.values = {globals()[k]: int(v) for k, v in values.items()}

    def __enter__(self):
        self.saved = {k: lib.acx_get_option(k) for k in self.values}
        for k, v in self.values.items():
            check(lib.acx_set_option(k, v), "acx_set_option")
        return self

    def __exit__(self, *exc):
        for k, v in self.saved.items():
            lib.acx_set_option(k, v)
        return False
