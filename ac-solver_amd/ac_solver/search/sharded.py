"""Breadth-first search with the frontier sharded over the GPUs of one node -- the multi-GPU form of
`bfs` (reference: ac_solver/search/breadth_first.py:15-97), bit-identical to it for every world size.

States are partitioned by `hash(packed key) mod world`.  A level is processed in chunks of consecutive
global frontier positions; for each chunk every rank

  1. expands the frontier nodes it owns (HIP kernel, 12 children per node; a child's tag
     `12 * global_parent_position + action` is the order in which the reference generates it),
  2. routes each child record to the owner of the child's key -- ONE all-to-all (RCCL over xGMI),
  3. deduplicates what it received against its slice of the visited set, minimum tag wins (HIP),
  4. all-reduces (sum) one 12-bit mask per parent of the chunk -- bit a set by the owner of child (parent, a)
     when that child is a new state -- so that every rank derives the same global FIFO numbering (position
     of a new state = new states of earlier parents + earlier set bits of its own parent), the same budget
     decision ("first parent after which len(tree_nodes) >= max_nodes", breadth_first.py:91-95) and the same
     success decision (smallest tag of a length-2 child, :84-85).  The exchange is 4 bytes per PARENT; no
     rank ever holds the tags of the other ranks' new states.

The per-rank work goes through an *engine* (the C ABI `acx_shard_*` of libacx in production; the CPU
tests plug in a NumPy engine built on the oracle) and the exchange through a *comm* (torch.distributed:
backend "nccl" is RCCL on ROCm; "gloo" in the CPU tests; an in-process thread communicator lets one GPU
play several ranks in the GPU tests).
"""
import ctypes as C
import sys

import numpy as np

INF = 1 << 62
_FORCE_EXCHANGE = False  # tests: route through the communicator even when world == 1
_ID_MASK = (1 << 40) - 1


def _torch():
    import torch

    return torch


# ------------------------------------------------------------------------------------------ comms ---
class TorchDistComm:
    """torch.distributed communicator (nccl == RCCL on ROCm, gloo on CPU)."""

    def __init__(self, device, group=None):
        import torch.distributed as dist

        self.dist, self.group, self.device = dist, group, device
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)

    def all_to_all_rows(self, send, counts):
        """send [m, C] int64 grouped by destination, counts[d] rows for rank d -> rows received, grouped by source"""
        torch = _torch()
        c_send = torch.tensor(counts, dtype=torch.int64, device=self.device)
        c_recv = torch.empty_like(c_send)
        self.dist.all_to_all_single(c_recv, c_send, group=self.group)
        recv_counts = c_recv.tolist()
        cols = send.shape[1]
        recv = torch.empty((sum(recv_counts), cols), dtype=send.dtype, device=self.device)
        self.dist.all_to_all_single(recv, send.contiguous(), output_split_sizes=recv_counts, input_split_sizes=list(counts), group=self.group)
        return recv

    def all_gather_var(self, t):
        """1-D int64 tensors of different lengths -> list of per-rank tensors"""
        torch = _torch()
        n = torch.tensor([t.numel()], dtype=torch.int64, device=self.device)
        sizes = [torch.empty_like(n) for _ in range(self.world)]
        self.dist.all_gather(sizes, n, group=self.group)
        sizes = [int(s.item()) for s in sizes]
        cap = max(max(sizes), 1)
        pad = torch.zeros(cap, dtype=t.dtype, device=self.device)
        pad[: t.numel()] = t
        out = [torch.empty_like(pad) for _ in range(self.world)]
        self.dist.all_gather(out, pad, group=self.group)
        return [o[:s] for o, s in zip(out, sizes)]

    def all_reduce(self, t, op):
        ops = {"min": self.dist.ReduceOp.MIN, "max": self.dist.ReduceOp.MAX, "sum": self.dist.ReduceOp.SUM}
        self.dist.all_reduce(t, op=ops[op], group=self.group)
        return t


class SingleComm:
    """world == 1: no exchange at all."""

    rank, world = 0, 1

    def all_to_all_rows(self, send, counts):
        return send

    def all_gather_var(self, t):
        return [t]

    def all_reduce(self, t, op):
        return t


# ---------------------------------------------------------------------------------------- engines ---
class HipShardEngine:
    """Per-rank frontier slice on one GPU: thin wrapper over the acx_shard_* C ABI (include/acx.h)."""

    def __init__(self, L, cyclical, node_cap, batch_cap, rank, world, device=None):
        from ac_solver import _acx

        torch = _torch()
        self._acx = _acx
        _acx.require_device()
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.KW = _acx.lib.acx_shard_key_words(L)
        self.rank = rank
        with torch.cuda.device(self.device):
            self.h = _acx.lib.acx_shard_create(L, int(bool(cyclical)), int(node_cap), int(batch_cap), rank, world)
        if not self.h:
            raise _acx.AcxError(f"acx_shard_create failed: {_acx.last_error()}")
        self.batch_cap = int(batch_cap)

    def __del__(self):
        # not during interpreter shutdown: the HIP runtime may already be tearing down (a hipFree then can block forever)
        if getattr(self, "h", None) and not sys.is_finalizing():
            self._acx.lib.acx_shard_destroy(self.h)
            self.h = None

    def _stream(self):
        return _torch().cuda.current_stream(self.device).cuda_stream

    def root_record(self, presentation):
        rec = np.zeros(self.KW + 2, np.int64)
        p = self._acx.as_i8_rows(np.asarray(presentation))
        rc = self._acx.lib.acx_shard_root_record(self.h, self._acx.ptr(p, C.c_int8), self._acx.ptr(rec, C.c_int64))
        if rc == self._acx.E_ROWERR:
            raise AssertionError(self._acx.last_error())
        self._acx.check(rc, "acx_shard_root_record")
        return rec

    def expand(self, ids, gpos, solved):
        torch = _torch()
        np_ = ids.numel()
        rec = torch.empty((12 * np_, self.KW + 2), dtype=torch.int64, device=self.device)
        if np_:
            self._acx.check(self._acx.lib.acx_shard_expand(self.h, ids.data_ptr(), gpos.data_ptr(), np_, rec.data_ptr(), solved.data_ptr(), self._stream()),
                            "acx_shard_expand")
        return rec

    def expand_routed(self, ids, gpos, solved, world):
        """expand + group by owner on the device: (records grouped by destination rank, counts per rank)"""
        torch = _torch()
        np_ = ids.numel()
        cap = 12 * np_  # a region can take every child: no-op moves keep a child on its parent's rank, so the split is far from uniform
        rec = torch.empty((world * cap, self.KW + 2), dtype=torch.int64, device=self.device)
        cnt = torch.empty(world, dtype=torch.int64, device=self.device)
        self._acx.check(self._acx.lib.acx_shard_expand_routed(self.h, ids.data_ptr() if np_ else None, gpos.data_ptr() if np_ else None, np_,
                                                              rec.data_ptr() if np_ else None, cap, cnt.data_ptr(), solved.data_ptr(), self._stream()),
                        "acx_shard_expand_routed")
        counts = cnt.tolist()
        if max(counts) > cap:
            raise RuntimeError(f"rank {self.rank}: a send region overflowed ({max(counts)} > {cap} records)")
        if world == 1:
            return rec[:counts[0]], counts  # one region: already contiguous
        return torch.cat([rec[o * cap:o * cap + c] for o, c in enumerate(counts)]), counts

    def insert(self, recv, max_tag=None):
        torch = _torch()
        n = recv.shape[0]
        win = torch.empty(max(n, 1), dtype=torch.int64, device=self.device)
        n_win = C.c_int64(0)
        bits = 0 if max_tag is None else max(int(max_tag).bit_length(), 1)
        self._acx.check(self._acx.lib.acx_shard_insert(self.h, recv.data_ptr() if n else None, n, bits, win.data_ptr(), C.byref(n_win), self._stream()),
                        "acx_shard_insert")
        return win[: n_win.value]

    def commit(self, cutoff):
        first, cnt = C.c_int64(0), C.c_int64(0)
        self._acx.check(self._acx.lib.acx_shard_commit(self.h, int(cutoff), C.byref(first), C.byref(cnt), self._stream()), "acx_shard_commit")
        return first.value, cnt.value

    def node_info(self, node_id):
        info = np.zeros(3, np.int64)
        self._acx.check(self._acx.lib.acx_shard_node_info(self.h, int(node_id), self._acx.ptr(info, C.c_int64)), "acx_shard_node_info")
        return int(info[0]), int(info[1]), int(info[2])

    def status(self):
        err, ml = C.c_int32(0), C.c_int32(0)
        self._acx.check(self._acx.lib.acx_shard_status(self.h, C.byref(err), C.byref(ml)), "acx_shard_status")
        return err.value, ml.value


def _default_engine(L, cyclical, node_cap, batch_cap, rank, world):
    return HipShardEngine(L, cyclical, node_cap, batch_cap, rank, world)


# ------------------------------------------------------------------------------------- orchestrator ---
def owner_of(keys, world):
    """Deterministic owner rank of each packed key [m, KW] int64 (same arithmetic on CPU and GPU tensors)."""
    torch = _torch()
    h = torch.zeros(keys.shape[0], dtype=torch.int64, device=keys.device)
    for j in range(keys.shape[1]):
        h = (h ^ keys[:, j]) * -7046029254386353131  # 0x9E3779B97F4A7C15 as int64, wraps
        h = h ^ ((h >> 29) & 0x7FFFFFFFF)
    return (h & 0x7FFFFFFFFFFFFFFF) % world


def bfs_sharded(presentation, max_nodes_to_explore=10000, verbose=False, cyclically_reduce_after_moves=False, comm=None,
                engine_factory=None, batch_parents=1 << 18, want_stats=False):
    """Same contract as `bfs`: returns (is_search_successful, path or None) [+ stats dict], identical on every rank."""
    from ac_solver.envs.utils import is_array_valid_presentation

    torch = _torch()
    assert is_array_valid_presentation(presentation), f"{presentation} is not a valid presentation"
    p = np.array(presentation, dtype=np.int8)
    L = len(p) // 2
    max_nodes = int(max_nodes_to_explore)
    comm = SingleComm() if comm is None else comm
    world, rank = comm.world, comm.rank
    B = int(max(1, min(batch_parents, max(max_nodes, 64))))          # global parents per chunk
    batch_cap = int(12 * B * (2.0 / world if world > 1 else 1.0)) + 4096  # records one rank may receive per chunk
    node_cap = (max_nodes + 12 if world == 1 else int(2.0 * max_nodes / world)) + 4096
    engine = (engine_factory or _default_engine)(L, cyclically_reduce_after_moves, node_cap, batch_cap, rank, world)
    dev = getattr(engine, "device", torch.device("cpu"))
    KW = engine.KW

    def i64(values):
        return torch.tensor(values, dtype=torch.int64, device=dev)

    pop12 = i64([bin(v).count("1") for v in range(4096)])  # popcount of a 12-bit child mask

    # root: inserted by its owner with tag 0, becomes global frontier position 0
    root = engine.root_record(p)
    root_t = i64(root[None, :])
    owner_root = int(owner_of(root_t[:, :KW], world)[0])
    win = engine.insert(root_t if rank == owner_root else root_t[:0])
    first, cnt = engine.commit(INF)
    f_ids = i64(list(range(first, first + cnt)))
    f_gpos = i64([0] * cnt)
    F = 1
    nodes_global = 1
    expanded = levels = 0

    def walk(pref, tail):
        """path of the node `pref` (rank << 40 | id) from the root + tail"""
        rev = []
        while pref >= 0:
            r, nid = pref >> 40, pref & _ID_MASK
            info = i64(list(engine.node_info(nid)) if rank == r else [0, 0, 0])
            if rank != r:
                info[2] = 0
            comm.all_reduce(info, "sum")
            a, tl, pref = (int(v) for v in info.tolist())
            rev.append((a, tl))
        return rev[::-1] + tail

    def finish(ok, path):
        _, min_len = engine.status()
        e = i64([0, -min_len])
        comm.all_reduce(e, "max")
        if want_stats:
            return ok, path, dict(nodes=nodes_global, expanded=expanded, levels=levels, min_len=2 if ok else -int(e[1]), world=world)
        return ok, path

    while F > 0:
        levels += 1
        next_ids, next_gpos, next_count = [], [], 0
        c0 = 0
        while c0 < F:
            c1 = min(F, c0 + B)
            lo = int(torch.searchsorted(f_gpos, i64([c0]))[0]) if f_gpos.numel() else 0
            hi = int(torch.searchsorted(f_gpos, i64([c1]))[0]) if f_gpos.numel() else 0
            solved = i64([INF, INF])  # [0] smallest tag of a length-2 child, [1] smallest (tag << 8 | code) of a move the reference raises on
            # Local failures (a capacity of this rank's engine, a HIP error) must not leave the other ranks waiting in a
            # collective: the rank keeps taking part with empty contributions and reports through the `solved` all-reduce
            # (-1 beats every tag), so that all ranks raise together.
            failure = None
            routed = hasattr(engine, "expand_routed")  # also for one rank: the device-side routing drops unchanged children
            send, counts = torch.empty((0, KW + 2), dtype=torch.int64, device=dev), [0] * world
            try:
                if routed:
                    send, counts = engine.expand_routed(f_ids[lo:hi].contiguous(), f_gpos[lo:hi].contiguous(), solved, world)
                else:
                    recs = engine.expand(f_ids[lo:hi].contiguous(), f_gpos[lo:hi].contiguous(), solved)
                    if world > 1 or _FORCE_EXCHANGE:
                        owners = owner_of(recs[:, :KW], world)
                        order = torch.argsort(owners, stable=True)
                        counts = torch.bincount(owners, minlength=world).tolist()
                        send = recs[order].contiguous()
            except Exception as e:  # noqa: BLE001
                failure = e
                send, counts = torch.empty((0, KW + 2), dtype=torch.int64, device=dev), [0] * world
            if world > 1 or _FORCE_EXCHANGE:
                recv = comm.all_to_all_rows(send, counts)
            else:
                recv = send if (routed or failure is not None) else recs
            win = i64([])
            try:
                if failure is None:
                    if recv.shape[0] > engine.batch_cap:
                        raise RuntimeError(f"rank {rank}: {recv.shape[0]} records exceed the per-chunk capacity {engine.batch_cap}")
                    win = engine.insert(recv, 12 * c1)       # tags of MY new states, ascending
            except Exception as e:  # noqa: BLE001
                failure, win = e, i64([])
            if failure is not None:
                solved = i64([-1, INF])
            # one 12-bit mask per parent of the chunk; every (parent, action) child has exactly one owner, so SUM == OR
            rel = win - 12 * c0
            par, bit = rel // 12, rel % 12
            mask = torch.zeros(c1 - c0, dtype=torch.int32, device=dev)
            if rel.numel():
                mask.index_add_(0, par, torch.bitwise_left_shift(torch.ones_like(bit), bit).to(torch.int32))
            comm.all_reduce(mask, "sum")
            mask = mask.to(torch.int64)
            incl = torch.cumsum(pop12[mask], 0)      # new states up to and including each parent (global)
            comm.all_reduce(solved, "min")
            solved_tag, err_word = (int(v) for v in solved.tolist())
            if solved_tag < 0:
                raise RuntimeError(f"sharded bfs failed on rank {rank}: {failure}" if failure is not None else "sharded bfs failed on another rank")
            total_new = int(incl[-1])

            def new_before(tag):
                """number of new states of the chunk with a tag smaller than `tag` (global)"""
                q, a = (tag - 12 * c0) // 12, (tag - 12 * c0) % 12
                if q >= c1 - c0:
                    return total_new
                return int(incl[q] - pop12[mask[q]] + pop12[mask[q] & ((1 << a) - 1)])

            p_end, budget_hit = c1 - 1, False
            if nodes_global >= max_nodes:          # only the very first parent can see this (budget <= 1)
                p_end, budget_hit = c0, True
            elif nodes_global + total_new >= max_nodes:
                # parent of the new state that reaches the budget = first parent whose inclusive count reaches it
                pb = c0 + int(torch.searchsorted(incl, i64([max_nodes - nodes_global]))[0])
                if pb <= p_end:
                    p_end, budget_hit = pb, True
            is_solved = solved_tag < INF and solved_tag // 12 <= p_end
            if err_word < INF and (err_word >> 8) // 12 <= p_end and not (is_solved and solved_tag < (err_word >> 8)):
                # the reference executes this move before it stops: its ACMove raises (every rank sees the same words)
                raise AssertionError("a move emptied a relator during the search: the reference's ACMove raises here")
            if is_solved:
                gp = solved_tag // 12
                k = int(torch.searchsorted(f_gpos, i64([gp]))[0]) if f_gpos.numel() else 0
                mine = k < f_gpos.numel() and int(f_gpos[k]) == gp
                pref = i64([(rank << 40) | int(f_ids[k]) if mine else -1])
                comm.all_reduce(pref, "max")
                expanded += gp + 1 - c0
                nodes_global += new_before(solved_tag)
                return finish(True, walk(int(pref[0]), [(solved_tag % 12, 2)]))
            cutoff = 12 * (p_end + 1)
            first, cnt = engine.commit(cutoff)
            if cnt:
                next_ids.append(torch.arange(first, first + cnt, dtype=torch.int64, device=dev))
                mp = mask[par[:cnt]]
                next_gpos.append(next_count + incl[par[:cnt]] - pop12[mp] + pop12[mp & (torch.bitwise_left_shift(torch.ones_like(bit[:cnt]), bit[:cnt]) - 1)])
            committed = int(incl[p_end - c0])
            next_count += committed
            nodes_global += committed
            expanded += p_end + 1 - c0
            if budget_hit:
                return finish(False, None)
            c0 = c1
        f_ids = torch.cat(next_ids) if next_ids else i64([])
        f_gpos = torch.cat(next_gpos) if next_gpos else i64([])
        F = next_count
    return finish(False, None)
