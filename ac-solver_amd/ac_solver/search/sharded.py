"""Breadth-first search with the frontier sharded over the GPUs of one node -- the multi-GPU form of
`bfs` (reference: ac_solver/search/breadth_first.py:15-97), bit-identical to it for every world size.

States are partitioned by `hash(packed key) mod world`.  A level is processed in chunks of consecutive
global frontier positions; for each chunk every rank

  1. expands the frontier nodes it owns (HIP kernel, 12 children per node; a child's tag
     `12 * global_parent_position + action` is the order in which the reference generates it) and writes each
     child record straight into the send region of the owner of the child's key,
  2. ONE all-to-all (RCCL over xGMI),
  3. deduplicates what it received against its slice of the visited set, minimum tag wins (HIP), and sets bit a of
     a 12-bit mask of parent p for every child (p, a) that is a new state it owns,
  4. all-reduces (sum) those masks -- 4 bytes per PARENT -- so that every rank derives the same global FIFO
     numbering (position of a new state = new states of earlier parents + earlier set bits of its own parent),
     the same budget decision ("first parent after which len(tree_nodes) >= max_nodes", breadth_first.py:91-95)
     and the same success decision (smallest tag of a length-2 child, :84-85); no rank ever holds the tags of the
     other ranks' new states,
  5. turns its new states into nodes, numbered through the masks (no sort); they are its slice of the next level.
Per chunk the host reads back the send counts and ONE pack of decision scalars.

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

    def all_to_all_regions(self, regions):
        """regions[d]: the rows [m_d, C] int64 for rank d (views of the engine's send buffer) -> the rows received, grouped by
        source, in one tensor.  RCCL takes the regions as they lie (grouped send / recv, no staging copy); gloo (CPU tests) gets
        one contiguous buffer with split sizes."""
        torch = _torch()
        counts = [int(r.shape[0]) for r in regions]
        c_send = torch.tensor(counts, dtype=torch.int64, device=self.device)
        c_recv = torch.empty_like(c_send)
        self.dist.all_to_all_single(c_recv, c_send, group=self.group)
        recv_counts = c_recv.tolist()
        cols = regions[0].shape[1]
        recv = torch.empty((sum(recv_counts), cols), dtype=regions[0].dtype, device=self.device)
        if self.dist.get_backend(self.group) == "nccl":
            offs = np.concatenate([[0], np.cumsum(recv_counts)])
            self.dist.all_to_all([recv[offs[k]:offs[k + 1]] for k in range(self.world)], [r.contiguous() for r in regions], group=self.group)
        else:
            self.dist.all_to_all_single(recv, torch.cat(regions).contiguous(), output_split_sizes=recv_counts, input_split_sizes=counts, group=self.group)
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

    def all_to_all_regions(self, regions):
        return regions[0]

    def all_gather_var(self, t):
        return [t]

    def all_reduce(self, t, op):
        return t


# ---------------------------------------------------------------------------------------- engines ---
class HipShardEngine:
    """Per-rank frontier slice on one GPU: thin wrapper over the acx_shard_* C ABI (include/acx.h)."""

    def __init__(self, L, cyclical, node_cap, batch_cap, chunk_parents, rank, world, device=None):
        from ac_solver import _acx

        torch = _torch()
        self._acx = _acx
        _acx.require_device()
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.KW = _acx.lib.acx_shard_key_words(L)
        self.rank = rank
        with torch.cuda.device(self.device):
            self.h = _acx.lib.acx_shard_create(L, int(bool(cyclical)), int(node_cap), int(batch_cap), int(chunk_parents), rank, world)
        if not self.h:
            raise _acx.AcxError(f"acx_shard_create failed: {_acx.last_error()}")
        self.batch_cap = int(batch_cap)
        self._keep = None  # the received records stay alive until the commit that reads them

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

    def seed(self, record):
        rec = None if record is None else np.ascontiguousarray(record, np.int64)
        self._acx.check(self._acx.lib.acx_shard_seed(self.h, None if rec is None else self._acx.ptr(rec, C.c_int64), self._stream()), "acx_shard_seed")

    def level_begin(self):
        n = C.c_int64(0)
        self._acx.check(self._acx.lib.acx_shard_level_begin(self.h, C.byref(n)), "acx_shard_level_begin")
        return n.value

    def expand_routed(self, c0, c1, n_local, solved, world):
        """children of my frontier nodes with global position in [c0, c1): one region of records per destination rank (views)"""
        torch = _torch()
        full = 12 * min(c1 - c0, n_local)  # a region that could take every child
        # the owner hash spreads a large chunk evenly: regions are sized for 1.25 x the even share and the expansion is simply
        # repeated with full-size regions in the (never yet seen) case that one overflows -- the kernel only writes records
        # and min-combines `solved`, so a second run is harmless
        cap = full if world == 1 else min(full, int(1.25 * full / world) + 4096)
        for attempt in (0, 1):
            rec = torch.empty((world * cap, self.KW + 2), dtype=torch.int64, device=self.device)
            cnt = torch.empty(world, dtype=torch.int64, device=self.device)
            self._acx.check(self._acx.lib.acx_shard_expand_routed(self.h, int(c0), int(c1), rec.data_ptr() if cap else None, cap, cnt.data_ptr(),
                                                                  solved.data_ptr(), self._stream()), "acx_shard_expand_routed")
            if cap == 0:
                return [rec[:0] for _ in range(world)]
            counts = cnt.tolist()
            if max(counts) <= cap:
                return [rec[o * cap:o * cap + c] for o, c in enumerate(counts)]
            if cap == full:
                raise RuntimeError(f"rank {self.rank}: a send region overflowed ({max(counts)} > {cap} records)")
            cap = full

    def insert(self, recv, c0, n_parents):
        """-> int32 [n_parents]: bit a of entry p - c0 set when child (p, a) is a new state of this rank"""
        torch = _torch()
        n = recv.shape[0]
        recv = recv.contiguous()
        self._keep = recv
        mask = torch.empty(max(n_parents, 1), dtype=torch.int32, device=self.device)
        self._acx.check(self._acx.lib.acx_shard_insert(self.h, recv.data_ptr() if n else None, n, int(c0), int(n_parents), mask.data_ptr(), self._stream()),
                        "acx_shard_insert")
        return mask[:n_parents]

    def commit(self, cutoff, lmask, lprefix, gmask, gprefix, gpos_base, n_commit):
        p = lambda t: t.data_ptr() if t is not None and t.numel() else None  # noqa: E731
        self._acx.check(self._acx.lib.acx_shard_commit(self.h, int(cutoff), p(lmask), p(lprefix), p(gmask), p(gprefix), int(gpos_base), int(n_commit),
                                                       self._stream()), "acx_shard_commit")
        self._keep = None

    def find(self, gpos):
        out = C.c_int64(-1)
        self._acx.check(self._acx.lib.acx_shard_find(self.h, int(gpos), C.byref(out), self._stream()), "acx_shard_find")
        return out.value

    def node_info(self, node_id):
        info = np.zeros(3, np.int64)
        self._acx.check(self._acx.lib.acx_shard_node_info(self.h, int(node_id), self._acx.ptr(info, C.c_int64)), "acx_shard_node_info")
        return int(info[0]), int(info[1]), int(info[2])

    def status(self):
        err, ml = C.c_int32(0), C.c_int32(0)
        self._acx.check(self._acx.lib.acx_shard_status(self.h, C.byref(err), C.byref(ml)), "acx_shard_status")
        return err.value, ml.value


def _default_engine(L, cyclical, node_cap, batch_cap, chunk_parents, rank, world):
    return HipShardEngine(L, cyclical, node_cap, batch_cap, chunk_parents, rank, world)


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
    node_cap = (max_nodes + 12 * min(B, max_nodes) if world == 1 else int(2.0 * max_nodes / world)) + 4096
    engine = (engine_factory or _default_engine)(L, cyclically_reduce_after_moves, node_cap, batch_cap, B, rank, world)
    dev = getattr(engine, "device", torch.device("cpu"))
    KW = engine.KW
    exchange = world > 1 or _FORCE_EXCHANGE

    def i64(values):
        return torch.tensor(values, dtype=torch.int64, device=dev)

    pop12 = i64([bin(v).count("1") for v in range(4096)])  # popcount of a 12-bit child mask

    # root: node 0 of its owner, global frontier position 0
    root = engine.root_record(p)
    owner_root = int(owner_of(torch.tensor(root[None, :KW], dtype=torch.int64), world)[0])
    engine.seed(root if rank == owner_root else None)
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
        n_local = engine.level_begin()  # my slice of the level: the nodes I committed while the previous level was expanded
        next_count = 0
        c0 = 0
        while c0 < F:
            c1 = min(F, c0 + B)
            n_par = c1 - c0
            solved = i64([INF, INF])  # [0] smallest tag of a length-2 child, [1] smallest (tag << 8 | code) of a move the reference raises on
            # Local failures (a capacity of this rank's engine, a HIP error) must not leave the other ranks waiting in a
            # collective: the rank keeps taking part with empty contributions and reports through the `solved` all-reduce
            # (-1 beats every tag), so that all ranks raise together.
            failure = None
            empty = [torch.empty((0, KW + 2), dtype=torch.int64, device=dev) for _ in range(world)]
            regions = empty
            try:
                regions = engine.expand_routed(c0, c1, n_local, solved, world)
            except Exception as e:  # noqa: BLE001
                failure, regions = e, empty
            recv = comm.all_to_all_regions(regions) if exchange else regions[0]
            lmask = torch.zeros(n_par, dtype=torch.int32, device=dev)
            try:
                if failure is None:
                    if recv.shape[0] > engine.batch_cap:
                        raise RuntimeError(f"rank {rank}: {recv.shape[0]} records exceed the per-chunk capacity {engine.batch_cap}")
                    lmask = engine.insert(recv, c0, n_par)   # one 12-bit mask per parent: MY new states
            except Exception as e:  # noqa: BLE001
                failure, lmask = e, torch.zeros(n_par, dtype=torch.int32, device=dev)
            if failure is not None:
                solved = i64([-1, INF])
            # every (parent, action) child has exactly one owner, so SUM == OR
            gmask = lmask.clone()
            comm.all_reduce(gmask, "sum")
            comm.all_reduce(solved, "min")
            lpop, gpop = pop12[lmask.to(torch.int64)], pop12[gmask.to(torch.int64)]
            lincl, gincl = torch.cumsum(lpop, 0), torch.cumsum(gpop, 0)   # new states up to and including each parent
            # ONE read-back for every decision of the chunk: the success / error words, the number of new states, the first
            # parent whose inclusive count reaches what is left of the budget, and the counts a cut at that parent (or at the
            # first parent) would commit
            need = max(max_nodes - nodes_global, 0)
            pb = torch.clamp(torch.searchsorted(gincl, i64([need])), max=n_par - 1)
            pack = torch.cat([solved, gincl[-1:], pb, gincl[pb], lincl[pb], lincl[-1:], gincl[:1], lincl[:1]]).tolist()
            solved_tag, err_word, total_new, pb_rel, g_at_pb, l_at_pb, l_total, g_first, l_first = (int(v) for v in pack)
            if solved_tag < 0:
                raise RuntimeError(f"sharded bfs failed on rank {rank}: {failure}" if failure is not None else "sharded bfs failed on another rank")

            p_end, budget_hit = c1 - 1, False
            commit_global, commit_local = total_new, l_total
            if nodes_global >= max_nodes:          # only the very first parent can see this (budget <= 1)
                p_end, budget_hit, commit_global, commit_local = c0, True, g_first, l_first
            elif nodes_global + total_new >= max_nodes:
                # parent of the new state that reaches the budget = first parent whose inclusive count reaches it
                p_end, budget_hit, commit_global, commit_local = c0 + pb_rel, True, g_at_pb, l_at_pb
            is_solved = solved_tag < INF and solved_tag // 12 <= p_end
            if err_word < INF and (err_word >> 8) // 12 <= p_end and not (is_solved and solved_tag < (err_word >> 8)):
                # the reference executes this move before it stops: its ACMove raises (every rank sees the same words)
                raise AssertionError("a move emptied a relator during the search: the reference's ACMove raises here")
            if is_solved:
                gp = solved_tag // 12
                mine = engine.find(gp)
                pref = i64([(rank << 40) | mine if mine >= 0 else -1])
                comm.all_reduce(pref, "max")
                q, a = gp - c0, solved_tag % 12
                before = int(gincl[q] - gpop[q] + pop12[int(gmask[q]) & ((1 << a) - 1)])  # new states with a smaller tag (global)
                expanded += gp + 1 - c0
                nodes_global += before
                return finish(True, walk(int(pref[0]), [(a, 2)]))
            # a capacity failure of one rank's commit goes through the same "everybody raises" path as above
            failure = None
            try:
                engine.commit(12 * (p_end + 1), lmask, lincl - lpop, gmask, gincl - gpop, next_count, commit_local)
            except Exception as e:  # noqa: BLE001
                failure = e
            ok_all = i64([0 if failure is None else 1])
            if exchange:
                comm.all_reduce(ok_all, "max")
            if int(ok_all[0]):
                raise RuntimeError(f"sharded bfs failed on rank {rank}: {failure}" if failure is not None else "sharded bfs failed on another rank")
            next_count += commit_global
            nodes_global += commit_global
            expanded += p_end + 1 - c0
            if budget_hit:
                return finish(False, None)
            c0 = c1
        F = next_count
    return finish(False, None)
