"""Breadth-first search with the frontier sharded over the GPUs of one node -- the multi-GPU form of
`bfs` (reference: ac_solver/search/breadth_first.py:15-97), bit-identical to it for every world size.

States are partitioned by `hash(packed key) mod world`.  A level is processed in chunks of consecutive
global frontier positions; for each chunk every rank

  1. expands the frontier nodes it owns (HIP kernel, 12 children per node; a child's tag
     `12 * global_parent_position + action` is the order in which the reference generates it) and writes each
     child record straight into the send region of the owner of the child's key,
  2. ONE all-to-all with equal splits (RCCL over xGMI): every region starts with a small header -- the number of
     records in it and the sender's success / error / failure words -- so the exchange needs no count round trip,
  3. deduplicates what it received against its slice of the visited set, minimum tag wins (HIP), and sets bit a of
     a 12-bit mask of parent p for every child (p, a) that is a new state it owns,
  4. all-reduces (sum) those masks -- 2 bytes per PARENT, two masks to a word -- so that every rank derives the same global FIFO
     numbering (position of a new state = new states of earlier parents + earlier set bits of its own parent),
     the same budget decision ("first parent after which len(tree_nodes) >= max_nodes", breadth_first.py:91-95)
     and the same success decision (smallest tag of a length-2 child, :84-85); no rank ever holds the tags of the
     other ranks' new states,
  5. takes those decisions ON THE DEVICE and turns its new states into nodes, numbered through the masks (no
     sort); they are its slice of the next level.

Round 3: nothing is read back inside a chunk.  The host enqueues chunk after chunk and looks at a snapshot of the
engine's control block `LAG` chunks late; when the status word has left 0 (solved / budget / error / a failed rank)
the chunks that were already enqueued are no-ops on the device -- their collectives still pair up, because every rank
sees the same status at the same chunk -- and the loop ends.  One synchronisation per LEVEL remains (the size of the
next level).

Round 4: a child whose key is owned by the rank that generates it claims its table slot inside the expansion kernel and never
becomes a record (all children at world 1; 1/world of them otherwise), so step 1 writes the visited table -- the orchestrator
therefore keeps the expansion at most ONE chunk ahead of the dedup (an event behind each commit).  Every level ends with one small
all-reduce (max) of [failure code, smallest length generated, fullest region]: a failure that no header carried ends every rank at
the same point.  `TorchDistComm(mask_group="own")` puts step 4 on a communicator of its own; `timeline=True` returns the device
time of every stage of a chunk.

The per-rank work goes through an *engine* (the C ABI `acx_shard_*` of libacx in production; the CPU tests plug in a
NumPy engine built on the oracle, tests/shard_helpers.py) and the exchange through a *comm* (torch.distributed:
backend "nccl" is RCCL on ROCm; "gloo" in the CPU tests; an in-process thread communicator lets one GPU play several
ranks in the GPU tests).
"""
import ctypes as C
import sys

import numpy as np

INF = 1 << 62
HDR = 4              # int64 header words of a region: records written, success tag, error word, failure code (tests parse regions with it)
LAG = 2              # chunks the host runs ahead of the control block it reads
_FORCE_EXCHANGE = False  # tests: route through the communicator even when world == 1
_FORCE_LAG = None        # tests: run the CPU engines with the GPU path's lagged control block
_CHECK_OWNERS = False    # tests: stats["owner_mismatches"] = nodes of this rank's engine that do not live on their owner (acx_shard_check_owners)
_ID_MASK = (1 << 40) - 1
# control block words (include/acx.h: ACX_SHARD_CTL_*)
CTL_WORDS = 16
CTL_STATUS, CTL_NODES_GLOBAL, CTL_NEXT_COUNT, CTL_EXPANDED, CTL_SOLVED_TAG, CTL_NODES = 0, 1, 2, 3, 4, 5
CTL_FAIL_LOCAL, CTL_MIN_LEN, CTL_FAIL_SEEN, CTL_LEVEL_FILL = 8, 9, 10, 12
FILL_DEFAULT = 320  # region capacity in 1/256 of the even share of all children: 1.25 x (acx_shard_layout)
FILL_HARD = 1 << 20  # the hard bound: every workgroup sends a region all it has -- cannot overflow, world^2 x the even share
# Levels of fewer parents than this are expanded by EVERY rank (no collectives); the first level at least this large is partitioned by
# owner.  What a level costs a rank: replicated F parents x ~0.2 ns (the world-1 kernels: 9.3 ms per 4.5e7 parents) + six launches; sharded
# F / world x that + ONE all-to-all + TWO all-reduces (>= 120-210 us at any size, tools/shard_host_cost.py) + eight launches: at 8 ranks the
# two meet near 2^19-2^20 parents; 2^18 keeps the replicated share of a 1e8-node search below 1 % of its expansions.
REPLICATE_BELOW = 1 << 18
ST_RUNNING, ST_SOLVED, ST_BUDGET, ST_MOVE_ERROR, ST_FAILED = 0, 1, 2, 3, 4
_FAIL_TEXT = {1: "a send region overflowed (the record log never does: the host grows it before a chunk is expanded)", 2: "node capacity exceeded", 3: "visited table full", 4: "engine call failed"}


def _torch():
    import torch

    return torch


# ------------------------------------------------------------------------------------------ comms ---
class TorchDistComm:
    """torch.distributed communicator (nccl == RCCL on ROCm, gloo on CPU).  Both collectives of a chunk are enqueued on the
    current stream and need no host synchronisation (RCCL); gloo (CPU tensors) blocks, which is what the CPU tests want."""

    def __init__(self, device, group=None, mask_group=None):
        """`mask_group`: the group (communicator) the per-chunk mask all-reduce runs on.  torch's NCCL backend keeps ONE internal
        stream per communicator and runs that communicator's collectives in issue order; the orchestrator issues the all-to-all of
        chunk k + 1 (side stream) BEFORE the mask all-reduce of chunk k (main stream), so on one communicator all-reduce(k) queues
        behind all-to-all(k + 1), which waits for expand(k + 1) -- the commit of chunk k then waits for the exchange it was meant to
        overlap.  "own" (or a group from dist.new_group) gives the masks their own communicator: the two collectives of a chunk
        period are then independent.  Same issue order on every rank either way.  None = the same group as everything else."""
        import torch.distributed as dist

        self.dist, self.group, self.device = dist, group, device
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        if isinstance(mask_group, str):
            assert mask_group in ("own", "shared"), mask_group
            mask_group = dist.new_group(ranks=list(range(self.world)) if group is None else dist.get_process_group_ranks(group)) if mask_group == "own" else None
        self.mask_group = mask_group if mask_group is not None else group
        self.stats = {"all_to_all_calls": 0, "all_to_all_bytes": 0, "all_reduce_calls": 0, "all_reduce_bytes": 0, "mask_all_reduce_calls": 0, "mask_all_reduce_bytes": 0}

    def all_to_all_single(self, recv, send):
        """equal splits: rank d receives send[d * k : (d + 1) * k] of every rank, k = numel / world"""
        assert recv.numel() == send.numel() and send.numel() % self.world == 0
        if recv.data_ptr() == send.data_ptr():  # world 1 (tests, bench.py's forced one-rank run): the engine's send and receive areas are the same, and RCCL rejects aliased buffers
            send = send.clone()
        self.dist.all_to_all_single(recv, send, group=self.group)
        self.stats["all_to_all_calls"] += 1
        self.stats["all_to_all_bytes"] += send.numel() * send.element_size()

    def all_reduce(self, t, op):
        ops = {"min": self.dist.ReduceOp.MIN, "max": self.dist.ReduceOp.MAX, "sum": self.dist.ReduceOp.SUM}
        self.dist.all_reduce(t, op=ops[op], group=self.group)
        self.stats["all_reduce_calls"] += 1
        self.stats["all_reduce_bytes"] += t.numel() * t.element_size()
        return t

    def all_reduce_masks(self, t):
        """the per-chunk child masks (sum == or: every (parent, action) child has one owner), on `mask_group`"""
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.mask_group)
        self.stats["mask_all_reduce_calls"] += 1
        self.stats["mask_all_reduce_bytes"] += t.numel() * t.element_size()
        return t


class SingleComm:
    """world == 1: no exchange at all (the engine expands straight into its receive area)."""

    rank, world = 0, 1

    def __init__(self):
        self.stats = {}

    def all_to_all_single(self, recv, send):
        if recv.data_ptr() != send.data_ptr():
            recv.copy_(send)

    def all_reduce(self, t, op):
        return t


# ---------------------------------------------------------------------------------------- engines ---
class HipShardEngine:
    """Per-rank frontier slice on one GPU: thin wrapper over the acx_shard_* C ABI (include/acx.h).  The record log, the
    send buffer and the mask buffer are torch tensors (torch.distributed moves them); the log grows by doubling when the
    host-side cursor says the next chunk would not fit."""

    def __init__(self, L, cyclical, node_cap, chunk_parents, rank, world, est_parents, device=None):
        from ac_solver import _acx

        torch = _torch()
        self._acx = _acx
        _acx.require_device()
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.KW = _acx.lib.acx_shard_key_words(L)
        self.RW = self.KW + 1
        self.rank, self.world, self.L = rank, world, int(L)
        self.B = int(chunk_parents)
        self.replicated = False  # set_replicated / partition: whole levels on every rank with the world-1 kernels, no exchange
        with torch.cuda.device(self.device):
            self.h = _acx.lib.acx_shard_create(L, int(bool(cyclical)), int(node_cap), self.B, rank, world)
            if not self.h:
                raise _acx.AcxError(f"acx_shard_create failed: {_acx.last_error()}")
            full = self.layout_words(self.B)
            words = 8 + int(full * (est_parents / self.B + 2)) + 64 * 40  # (+ the header-only regions of the replicated levels' chunks)
            self.log = torch.empty(words, dtype=torch.int64, device=self.device)
            self.send = torch.empty(full, dtype=torch.int64, device=self.device) if world > 1 else None
            self.gmask = torch.empty((self.B + 3) // 4 * 2, dtype=torch.int32, device=self.device)  # two parents' 12-bit masks per word; whole quads of parents (the kernels move four at a time)
            self._attach()
        self._cursor = 8  # host mirror of the engine's log cursor (deterministic: it advances by the chunk's words)
        self._ctl = np.zeros(CTL_WORDS, np.int64)

    def close(self):
        """Give the engine's device memory back NOW (bfs_sharded calls this when it returns) instead of whenever the garbage collector
        gets to the reference cycle the orchestrator's closures keep the engine in: the next search then finds the 5 GB of a
        1e8-node engine in the block pool."""
        if getattr(self, "h", None) and not sys.is_finalizing():
            _torch().cuda.synchronize(self.device)  # (a chunk that was expanded but never consumed may still be running)
            self._acx.lib.acx_shard_destroy(self.h)
            self.h = None
        self.log = self.send = self.gmask = None

    def __del__(self):
        # not during interpreter shutdown: the HIP runtime may already be tearing down (a hipFree then can block forever)
        if getattr(self, "h", None) and not sys.is_finalizing():
            self._acx.lib.acx_shard_destroy(self.h)
            self.h = None

    def _stream(self):
        return _torch().cuda.current_stream(self.device).cuda_stream

    def _attach(self):
        self._acx.check(self._acx.lib.acx_shard_attach(self.h, self.log.data_ptr(), self.log.numel(), self.send.data_ptr() if self.send is not None else None,
                                                       self.send.numel() if self.send is not None else 0, self.gmask.data_ptr()), "acx_shard_attach")

    def layout(self, n_par, fill_q8=0):
        s, cap, rw = C.c_int64(), C.c_int64(), C.c_int64()
        self._acx.check(self._acx.lib.acx_shard_layout(int(n_par), self.world_eff, self.KW, int(fill_q8), C.byref(s), C.byref(cap), C.byref(rw)), "acx_shard_layout")
        return s.value, cap.value, rw.value

    def layout_words(self, n_par, fill_q8=0):
        s, _, rw = self.layout(n_par, fill_q8)
        return s * self.world_eff * rw

    @property
    def world_eff(self):
        """the world size the chunk geometry is computed for: 1 during the replicated phase"""
        return 1 if self.replicated else self.world

    def set_replicated(self):
        """before the root is seeded, on every rank: the levels that follow are processed whole by every rank (acx_shard_set_replicated)"""
        self._acx.check(self._acx.lib.acx_shard_set_replicated(self.h, 1), "acx_shard_set_replicated")
        self.replicated = True

    def partition(self):
        """end of the replicated phase, between two levels: this rank's share of the newest level becomes its frontier slice"""
        try:
            self._acx.check(self._acx.lib.acx_shard_partition(self.h, self._stream()), "acx_shard_partition")
        finally:
            self.replicated = False

    def walk(self, node_id, cap=256):
        """-> (next parent reference that is not local, or -1 at the root; [(action, total length), ...] from node_id upwards)"""
        out = np.zeros(2 + 2 * cap, np.int64)
        self._acx.check(self._acx.lib.acx_shard_walk(self.h, int(node_id), int(cap), self._acx.ptr(out, C.c_int64), self._stream()), "acx_shard_walk")
        n = int(out[1])
        return int(out[0]), [(int(out[2 + 2 * k]), int(out[3 + 2 * k])) for k in range(n)]

    def root_record(self, presentation):
        rec = np.zeros(self.KW + 2, np.int64)
        p = self._acx.as_i8_rows(np.asarray(presentation))
        rc = self._acx.lib.acx_shard_root_record(self.h, self._acx.ptr(p, C.c_int8), self._acx.ptr(rec, C.c_int64))
        if rc == self._acx.E_ROWERR:
            raise AssertionError(self._acx.last_error())
        self._acx.check(rc, "acx_shard_root_record")
        return rec

    def root_owner(self, record):
        """rank that owns the root (csrc/acx_owner.h through acx_shard_owner: host arithmetic)"""
        rec = np.ascontiguousarray(record, np.int64)
        o = self._acx.lib.acx_shard_owner(self.L, self._acx.ptr(rec, C.c_int64), self.world)
        self._acx.check(min(o, 0), "acx_shard_owner")
        return o

    def check_owners(self):
        """test hook: local nodes that are not where acx_shard_owner says (must be 0)"""
        bad = C.c_int64(-1)
        self._acx.check(self._acx.lib.acx_shard_check_owners(self.h, C.byref(bad), self._stream()), "acx_shard_check_owners")
        return bad.value

    def seed(self, record):
        rec = None if record is None else np.ascontiguousarray(record, np.int64)
        self._acx.check(self._acx.lib.acx_shard_seed(self.h, None if rec is None else self._acx.ptr(rec, C.c_int64), self._stream()), "acx_shard_seed")

    def chunk_expand(self, c0, c1, level_first, fill_q8=0):
        """-> (send, recv): 1-D int64 views of equal length for the all-to-all (the same view at world 1)"""
        torch = _torch()
        need = self.layout_words(c1 - c0, fill_q8)
        if self._cursor + need > self.log.numel():  # grow the log: rare, so simply behind everything that is in flight on any stream
            torch.cuda.synchronize(self.device)
            with torch.cuda.device(self.device):
                bigger = torch.empty(max(2 * self.log.numel(), self._cursor + 2 * need), dtype=torch.int64, device=self.device)
                bigger[: self._cursor].copy_(self.log[: self._cursor])
                self.log = bigger
                self._attach()
            torch.cuda.synchronize(self.device)  # the copy ran on the calling stream; the other stream's next kernels read the new block
        if self.send is not None and not self.replicated and need > self.send.numel():  # (only under a capacity above the default: the hard bound of the last resort)
            torch.cuda.synchronize(self.device)
            with torch.cuda.device(self.device):
                self.send = torch.empty(need, dtype=torch.int64, device=self.device)
                self._attach()
        off, words = C.c_int64(), C.c_int64()
        self._acx.check(self._acx.lib.acx_shard_chunk_expand(self.h, int(c0), int(c1), int(bool(level_first)), int(fill_q8), C.byref(off), C.byref(words), self._stream()),
                        "acx_shard_chunk_expand")
        assert off.value == self._cursor and words.value == need, (off.value, self._cursor, words.value, need)
        recv = self.log[off.value: off.value + need]
        self._cursor += need
        return (recv if (self.send is None or self.replicated) else self.send[:need]), recv

    def chunk_insert(self, n_par):
        """-> the chunk's child masks, two parents per int32 word (parent p: bits 16 (p & 1) .. + 11 of word p >> 1), to be summed over the ranks"""
        self._acx.check(self._acx.lib.acx_shard_chunk_insert(self.h, self._stream()), "acx_shard_chunk_insert")
        return self.gmask[:(n_par + 1) // 2]

    def chunk_insert_dead(self, n_par):
        """the failure path: the chunk's masks without the dedup (acx_shard_chunk_insert_dead)"""
        self._acx.check(self._acx.lib.acx_shard_chunk_insert_dead(self.h, self._stream()), "acx_shard_chunk_insert_dead")
        return self.gmask[:(n_par + 1) // 2]

    def gmask_view(self, n_par):
        return self.gmask[:(n_par + 1) // 2]

    def chunk_commit(self, max_nodes):
        self._acx.check(self._acx.lib.acx_shard_chunk_commit(self.h, int(max_nodes), self._stream()), "acx_shard_chunk_commit")

    def ctl_snapshot(self, slot):
        self._acx.check(self._acx.lib.acx_shard_ctl_snapshot(self.h, int(slot), self._stream()), "acx_shard_ctl_snapshot")

    def ctl_wait(self, slot):
        self._acx.check(self._acx.lib.acx_shard_ctl_wait(self.h, int(slot), self._acx.ptr(self._ctl, C.c_int64)), "acx_shard_ctl_wait")
        return self._ctl.copy()

    def fail_local(self):
        self._acx.lib.acx_shard_fail(self.h, self._stream())

    def find(self, gpos):
        out = C.c_int64(-1)
        self._acx.check(self._acx.lib.acx_shard_find(self.h, int(gpos), C.byref(out), self._stream()), "acx_shard_find")
        return out.value

    def node_info(self, node_id):
        info = np.zeros(3, np.int64)
        self._acx.check(self._acx.lib.acx_shard_node_info(self.h, int(node_id), self._acx.ptr(info, C.c_int64)), "acx_shard_node_info")
        return int(info[0]), int(info[1]), int(info[2])


def _default_engine(L, cyclical, node_cap, chunk_parents, rank, world, est_parents):
    return HipShardEngine(L, cyclical, node_cap, chunk_parents, rank, world, est_parents)


# ------------------------------------------------------------------------------------- orchestrator ---
_CLASS_K = (0x85EBCA6B, 0xC2B2AE35, 0x27D4EB2F, 0x165667B1, 0xD3A2646D, 0xFD7046C5, 0xB55A4F09, 0x9E3779B9,
            0x7F4A7C15, 0x94D049BB, 0xBF58476D, 0x1CE4E5B9, 0x2545F491, 0x4F6CDD1D, 0x6C62272F, 0x07BB0143)


def conj_prefix(codes):
    """letters the cyclic reduction strips from each end (acx_word.h:cyclic_reduce): p = letters on which the word and its inverse
    agree from the front -- for a freely reduced word r = u c u^-1 that is |u| and stops before the middle --, 0 when that would
    eat the whole word"""
    n = len(codes)
    p = 0
    while p < n and codes[p] == (codes[n - 1 - p] ^ 3):
        p += 1
    return 0 if (p == n or 2 * p >= n) else p


def inner_letter(codes):
    """csrc/acx_owner.h: 0 when u is empty (r = u c u^-1), else 1 + the code of u's last letter.  A conjugation of r by a generator
    adds or removes a letter at the FRONT of u: this only changes while |u| <= 1."""
    p = conj_prefix(codes)
    return 1 + codes[p - 1] if p else 0


def class_hash(codes):
    """csrc/acx_owner.h: class_hash on the 2-bit letter codes of one relator (-2, -1, +1, +2 -> 0, 1, 2, 3; inverse = code ^ 3).
    A function of the word's cyclic reduction read as a CYCLIC word -- its length and the multiset of (letter, cyclic successor)
    pairs -- so the eight conjugation moves of ACMove (ac_moves.py:192-229) do not change it."""
    p = conj_prefix(codes)
    c = codes[p:len(codes) - p]
    n = len(c)
    if n == 0:
        return 0
    h = (n * 0x9E3779B1) & 0xFFFFFFFF
    for i in range(n):
        x, y = c[i], c[(i + 1) % n]
        if x ^ y != 3:
            h = (h + _CLASS_K[4 * x + y]) & 0xFFFFFFFF
    return h


def owner_of(keys, world):
    """Owner rank of each packed key (rows of KW int64 words: 2 for L <= 29 -- word | length << 58 per relator --, 4 for L <= 61:
    two words per relator, the length in the top 6 bits of the second).  The same arithmetic as csrc/acx_owner.h, which
    `acx_shard_owner` exposes (tests/test_sharded_cpu.py compares the two):
    scale(mix(class_hash(r_0) + class_hash(r_1) + K0 inner_letter(r_0) + K1 inner_letter(r_1)), world).
    A rank therefore owns most conjugation children of its own nodes, and most children never cross the exchange."""
    rows = np.asarray(keys.cpu() if hasattr(keys, "cpu") else keys, dtype=np.int64).reshape(len(keys), -1)
    half = rows.shape[1] // 2
    bits = 64 * half
    out = np.zeros(len(rows), np.int64)
    for i, row in enumerate(rows):
        csum = 0
        for r in range(2):
            k = 0
            for j in range(half):
                k |= (int(row[r * half + j]) & 0xFFFFFFFFFFFFFFFF) << (64 * j)
            n = k >> (bits - 6)
            codes = [(k >> (2 * t)) & 3 for t in range(n)]
            csum += class_hash(codes) + inner_letter(codes) * (0x85EBCA77 if r else 0x9E3779B1)
        h = ((csum & 0xFFFFFFFF) * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        h ^= h >> 29
        out[i] = (((h >> 20) & 0xFFFFFFFF) * int(world)) >> 32
    return out


_SIDE_STREAMS = {}


def _side_stream(dev):
    """ONE side stream per device for the life of the process.  `torch.cuda.Stream()` hands out the 32 streams of torch's pool in
    turn, and the first submission to each of them pays for a hardware queue: a search that took a new stream per call ran
    20 ms instead of 15 ms at 1e8 nodes."""
    torch = _torch()
    key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device())
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(dev)
    return _SIDE_STREAMS[key]


class _RegionOverflow(RuntimeError):
    """a send region overflowed under a capacity tighter than the default: the search is rerun with the default"""


def bfs_sharded(presentation, max_nodes_to_explore=10000, verbose=False, cyclically_reduce_after_moves=False, comm=None,
                engine_factory=None, batch_parents=None, want_stats=False, log_fraction=0.5, overlap=None, region_fill=None, timeline=False,
                replicate_below=None):
    """`bfs` with the frontier sharded over the ranks of `comm`: see _bfs_sharded_once.  A search whose adaptive (or given) region
    capacity turns out too tight fails on every rank at the same chunk and is rerun from scratch with the safe default."""
    kw = dict(verbose=verbose, cyclically_reduce_after_moves=cyclically_reduce_after_moves, comm=comm, engine_factory=engine_factory,
              want_stats=want_stats, log_fraction=log_fraction, overlap=overlap, timeline=timeline, replicate_below=replicate_below)
    # (the owner function keeps families of states on one rank -- csrc/acx_owner.h --, so no capacity below the hard bound is safe
    # for EVERY input: the third attempt uses it, with small chunks, since a region then has room for every child of its senders)
    if batch_parents is None:
        # a chunk's fixed costs (eight launches per rank) against the memory of its flag arrays: 2^21 global parents, more from 8 ranks
        # on, where a rank's share of a chunk is small (profiles/r5_shard_thread_ranks_device_work.txt: 3.7 -> 3.4 ms per rank at 8 with 2^22)
        # (round 6: the flags of a chunk are one word per parent instead of a byte per child -- 16 B of zeroed memory per parent, not 48 -- so a larger
        # chunk no longer pays for itself in fills: 2^23 from 8 ranks on, profiles/r6_shard_thread_ranks_device_work.txt)
        batch_parents = 1 << (23 if (comm is not None and comm.world >= 8) else 21)
    reruns = 0
    tried = []
    for fill, bp in ((region_fill, batch_parents), (FILL_DEFAULT, batch_parents), (FILL_HARD, min(batch_parents, 1 << 17))):
        if (fill, bp) in tried:  # (a caller who asked for the default itself: the identical attempt would overflow again)
            continue
        tried.append((fill, bp))
        try:
            out = _bfs_sharded_once(presentation, max_nodes_to_explore, region_fill=fill, **dict(kw, batch_parents=bp))
        except _RegionOverflow:
            reruns += 1
            continue
        if want_stats and reruns:
            out[2]["region_overflow_reruns"] = reruns
        return out
    raise RuntimeError("sharded bfs: a region overflowed under the hard capacity bound")  # (cannot happen: acx_shard_layout)


def _bfs_sharded_once(presentation, max_nodes_to_explore=10000, verbose=False, cyclically_reduce_after_moves=False, comm=None,
                      engine_factory=None, batch_parents=1 << 21, want_stats=False, log_fraction=0.5, overlap=None, region_fill=None, timeline=False,
                      replicate_below=None):
    """Same contract as `bfs`: returns (is_search_successful, path or None) [+ stats dict], identical on every rank.
    `batch_parents`: global frontier positions per chunk.  `log_fraction`: expected expanded parents / max_nodes, sizes the
    record log (it grows by doubling if the estimate is short).  `overlap`: expansion + exchange of chunk k + 1 on a side stream, started beside the dedup of chunk k ("insert": the default when there is an exchange, i.e. world > 1) or beside its commit ("commit"); False: one stream, the default without an exchange.  `region_fill`: capacity of the exchanged regions in 1/256 of the even share of all children -- None: adaptive (1.25 x the fullest region of the previous level, at most the default), an int: that value for every chunk, FILL_DEFAULT: 1.25 x the even share of all children, FILL_HARD: the bound that cannot overflow; a search whose regions overflow is rerun with the default, then with the hard bound.  `timeline` (GPU, with want_stats): HIP events around every stage of every chunk -- stats["timeline"] gives the median device time of
    expansion, all-to-all, dedup, mask all-reduce and commit per full-size chunk, the chunk period and `overlap_effective` = their
    sum / the period (1 = the stages run one after the other, > 1 = the side stream hides work).  A diagnostic: the events cost a
    little, so timed runs leave it off.  `replicate_below` (world > 1): levels of fewer parents than this are processed WHOLE by every
    rank with the world-1 kernels -- no all-to-all, no mask all-reduce, no closing all-reduce: a small level costs a rank a few
    microseconds of expansion against >= 100 us of collectives --; at the first level that is at least this large the frontier is
    partitioned by owner (engine.partition) and the chunks are exchanged.  One all-reduce closes the replicated phase (failure code:
    a rank whose engine raised ends every rank there).  None: REPLICATE_BELOW; 0 or 1: every level is exchanged (rounds 3-5)."""
    from ac_solver.envs.utils import is_array_valid_presentation

    import time

    torch = _torch()
    t_begin = time.perf_counter()
    assert is_array_valid_presentation(presentation), f"{presentation} is not a valid presentation"
    p = np.array(presentation, dtype=np.int8)
    L = len(p) // 2
    from ac_solver.search._common import _check_width

    _check_width(L, p)
    max_nodes = int(max_nodes_to_explore)
    comm = SingleComm() if comm is None else comm
    world, rank = comm.world, comm.rank
    B = int(max(1, min(batch_parents, max(max_nodes, 64))))  # global parents per chunk
    replicate_below = REPLICATE_BELOW if replicate_below is None else int(replicate_below)
    if world == 1:
        replicate_below = 0
    # (the nodes of the replicated levels live on every rank, and so does a copy of a rank's share of the level the partition happens at:
    # the levels below `replicate_below` parents hold < 2 x that many nodes unless the search hardly grows, the level after them < 12 x)
    node_cap = (max_nodes + 64 if world == 1 else int(2.0 * max_nodes / world) + min(max_nodes, 16 * max(replicate_below, 0))) + 4096
    engine = (engine_factory or _default_engine)(L, cyclically_reduce_after_moves, node_cap, B, rank, world, max(1.0, log_fraction * max_nodes))
    # The orchestrator allocates a few small Python objects per chunk; in a process with a large heap (bench.py) that now and then
    # triggers a full garbage collection (25-30 ms there) in the middle of a 13 ms search.  No collections while the search runs
    # (the collector's state is restored on the way out: the collection then happens behind the search).
    import gc

    gc_was_on = gc.isenabled()
    gc.disable()
    try:
        return _bfs_sharded_run(engine, p, L, max_nodes, B, comm, world, rank, verbose, want_stats, overlap, region_fill, t_begin, timeline, replicate_below)
    finally:
        if hasattr(engine, "close"):
            engine.close()
        if gc_was_on:
            gc.enable()


def _timeline_summary(torch, rows, B):
    """rows: per chunk (n_par, side events e0 e1 e2, main events m0 m1 m2 m3) -> medians over the full-size chunks (all chunks if none is)"""
    torch.cuda.synchronize()
    any_full = any(r[0] == B for r in rows)
    full = [r for r in rows if r[0] == B] if any_full else rows
    med = lambda v: float(sorted(v)[len(v) // 2]) if v else None  # noqa: E731
    us = lambda a, b: a.elapsed_time(b) * 1e3  # noqa: E731
    stages = {"expand_us": [us(r[1], r[2]) for r in full], "all_to_all_us": [us(r[2], r[3]) for r in full], "insert_us": [us(r[4], r[5]) for r in full],
              "mask_all_reduce_us": [us(r[5], r[6]) for r in full], "commit_us": [us(r[6], r[7]) for r in full],
              "wait_for_exchange_us": [max(0.0, us(r[3], r[4])) for r in full]}
    # period: from the end of one chunk's commit to the end of the next one's, between consecutive full-size chunks of a level
    period = [us(a[7], b[7]) for a, b in zip(rows, rows[1:]) if (not any_full or (a[0] == B and b[0] == B)) and a[8] == b[8]]
    out = {k: med(v) for k, v in stages.items()}
    out["chunk_period_us"] = med(period)
    out["chunks_timed"] = len(full)
    out["chunk_parents"] = full[0][0] if full else 0
    parts = [out[k] for k in ("expand_us", "all_to_all_us", "insert_us", "mask_all_reduce_us", "commit_us")]
    if out["chunk_period_us"] and all(v is not None for v in parts):
        out["stage_sum_us"] = sum(parts)
        out["overlap_effective"] = sum(parts) / out["chunk_period_us"]
    return out


def _bfs_sharded_run(engine, p, L, max_nodes, B, comm, world, rank, verbose, want_stats, overlap, region_fill, t_begin, timeline=False, replicate_below=0):
    import time

    torch = _torch()
    dev = getattr(engine, "device", torch.device("cpu"))
    KW = engine.KW
    # the replicated phase (small levels whole on every rank): from the root until the first level of >= replicate_below parents
    replicating = world > 1 and replicate_below > 1 and hasattr(engine, "set_replicated")
    exchange = (world > 1 or _FORCE_EXCHANGE) and not replicating  # flips to True at the partition
    lag = LAG if dev.type == "cuda" else 0
    if _FORCE_LAG is not None:
        lag = int(_FORCE_LAG)
    reduce_masks = getattr(comm, "all_reduce_masks", None) or (lambda t: comm.all_reduce(t, "sum"))

    def i64(values):
        return torch.tensor(values, dtype=torch.int64, device=dev)

    # root: node 0 of its owner, global frontier position 0
    root = engine.root_record(p)
    if replicating:
        engine.set_replicated()
        engine.seed(root)  # every rank holds the replicated levels
    else:
        owner_root = int(engine.root_owner(root) if hasattr(engine, "root_owner") else owner_of(root[None, :KW], world)[0])
        engine.seed(root if rank == owner_root else None)
    F = 1
    repl_levels = 0
    levels = chunks = 0
    failure = None         # the first exception on this rank's host side
    fail_hdr_chunk = None  # index (inside the running level) of the first chunk this rank sent with "failed" headers
    ctl = None
    t_ready = time.perf_counter()  # engine built (device allocations, table fill), root seeded

    # capacity of the exchanged regions (acx_shard_layout): the safe default, a given value, or -- adaptive -- 1.3 x the fullest
    # region of the previous level (the maximum over the ranks: it rides on the closing all-reduce of the level)
    adaptive = region_fill is None and exchange
    fill = FILL_DEFAULT if region_fill is None else int(region_fill)
    fills = []
    tight_used = False  # some chunk of this search was expanded with less than the default capacity

    def raise_failed(code):
        """every rank gets here at the same point with the same code (the largest any rank reported)"""
        if code == 1 and failure is None and tight_used:  # (an overflow shows one chunk late: by then `fill` may be the next level's)
            raise _RegionOverflow()
        mine = failure is not None or (ctl is not None and int(ctl[CTL_FAIL_LOCAL]))
        raise RuntimeError(f"sharded bfs failed on rank {rank}: {failure or _FAIL_TEXT.get(int(ctl[CTL_FAIL_LOCAL]) if ctl is not None else 4, 'engine failure')}"
                           if mine else f"sharded bfs failed on another rank: {_FAIL_TEXT.get(code, 'engine failure')}")

    WALK_CAP = 256

    def walk(pref, tail, collective=True):
        """path of the node `pref` (rank << 40 | id) from the root + tail.  The owner of a node walks up for as long as the parents are
        its own (ONE launch and one copy per segment, engine.walk) and shares the segment with ONE all-reduce; the owner function keeps
        three quarters of the children on their parent's rank, and the nodes of the replicated levels name the rank that holds them, so a
        path is a handful of segments.  `collective` False (the search ended in its replicated phase): every rank walks its own copy."""
        rev = []
        while pref >= 0:
            r, nid = pref >> 40, pref & _ID_MASK
            if hasattr(engine, "walk"):
                buf = np.zeros(2 + 2 * WALK_CAP, np.int64)
                if rank == r or not collective:
                    nxt, pairs = engine.walk(nid, WALK_CAP)
                    buf[0], buf[1] = nxt, len(pairs)
                    buf[2:2 + 2 * len(pairs)] = np.asarray(pairs, np.int64).reshape(-1)
                if collective:
                    t = i64(buf.tolist())
                    comm.all_reduce(t, "sum")
                    buf = np.asarray(t.tolist(), np.int64)
                n = int(buf[1])
                rev.extend((int(buf[2 + 2 * k]), int(buf[3 + 2 * k])) for k in range(n))
                pref = int(buf[0])
                continue
            info = i64(list(engine.node_info(nid)) if (rank == r or not collective) else [0, 0, 0])
            if collective:
                if rank != r:
                    info[2] = 0
                comm.all_reduce(info, "sum")
            a, tl, pref = (int(v) for v in info.tolist())
            rev.append((a, tl))
        return rev[::-1] + tail

    tl_rows = [] if (timeline and dev.type == "cuda") else None

    def finish(ok, path, min_len):
        if want_stats:
            st = dict(nodes=int(ctl[CTL_NODES_GLOBAL]), expanded=int(ctl[CTL_EXPANDED]), levels=levels, chunks=chunks, min_len=2 if ok else min_len, world=world)
            st.update({"comm_" + k: v for k, v in getattr(comm, "stats", {}).items()})
            st.update(setup_seconds=t_ready - t_begin, loop_seconds=time.perf_counter() - t_ready, region_fill_q8=fills[-4:] if adaptive else fill, region_fill_all=list(fills) if adaptive else fill)
            if tl_rows:
                st["timeline"] = _timeline_summary(torch, tl_rows, B)
            st["local_nodes"] = int(ctl[CTL_NODES])
            st["replicated_levels"] = repl_levels
            if _CHECK_OWNERS and hasattr(engine, "check_owners"):
                st["owner_mismatches"] = engine.check_owners()
            return ok, path, st
        return ok, path

    # Two streams on a GPU: `side` runs the expansion of chunk k + 1 and its all-to-all beside the main stream's work on chunk k.
    # Inside a level chunk k + 1 needs nothing from chunk k: its parents were committed during the previous level and it has its
    # own slice of the record log; only the send buffer is shared, and that is reused in `side`'s own order.  The CPU engines of
    # the tests run everything in program order.  What running two of the engine's kernels side by side is worth on ONE GPU
    # (measured in round 3 on rocprofv3 kernel traces of 1e8-node searches):
    #   * expansion beside k_shard_insert: the dedup keeps ~2.6e5 table atomics queued at the memory side, the expansion's
    #     region reservations (one returning atomic per workgroup) wait ~14 us each behind them and it loses its share of the
    #     compute units: 0.21 -> 0.75-0.9 ms per 2^21-parent chunk, the dedup slows by a third, the chunk period stays;
    #   * expansion beside pack / scan / decide / commit ("commit"): 0.21 -> 0.33 ms, the commit slows too: no gain either;
    #   * k_shard_commit of chunk k beside the dedup of chunk k + 1 (tried with a third stream): no gain.
    # The kernels are bound by vector issue (expansion) or by the memory system (the rest); what shortens the search is
    # shortening them.  So without an exchange everything runs in stream order; with one, expansion + all-to-all start at once
    # on the side stream ("insert"): the per-rank kernels shrink with the world size and the collectives are what has to be hidden.
    if overlap is None or overlap is True:
        overlap = "insert" if (world > 1 or _FORCE_EXCHANGE) else False
    assert overlap in (False, "insert", "commit"), overlap
    on_gpu = dev.type == "cuda"
    main = torch.cuda.current_stream(dev) if on_gpu else None
    side = (_side_stream(dev) if overlap else main) if on_gpu else None

    class _on_side:
        def __enter__(self):
            if on_gpu:
                self.ctx = torch.cuda.stream(side)
                self.ctx.__enter__()

        def __exit__(self, *a):
            if on_gpu:
                self.ctx.__exit__(*a)

    def stamp(stream):
        if tl_rows is None:
            return None
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(stream)
        return ev

    def set_failed(e):
        nonlocal failure
        failure = failure or e
        try:
            engine.fail_local()  # sticky on the device: the headers of every later chunk this engine expands say so
        except Exception:  # noqa: BLE001
            pass

    # A failure on this rank's HOST side (an engine call that raises: a HIP error, an exhausted allocation) must not leave the
    # other ranks waiting in a collective, and must not leave THIS rank issuing collectives the others have stopped issuing.
    #   * From the failure on, every chunk this rank produces is "dead": no engine call, empty regions whose headers carry failure
    #     code 4.  Every healthy rank's k_shard_decide turns its status to 4 at the first dead chunk (index `fail_hdr_chunk` of the
    #     level) and leaves the chunk loop once it has read that control block, `lag` chunks later.  This rank cannot see that in
    #     its own control block (a dead chunk does not go through its engine), so it leaves by the same rule: after the iteration
    #     of chunk fail_hdr_chunk + lag -- every rank has then issued the same collectives.
    #   * Chunks that were expanded by the engine before the failure are still dedup'ed and committed through it (a dedup call that
    #     raised is replaced by `chunk_insert_dead`: masks only), so the engine's ring of open chunks stays in step and its decide
    #     kernel keeps taking the global decisions (budget / solved / a failure seen) from the all-reduced masks: the control
    #     blocks this rank reads for chunks before the first dead one are as valid as every other rank's.
    #   * The level's closing all-reduce carries the failure code: a failure in the last chunk of a level, or one that no header
    #     carried, ends every rank there, whatever status each of them read.
    #   * The control-block calls of a failed rank may raise as well (a sticky HIP error): they are guarded, the rank keeps the last
    #     block it read (or a synthetic "running" one) and leaves the loop by the fail_hdr_chunk + lag rule and the closing all-reduce
    #     alone.  (engine.layout is host arithmetic on (parents, world, key words): it cannot fail with the device.)
    bad_slots = set()  # a snapshot that was never taken leaves WHATEVER an earlier chunk (or search: the pinned slots are pooled) wrote in its slot

    def ctl_snapshot(slot):
        bad_slots.discard(slot)
        try:
            engine.ctl_snapshot(slot)
        except Exception as e:  # noqa: BLE001
            if not exchange:
                raise
            set_failed(e)
            bad_slots.add(slot)

    def ctl_wait(slot):
        try:
            if slot in bad_slots:
                raise RuntimeError(f"no snapshot in slot {slot}")
            return engine.ctl_wait(slot)
        except Exception as e:  # noqa: BLE001
            if not exchange:
                raise
            set_failed(e)
            if ctl is not None:
                return ctl
            blank = np.zeros(CTL_WORDS, np.int64)  # "running", nothing known: the loop goes on until the failure rule ends it
            blank[CTL_MIN_LEN], blank[CTL_NODES_GLOBAL] = INF, nodes_seen
            return blank

    def produce(c0, c1, idx):
        """expansion + exchange of one chunk on the side stream -> (n_par, event after which its receive area is complete, dead, stamps)"""
        nonlocal tight_used, fail_hdr_chunk
        n_par = c1 - c0
        tight_used = tight_used or (exchange and 0 < fill < FILL_HARD)
        with _on_side():
            # the expansion runs at most ONE chunk ahead of the dedup: the children a rank owns itself claim their table slots from
            # the expansion kernel, and the engine tells the claims of two chunks in flight apart by the chunk's parity
            if on_gpu and side is not main and done:
                side.wait_event(done[-1])  # chunk idx - 2 is committed (the last chunk consumed so far; chunk idx - 1 is the one being dedup'ed)
            e0 = stamp(side)
            dead = failure is not None and exchange
            if not dead:
                try:
                    send, recv = engine.chunk_expand(c0, c1, c0 == 0, fill)
                except Exception as e:  # noqa: BLE001
                    if not exchange:
                        raise
                    set_failed(e)
                    dead = True
            if dead:
                S, _, rw = engine.layout(n_par, fill)
                send = torch.zeros(S * world * rw, dtype=torch.int64, device=dev)
                hdr = send.view(S * world, rw)
                hdr[:, 1], hdr[:, 2], hdr[:, 3] = INF, INF, 4
                recv = torch.empty_like(send)
                if fail_hdr_chunk is None:
                    fail_hdr_chunk = idx
            e1 = stamp(side)
            if exchange:
                comm.all_to_all_single(recv, send)
            ev = None
            if on_gpu:
                ev = torch.cuda.Event(enable_timing=tl_rows is not None)
                ev.record(side)
        return n_par, ev, dead, (e0, e1)

    # Chunk sizes.  A chunk is expanded and deduplicated as a whole even when the budget runs out at its first parents, so near
    # the end of the budget the chunks shrink to what the remaining budget is expected to need: new states per parent so far in
    # this level (at its start: of the previous level) x the parents already in flight, from the lagged control block -- the
    # same integers on every rank, so every rank cuts the same chunks.  An estimate that falls short only costs another small
    # chunk.  (1e8-node search: the last 2^21-parent chunk was needed for a fraction of its parents.)
    min_chunk = min(B, 1 << 16)
    F_prev = 0  # parents of the previous level
    nodes_seen = 1

    def next_size(c0, sizes, n_read, new_read):
        n = min(B, F - c0)
        if n_read > 0:
            num, den = new_read, sum(sizes[:n_read])  # new states per parent, this level
        elif F_prev > 0:
            num, den = F, F_prev
        else:
            return n
        if num <= 0:
            return n
        in_flight = sum(sizes[n_read:])
        remaining = max_nodes - nodes_seen - (in_flight * num + den - 1) // den
        want = max(min_chunk, -(-(max(remaining, 0) * den * 9) // (num * 8)))
        want = (want + 2047) // 2048 * 2048
        return min(n, want)

    min_len = INF
    phase_closed = not replicating  # the replicated phase ends with ONE all-reduce (failure code), whichever way it ends
    try:
        while F > 0:
            levels += 1
            pending = []  # snapshot slots of the chunks whose control block has not been read yet
            k = 0
            ctl = None
            if on_gpu:
                side.wait_stream(main)  # the level's parents are the nodes the main stream committed during the previous level
            sizes, n_read, new_read, c_next = [], 0, 0, 0
            done = []  # per consumed chunk of this level: event behind its commit on the main stream

            def produce_next():
                nonlocal c_next
                n = next_size(c_next, sizes, n_read, new_read)
                sizes.append(n)
                c_next += n
                return produce(c_next - n, c_next, len(sizes) - 1)

            ready = produce_next()
            while ready is not None and (ctl is None or ctl[CTL_STATUS] == ST_RUNNING) and (fail_hdr_chunk is None or k <= fail_hdr_chunk + lag):
                n_par, ev, dead, (e0, e1) = ready
                ready = None
                if c_next < F and overlap != "commit":
                    ready = produce_next()  # runs beside this chunk's dedup and commit
                if ev is not None:
                    main.wait_event(ev)
                m0 = stamp(main)
                if dead:  # never went through the engine: nothing to dedup, nothing to commit; the collectives still pair up
                    gmask = torch.zeros((n_par + 1) // 2, dtype=torch.int32, device=dev)
                else:
                    try:
                        gmask = engine.chunk_insert(n_par)
                    except Exception as e:  # noqa: BLE001
                        set_failed(e)
                        try:  # masks only: the chunk stays in the engine's ring, so its commit (and the decide kernel in it) still runs
                            gmask = engine.chunk_insert_dead(n_par)
                        except Exception:  # noqa: BLE001
                            gmask = engine.gmask_view(n_par)  # the engine's own buffer: its commit reads the all-reduced masks from there
                            gmask.zero_()
                m1 = stamp(main)
                if c_next < F and overlap == "commit":
                    if on_gpu:
                        side.wait_stream(main)  # not before this chunk's dedup is through
                    ready = produce_next()  # runs beside this chunk's mask all-reduce and commit
                if exchange:
                    reduce_masks(gmask)  # every (parent, action) child has exactly one owner, so SUM == OR
                m2 = stamp(main)
                if not dead:
                    try:
                        engine.chunk_commit(max_nodes)
                    except Exception as e:  # noqa: BLE001
                        set_failed(e)
                m3 = stamp(main)
                if on_gpu and side is not main:
                    ev_done = torch.cuda.Event()
                    ev_done.record(main)
                    done.append(ev_done)
                if tl_rows is not None:
                    tl_rows.append((n_par, e0, e1, ev, m0, m1, m2, m3, levels))
                ctl_snapshot(k % (lag + 2))
                pending.append(k % (lag + 2))
                chunks += 1
                if len(pending) > lag:
                    ctl = ctl_wait(pending.pop(0))
                    n_read += 1
                    new_read, nodes_seen = int(ctl[CTL_NEXT_COUNT]), int(ctl[CTL_NODES_GLOBAL])
                k += 1
            if on_gpu:
                main.wait_stream(side)  # (a chunk that was produced but never consumed: the search ended)
            while pending and (ctl is None or ctl[CTL_STATUS] == ST_RUNNING):  # end of the level: the one synchronisation
                ctl = ctl_wait(pending.pop(0))
            status = int(ctl[CTL_STATUS])
            # closing all-reduce of the level (max): [failure code, -(smallest total length generated), fullest region received].
            # Every rank does it here, whatever status it read, so a failure that no header carried (the last chunk of a level; a
            # region overflow noticed after the search ended; an exception on one rank's host side) ends every rank at the same point.
            code = int(ctl[CTL_FAIL_LOCAL])
            if status == ST_FAILED:
                code = max(code, int(ctl[CTL_FAIL_SEEN]))
            if failure is not None:
                code = max(code, 4)
            closing = [code, -int(ctl[CTL_MIN_LEN]), int(ctl[CTL_LEVEL_FILL])]
            # the replicated phase: no closing all-reduce per level (every rank computes the same level) -- ONE when the phase ends, i.e. the
            # search has ended or the next level is large enough to be partitioned: a rank whose engine raised joins it from the handler below
            phase_end = replicating and (status != ST_RUNNING or int(ctl[CTL_NEXT_COUNT]) >= replicate_below or int(ctl[CTL_NEXT_COUNT]) == 0)
            if replicating:
                repl_levels += 1
            if phase_end:
                phase_closed = True
            if exchange or phase_end:
                e = i64(closing)
                comm.all_reduce(e, "max")
                closing = [int(v) for v in e.tolist()]
            min_len = min(min_len, -closing[1])
            if closing[0]:
                raise_failed(closing[0])
            if status == ST_MOVE_ERROR:
                # the reference executes this move before it stops: its ACMove raises (every rank sees the same words)
                raise AssertionError("a move emptied a relator during the search: the reference's ACMove raises here")
            if status == ST_SOLVED:
                tag = int(ctl[CTL_SOLVED_TAG])
                mine = engine.find(tag // 12)
                if replicating:  # every rank holds the whole tree: no collective
                    return finish(True, walk((rank << 40) | mine, [(tag % 12, 2)], collective=False), min_len)
                pref = i64([(rank << 40) | mine if mine >= 0 else -1])
                comm.all_reduce(pref, "max")
                return finish(True, walk(int(pref[0]), [(tag % 12, 2)]), min_len)
            if status == ST_BUDGET:
                return finish(False, None, min_len)
            F_prev, F, nodes_seen = F, int(ctl[CTL_NEXT_COUNT]), int(ctl[CTL_NODES_GLOBAL])
            if replicating:
                if phase_end and F > 0:
                    # the first large level: every rank keeps its own share of it (copies, in frontier order) and the chunks are exchanged
                    # from here on.  A rank whose partition raises goes on as a failed rank (dead chunks: the others learn it from the headers).
                    replicating, exchange = False, True
                    adaptive = region_fill is None
                    try:
                        engine.partition()
                    except Exception as e:  # noqa: BLE001
                        set_failed(e)
                        engine.replicated = False  # (the dead chunks this rank now sends are laid out for the real world size)
            if adaptive:  # the fullest region any rank received in this level (0: no chunk large enough to tell) sizes the next level's regions
                lf = closing[2]
                # (round 3, owner = hash of the whole key, measured level by level on five searches: the fullest region falls by up to 25 % from
                # one level to the next and rises by at most 7 %.  Round 5's owner function sends a quarter of the children: on AK(3) the
                # fullest region of a level is 0.58 / 0.44 / 0.36 / 0.34 of the even share of ALL children at 8 ranks, falling as the
                # conjugators grow.  An overflow only costs a rerun.)
                fill = FILL_DEFAULT if lf == 0 else min(FILL_DEFAULT, max(24, lf * 5 // 4 + 12))
                fills.append(fill)
    except _RegionOverflow:
        raise
    except Exception as e:  # noqa: BLE001
        if phase_closed or world == 1:  # (the reference's AssertionError for a move that empties a relator is raised behind the closing all-reduce)
            raise
        # an engine call of the replicated phase raised on THIS rank: the healthy ranks meet at the phase's closing all-reduce -- join it
        # with a failure code, so that every rank ends there
        phase_closed = True
        comm.all_reduce(i64([4, 0, 0]), "max")
        raise RuntimeError(f"sharded bfs failed on rank {rank}: {e}") from e
    return finish(False, None, min_len)


# ------------------------------------------------------------------ the whole search as ONE C call (round 6) ---
class NativeComm:
    """An `acx_comm` (include/acx.h) for acx_bfs_sharded: the two collectives of the sharded search as C function pointers.

    * `NativeComm.from_process_group(group=None)`: RCCL itself -- the ncclComm_t of a torch.distributed process group (backend "nccl"
      == RCCL on ROCm; `ProcessGroupNCCL._comm_ptr()`), driven from C through the librccl the process already runs.  No Python between
      the chunks of a search.
    * `NativeComm.create_rccl(rank, world, broadcast)`: a communicator of the library's own (ncclCommInitRank): rank 0 makes the 128-byte
      id, `broadcast(bytes_or_None) -> bytes` hands it to every rank (any transport: a torch.distributed store, MPI, a file).
    * `NativeComm.from_python(comm, device)`: any communicator object of this module (ThreadComm of the tests, TorchDistComm over gloo)
      behind ctypes callbacks -- the GPU tests run thread ranks of one GPU through it."""

    def __init__(self, struct, keep=(), owned=None):
        self.c, self._keep, self._owned = struct, keep, owned
        self.rank, self.world = int(struct.rank), int(struct.world)

    @classmethod
    def from_comm_ptr(cls, ptr):
        from ac_solver import _acx

        c = _acx.Comm()
        _acx.check(_acx.lib.acx_comm_rccl(C.c_void_p(int(ptr)), C.byref(c)), "acx_comm_rccl")
        return cls(c)

    @classmethod
    def from_process_group(cls, group=None, device=None):
        torch = _torch()
        import torch.distributed as dist

        pg = group if group is not None else dist.distributed_c10d._get_default_group()
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        backend = pg._get_backend(dev)
        if not hasattr(backend, "_comm_ptr"):
            raise RuntimeError("the process group's backend for CUDA tensors is not NCCL / RCCL: no communicator to hand to libacx")
        # the communicator exists once the group has run a collective on this device
        dist.all_reduce(torch.zeros(1, device=dev), group=group)
        torch.cuda.synchronize(dev)
        return cls.from_comm_ptr(backend._comm_ptr())

    @classmethod
    def create_rccl(cls, rank, world, broadcast):
        from ac_solver import _acx

        _acx.require_device()
        uid = (C.c_char * 128)()
        if rank == 0:
            _acx.check(_acx.lib.acx_rccl_unique_id(uid), "acx_rccl_unique_id")
        data = broadcast(bytes(uid.raw) if rank == 0 else None)
        uid.raw = bytes(data)
        handle = C.c_void_p()
        _acx.check(_acx.lib.acx_rccl_comm_create(uid, int(rank), int(world), C.byref(handle)), "acx_rccl_comm_create")
        out = cls.from_comm_ptr(handle.value)
        out._owned = handle
        return out

    def close(self):
        if self._owned is not None:
            from ac_solver import _acx

            _acx.lib.acx_rccl_comm_destroy(self._owned)
            self._owned = None

    @classmethod
    def from_python(cls, comm, device=None):
        from ac_solver import _acx

        torch = _torch()
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)

        class _Span:  # a device pointer as a torch tensor (the __cuda_array_interface__ protocol)
            def __init__(self, ptr, n, typestr):
                self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr, "data": (int(ptr), False), "version": 2}

        def tensor(ptr, n, typestr):
            return torch.as_tensor(_Span(ptr, n, typestr), device=dev)

        def on(stream):
            return torch.cuda.stream(torch.cuda.ExternalStream(int(stream), device=dev) if stream else torch.cuda.default_stream(dev))

        errors = []

        def a2a(ctx, send, recv, words, stream):
            try:
                with on(stream):
                    comm.all_to_all_single(tensor(recv, words, "<i8"), tensor(send, words, "<i8"))
                return 0
            except BaseException as e:  # noqa: BLE001
                errors.append(e)
                return -1

        def red(ctx, buf, n, dtype, op, stream):
            try:
                with on(stream):
                    comm.all_reduce(tensor(buf, n, "<i4" if dtype == _acx.I32 else "<i8"), "sum" if op == _acx.RED_SUM else "max")
                return 0
            except BaseException as e:  # noqa: BLE001
                errors.append(e)
                return -1

        fa, fr = _acx.ALL_TO_ALL_FN(a2a), _acx.ALL_REDUCE_FN(red)
        out = cls(_acx.Comm(int(comm.rank), int(comm.world), None, fa, fr), keep=(fa, fr, comm))
        out.errors = errors
        return out


def bfs_sharded_native(presentation, max_nodes_to_explore=10000, verbose=False, cyclically_reduce_after_moves=False, comm=None, batch_parents=None,
                       want_stats=False, overlap=None, region_fill=None, replicate_below=None, mask_comm=None, log_fraction=None, _fail_at_call=0, _fail_rank=0):
    """`bfs` over the ranks of `comm` (a NativeComm; None = one rank) as ONE C call per rank: acx_bfs_sharded (csrc/acx_shard_run.hip) runs
    what bfs_sharded above orchestrates from Python -- same engine, same chunk loop, same result.  Same contract as `bfs`
    (breadth_first.py:15-97): (is_search_successful, path or None) [+ stats], identical on every rank."""
    from ac_solver import _acx
    from ac_solver.envs.utils import is_array_valid_presentation
    from ac_solver.search._common import _check_width

    assert is_array_valid_presentation(presentation), f"{presentation} is not a valid presentation"
    p = _acx.as_i8_rows(np.array(presentation))
    L = p.size // 2
    _check_width(L, p)
    _acx.require_device()
    torch = _torch()
    opts = _acx.ShardOpts()
    opts.batch_parents = int(batch_parents or 0)
    opts.replicate_below = 0 if replicate_below is None else (-1 if int(replicate_below) <= 1 else int(replicate_below))
    opts.region_fill = int(region_fill or 0)
    opts.overlap = {None: 0, True: 0, "insert": 2, False: 1}[overlap]
    if mask_comm is not None:
        opts.mask_comm = C.pointer(mask_comm.c)
    opts.fail_at_call, opts.fail_rank = int(_fail_at_call), int(_fail_rank)
    opts.log_fraction_q8 = 0 if log_fraction is None else max(1, int(log_fraction * 256))
    cap = 1 << 12
    while True:
        pa, pl = np.empty(cap, np.int32), np.empty(cap, np.int32)
        solved, n, st = C.c_int32(), C.c_int64(), _acx.ShardRunStats()
        rc = _acx.lib.acx_bfs_sharded(_acx.ptr(p, C.c_int8), L, int(max_nodes_to_explore), int(bool(cyclically_reduce_after_moves)),
                                      None if comm is None else C.byref(comm.c), C.byref(opts), C.byref(solved), _acx.ptr(pa, C.c_int32), _acx.ptr(pl, C.c_int32),
                                      cap, C.byref(n), C.byref(st), torch.cuda.current_stream().cuda_stream)
        if rc == _acx.E_CAPACITY and n.value > cap:
            cap = int(n.value)
            continue
        if rc == _acx.E_ROWERR:
            raise AssertionError(_acx.last_error())
        if rc != _acx.OK:
            raise RuntimeError(_acx.last_error() or f"acx_bfs_sharded failed (code {rc})")
        break
    path = list(zip(pa[: n.value].tolist(), pl[: n.value].tolist())) if solved.value else None
    if want_stats:
        return bool(solved.value), path, {k: getattr(st, k) for k, _ in _acx.ShardRunStats._fields_} | {"world": 1 if comm is None else comm.world}
    return bool(solved.value), path
