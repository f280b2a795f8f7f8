"""Test infrastructure for the sharded BFS orchestrator (ac_solver/search/sharded.py):

* OracleShardEngine -- the per-rank engine interface implemented with NumPy + the CPU oracle, so that the
  orchestration (routing, global numbering, budget / success decisions, path walk) can be exercised on CPU
  with torch.distributed's gloo backend.  Never used by the product.
* ThreadComm -- an in-process communicator: several ranks run as threads of one process (one GPU can then
  play all ranks of the HIP engine in the GPU tests).
"""
import threading

import numpy as np
import torch

from oracle import ac_oracle as O

_CODE = {-2: 0, -1: 1, 1: 2, 2: 3}
_LETTER = {0: -2, 1: -1, 2: 1, 3: 2}


def pack_word(word):
    k = 0
    for i, a in enumerate(word):
        k |= _CODE[int(a)] << (2 * i)
    return k | (len(word) << 58)


def to_i64(u):
    return u - (1 << 64) if u >= (1 << 63) else u


def key_of_state(state, L):
    r0 = [a for a in state[:L] if a != 0]
    r1 = [a for a in state[L:] if a != 0]
    return to_i64(pack_word(r0)), to_i64(pack_word(r1))


def state_of_key(k0, k1, L):
    row = np.zeros(2 * L, np.int8)
    for h, k in enumerate((k0, k1)):
        u = k & ((1 << 64) - 1)
        n = u >> 58
        for i in range(n):
            row[h * L + i] = _LETTER[(u >> (2 * i)) & 3]
    return row


class OracleShardEngine:
    """The engine interface of ac_solver/search/sharded.py (seed / level_begin / expand_routed / insert / commit / find /
    node_info / status) on the CPU: NumPy + the C oracle's ACMove.  Test infrastructure only."""

    KW = 2

    def __init__(self, L, cyclical, node_cap, batch_cap, chunk_parents, rank, world):
        assert L <= 29
        self.L, self.cyc, self.rank, self.world = L, bool(cyclical), rank, world
        self.batch_cap, self.node_cap = batch_cap, node_cap
        self.device = torch.device("cpu")
        self.states, self.prefs, self.acts, self.tlens, self.gpos = [], [], [], [], []
        self.visited = {}
        self.pending = []
        self.lvl_lo = self.lvl_hi = 0
        self.err = 0
        self.min_len = 1 << 30

    def root_record(self, p):
        k0, k1 = key_of_state(np.asarray(p), self.L)
        return np.array([k0, k1, 0, -1], np.int64)

    def _add(self, k0, k1, act, pref, gpos):
        st = state_of_key(k0, k1, self.L)
        self.visited[(k0, k1)] = len(self.states)
        self.states.append(st)
        self.prefs.append(pref)
        self.acts.append(act)
        self.tlens.append(int(np.count_nonzero(st)))
        self.gpos.append(gpos)

    def seed(self, record):
        if record is not None:
            self._add(int(record[0]), int(record[1]), 0, -1, 0)

    def level_begin(self):
        self.lvl_lo, self.lvl_hi = self.lvl_hi, len(self.states)
        return self.lvl_hi - self.lvl_lo

    def expand_routed(self, c0, c1, n_local, solved, world):
        from ac_solver.search.sharded import owner_of

        rows = []
        for nid in range(self.lvl_lo, self.lvl_hi):
            gp = self.gpos[nid]
            if not c0 <= gp < c1:
                continue
            st = np.repeat(self.states[nid][None], 12, axis=0)
            out, lens, err = O.move_batch(st, np.arange(12, dtype=np.uint8), self.L, cyclical=self.cyc)
            for a in range(12):
                tag = 12 * gp + a
                if err[a]:
                    solved[1] = min(int(solved[1]), (tag << 8) | int(err[a]))
                tl = int(lens[a].sum())
                self.min_len = min(self.min_len, tl)
                if tl == 2:
                    solved[0] = min(int(solved[0]), tag)
                if np.array_equal(out[a], self.states[nid]):
                    continue  # an unchanged child is its (visited) parent: never sent
                k0, k1 = key_of_state(out[a], self.L)
                rows.append([k0, k1, tag, (self.rank << 40) | nid])
        recs = torch.tensor(rows, dtype=torch.int64).reshape(-1, 4)
        owners = owner_of(recs[:, :2], world) if len(rows) else torch.zeros(0, dtype=torch.int64)
        return [recs[owners == o].contiguous() for o in range(world)]

    def insert(self, recv, c0, n_parents):
        recs = sorted(recv.tolist(), key=lambda r: r[2])
        seen, self.pending = set(), []
        mask = torch.zeros(n_parents, dtype=torch.int32)
        for k0, k1, tag, pref in recs:
            if (k0, k1) in self.visited or (k0, k1) in seen:
                continue
            seen.add((k0, k1))
            self.pending.append((k0, k1, tag, pref))
            mask[tag // 12 - c0] |= 1 << (tag % 12)
        self.c0 = c0
        return mask

    def commit(self, cutoff, lmask, lprefix, gmask, gprefix, gpos_base, n_commit):
        todo = [r for r in self.pending if r[2] < cutoff]
        assert len(todo) == n_commit, (len(todo), n_commit)
        if len(self.states) + n_commit > self.node_cap + 64:
            raise RuntimeError("engine capacity exceeded")
        for k0, k1, tag, pref in todo:
            par, a = tag // 12 - self.c0, tag % 12
            gp = gpos_base + int(gprefix[par]) + bin(int(gmask[par]) & ((1 << a) - 1)).count("1")
            lid = len(self.states)
            assert lid == self._base(lprefix, lmask, par, a, todo)
            self._add(k0, k1, a, pref, gp)
        self.pending = []

    def _base(self, lprefix, lmask, par, a, todo):
        first = len(self.states) - sum(1 for r in todo if (r[0], r[1]) in self.visited)
        return first + int(lprefix[par]) + bin(int(lmask[par]) & ((1 << a) - 1)).count("1")

    def find(self, gpos):
        for nid in range(self.lvl_lo, self.lvl_hi):
            if self.gpos[nid] == gpos:
                return nid
        return -1

    def node_info(self, nid):
        return (-1 if self.prefs[nid] < 0 else self.acts[nid]), self.tlens[nid], self.prefs[nid]

    def status(self):
        return self.err, self.min_len


class ThreadComm:
    """Ranks are threads of one process; collectives meet at a barrier."""

    class Shared:
        def __init__(self, world):
            self.world = world
            self.barrier = threading.Barrier(world)
            self.slots = [None] * world

    def __init__(self, shared, rank):
        self.s, self.rank, self.world = shared, rank, shared.world

    def _exchange(self, value):
        self.s.slots[self.rank] = value
        self.s.barrier.wait()
        vals = list(self.s.slots)
        self.s.barrier.wait()
        return vals

    def all_to_all_regions(self, regions):
        everyone = self._exchange(list(regions))
        return torch.cat([everyone[src][self.rank] for src in range(self.world)])

    def all_gather_var(self, t):
        return self._exchange(t)

    def all_reduce(self, t, op):
        vals = self._exchange(t.clone())
        st = torch.stack(vals)
        red = {"min": st.min(0)[0], "max": st.max(0)[0], "sum": st.sum(0)}[op]
        t.copy_(red)
        return t


def run_threads(world, fn):
    """fn(comm) on `world` threads -> list of results (exceptions re-raised)"""
    shared = ThreadComm.Shared(world)
    out, errs = [None] * world, []

    def work(r):
        try:
            out[r] = fn(ThreadComm(shared, r))
        except BaseException as e:  # noqa: BLE001
            errs.append(e)
            shared.barrier.abort()

    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join() for t in th]
    if errs:
        raise errs[0]
    return out
