"""Test infrastructure for the sharded BFS orchestrator (ac_solver/search/sharded.py):

* OracleShardEngine -- the per-rank engine interface implemented with NumPy + the CPU oracle, in the SAME buffer
  formats as the HIP engine (fixed-size regions with headers, the record log, the control block), so that the
  orchestration (equal-split all-to-all, mask all-reduce, lagged control block, path walk) can be exercised on CPU
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
INF = 1 << 62
HDR, SUB, TILE = 4, 16, 1536  # header words, sub-regions per destination, children per workgroup = 128 parents x 12 (csrc/acx_shard.hip)
_INVERSE = {0: 2, 2: 0, 1: 3, 3: 1, 4: 8, 8: 4, 5: 9, 9: 5, 6: 10, 10: 6, 7: 11, 11: 7}


FILL_DEFAULT = 320
FILL_HARD = 1 << 20
FILL_MIN_EVEN = 2048  # chunks with a smaller even share do not update the level's fill word (csrc/acx_shard.hip: k_shard_decide)


def even_share(n_par, world):
    return -(-12 * n_par // (world * world * SUB)) if world > 1 else 0


def layout(n_par, world, KW, fill_q8=0):
    """Python mirror of shard_layout (csrc/acx_shard.hip); tests/test_sharded_cpu.py checks it against acx_shard_layout"""
    n_blocks = -(-12 * n_par // TILE)
    hard = -(-n_blocks // SUB) * TILE
    cap = 0 if world == 1 else hard  # world 1: every child is born where it is owned (no record), a region is its header
    if world > 1 and fill_q8 < FILL_HARD:
        if fill_q8 <= 0 or fill_q8 > FILL_DEFAULT:
            fill_q8 = FILL_DEFAULT
        cap = min(hard, -(-even_share(n_par, world) * fill_q8 // 256) + 2 * TILE)
    return SUB, cap, HDR + cap * (KW + 1)


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


def is_normal_form(state, L, cyclical):
    for h in (0, 1):
        w = [int(a) for a in state[h * L:(h + 1) * L] if a != 0]
        if not w or any(w[i] == -w[i + 1] for i in range(len(w) - 1)):
            return False
        if cyclical and len(w) >= 2 and w[0] == -w[-1]:
            return False
    return True


class OracleShardEngine:
    """The engine interface of ac_solver/search/sharded.py on the CPU: NumPy + the C oracle's ACMove.  Test infrastructure only."""

    KW, RW = 2, 3

    def __init__(self, L, cyclical, node_cap, chunk_parents, rank, world, est_parents):
        assert L <= 29
        self.L, self.cyc, self.rank, self.world = L, bool(cyclical), rank, world
        self.node_cap, self.B = node_cap + 64, chunk_parents
        self.device = torch.device("cpu")
        self.states, self.prefs, self.acts, self.tlens, self.gpos = [], [], [], [], []
        self.visited = set()
        self.lvl_lo = self.lvl_hi = 0
        self.ctl = np.zeros(16, np.int64)
        self.ctl[9] = INF
        self.ctl[1] = 1  # len(tree_nodes) counts the root on every rank
        self.snap = {}
        self.nf = False
        self.inverse_dropped = 0
        self.open, self.inserted = [], 0
        self.fill_min_even = FILL_MIN_EVEN
        self.replicated = False
        self.own_from = 0

    @property
    def world_eff(self):
        return 1 if self.replicated else self.world

    def layout(self, n_par, fill_q8=0):
        return layout(n_par, self.world_eff, self.KW, fill_q8)

    def layout_words(self, n_par, fill_q8=0):
        s, _, rw = self.layout(n_par, fill_q8)
        return s * self.world_eff * rw

    def set_replicated(self):
        """csrc/acx_shard.hip acx_shard_set_replicated: whole levels on every rank, as world 1, until partition()"""
        assert not self.states
        self.replicated = True

    def partition(self):
        """acx_shard_partition: copies of the newest level's nodes this rank owns become its slice of the next level"""
        from ac_solver.search.sharded import owner_of

        assert self.replicated and not self.open
        self.replicated = False
        base, n_old = self.lvl_hi, len(self.states)
        self.own_from = n_old
        for nid in range(base, n_old):
            k0, k1 = key_of_state(self.states[nid], self.L)
            if int(owner_of(torch.tensor([[k0, k1]], dtype=torch.int64), self.world)[0]) == self.rank:
                self.states.append(self.states[nid])
                self.prefs.append(self.prefs[nid])
                self.acts.append(self.acts[nid])
                self.tlens.append(self.tlens[nid])
                self.gpos.append(self.gpos[nid])
        self.lvl_hi = n_old  # the next level switch makes [n_old, len(states)) the running level
        self.ctl[5] = len(self.states)

    def walk(self, nid, cap=256):
        pairs = []
        while True:
            pref = self.prefs[nid]
            pairs.append((-1 if pref < 0 else self.acts[nid], self.tlens[nid]))
            if pref < 0 or (pref >> 40) != self.rank or len(pairs) == cap:
                return pref, pairs
            nid = pref & ((1 << 40) - 1)

    def root_record(self, p):
        p = np.asarray(p)
        k0, k1 = key_of_state(p, self.L)
        self.nf = is_normal_form(p, self.L, self.cyc)  # the HIP engine drops "undo" children only in normal-form searches, cyclical = False
        return np.array([k0, k1, 0, -1], np.int64)

    def _add(self, k0, k1, act, pref, gpos):
        st = state_of_key(k0, k1, self.L)
        self.visited.add((k0, k1))
        self.states.append(st)
        self.prefs.append(pref)
        self.acts.append(act)
        self.tlens.append(int(np.count_nonzero(st)))
        self.gpos.append(gpos)

    def seed(self, record):
        if record is not None:
            self._add(int(record[0]), int(record[1]), 0xff, -1, 0)
            self.ctl[5] = 1

    def chunk_expand(self, c0, c1, level_first, fill_q8=0):
        from ac_solver.search.sharded import owner_of

        n_par = c1 - c0
        S, cap, rw = self.layout(n_par, fill_q8)
        we = self.world_eff
        send = torch.zeros(S * we * rw, dtype=torch.int64)
        recv = torch.zeros_like(send) if we > 1 else send
        born = []  # children of local parents that this rank owns itself: they never travel (the HIP engine: BORN stamps)
        self.open.append(((c0, n_par, S, cap, rw, we), recv, born))  # the orchestrator expands chunk k + 1 before it inserts chunk k
        if self.ctl[0] != 0:
            return send, recv
        if level_first:
            self.lvl_lo, self.lvl_hi = self.lvl_hi, len(self.states)
            self.ctl[2] = 0
            self.ctl[12] = 0
        regs = send.view(S * we, rw)
        regs[:, 1] = INF
        regs[:, 2] = INF
        regs[:, 3] = int(self.ctl[8])
        local = [nid for nid in range(self.lvl_lo, self.lvl_hi) if c0 <= self.gpos[nid] < c1]
        solved = err = INF
        for li, nid in enumerate(local):
            gp = self.gpos[nid]
            st = np.repeat(self.states[nid][None], 12, axis=0)
            out, lens, errs = O.move_batch(st, np.arange(12, dtype=np.uint8), self.L, cyclical=self.cyc)
            for a in range(12):
                tag = 12 * gp + a
                if errs[a]:
                    err = min(err, (tag << 8) | int(errs[a]))
                tl = int(lens[a].sum())
                self.ctl[9] = min(int(self.ctl[9]), tl)
                if tl == 2:
                    solved = min(solved, tag)
                if np.array_equal(out[a], self.states[nid]):
                    continue  # an unchanged child is its (visited) parent: never sent
                if self.nf and not self.cyc and self.acts[nid] < 12 and a == _INVERSE[self.acts[nid]]:
                    # the HIP engine does not send this child: it must be the parent's own tree parent, a visited state
                    assert key_of_state(out[a], self.L) in self.visited or self._is_parent_of(nid, out[a])
                    self.inverse_dropped += 1
                    continue
                k0, k1 = key_of_state(out[a], self.L)
                o = self.rank if we == 1 else int(owner_of(torch.tensor([[k0, k1]], dtype=torch.int64), self.world)[0])
                if o == self.rank:
                    born.append((tag - 12 * c0, k0, k1, (self.rank << 40) | nid))
                    continue
                sub = ((12 * li + a) // TILE) % S
                reg = regs[o * S + sub]
                n = int(reg[0])
                assert n < cap
                reg[HDR + n * 3: HDR + n * 3 + 3] = torch.tensor([k0, k1, ((tag - 12 * c0) << 32) | nid], dtype=torch.int64)
                reg[0] = n + 1
        regs[:, 1] = torch.minimum(regs[:, 1], torch.tensor(solved))
        regs[:, 2] = torch.minimum(regs[:, 2], torch.tensor(err))
        return send, recv

    def _is_parent_of(self, nid, state):
        """the tree parent of a local node lives on rank prefs[nid] >> 40; when it is local its state can be compared"""
        pref = self.prefs[nid]
        return pref >= 0 and (pref >> 40 != self.rank or np.array_equal(self.states[pref & ((1 << 40) - 1)], state))

    def gmask_view(self, n_par):
        """the chunk's mask buffer, zeroed (what the orchestrator all-reduces when chunk_insert itself failed): two parents per word"""
        self.packed = torch.zeros((n_par + 1) // 2, dtype=torch.int32)
        self.lmask = torch.zeros(n_par, dtype=torch.int32)
        self.winners = []
        return self.packed

    def _unpack(self, n_par):
        w = self.packed.to(torch.int64)
        both = torch.stack([w & 0xFFF, (w >> 16) & 0xFFF], dim=1).reshape(-1)
        return both[:n_par].to(torch.int32)

    def chunk_insert(self, n_par):
        self.geo, self.recv, born = self.open[self.inserted]
        self.inserted += 1
        c0, n_par, S, cap, rw, we = self.geo
        self.gmask_view(n_par)
        if self.ctl[0] != 0:
            return self.packed
        regs = self.recv.view(S * we, rw)
        recs = list(born)
        for r in range(S * we):
            n = int(regs[r, 0])
            if n > cap:
                self.ctl[8] = max(int(self.ctl[8]), 1)
                n = cap
            for i in range(n):
                k0, k1, x = (int(v) for v in regs[r, HDR + 3 * i: HDR + 3 * i + 3])
                recs.append((x >> 32, k0, k1, ((r // S) << 40) | (x & 0xFFFFFFFF)))
        seen = set()
        for tag, k0, k1, pref in sorted(recs):
            if (k0, k1) in self.visited or (k0, k1) in seen:
                continue
            seen.add((k0, k1))
            self.winners.append((tag, k0, k1, pref))
            self.lmask[tag // 12] |= 1 << (tag % 12)
        lm = torch.cat([self.lmask.to(torch.int64), torch.zeros(n_par % 2, dtype=torch.int64)]).view(-1, 2)
        self.packed.copy_((lm[:, 0] | (lm[:, 1] << 16)).to(torch.int32))
        return self.packed

    def chunk_insert_dead(self, n_par):
        """masks only (all zero), the chunk stays in the ring: csrc/acx_shard.hip acx_shard_chunk_insert_dead"""
        self.geo, self.recv, _ = self.open[self.inserted]
        self.inserted += 1
        return self.gmask_view(self.geo[1])

    def chunk_commit(self, max_nodes):
        if not self.open or self.inserted < 1:
            raise RuntimeError("chunk_commit: no inserted chunk is waiting")
        self.geo, self.recv, _ = self.open.pop(0)
        self.inserted = max(self.inserted - 1, 0)
        if self.ctl[0] != 0:
            return
        c0, n_par, S, cap, rw, we = self.geo
        regs = self.recv.view(S * we, rw)
        even = even_share(n_par, we)
        if even >= self.fill_min_even:  # the fullest region of the level, in 1/256 of the even share
            self.ctl[12] = max(int(self.ctl[12]), -(-int(regs[:, 0].max()) * 256 // even))
        fail = int(regs[:, 3].max())
        if fail:
            self.ctl[0], self.ctl[10] = 4, fail
            return
        solved, err = int(regs[:, 1].min()), int(regs[:, 2].min())
        self.gmask = self._unpack(n_par)  # (the orchestrator has all-reduced self.packed in place)
        pop = lambda v: bin(int(v)).count("1")  # noqa: E731
        gpop = np.array([pop(v) for v in self.gmask.tolist()], np.int64)
        lpop = np.array([pop(v) for v in self.lmask.tolist()], np.int64)
        gincl, lincl = np.cumsum(gpop), np.cumsum(lpop)
        nodes_global = int(self.ctl[1])
        need = max(max_nodes - nodes_global, 0)
        p_end, budget_hit, cg, cl = n_par - 1, False, int(gincl[-1]), int(lincl[-1])
        if need < 1:
            p_end, budget_hit, cg, cl = 0, True, int(gincl[0]), int(lincl[0])
        elif nodes_global + int(gincl[-1]) >= max_nodes:
            p_end = int(np.searchsorted(gincl, need))
            budget_hit, cg, cl = True, int(gincl[p_end]), int(lincl[p_end])
        end_pos = c0 + p_end
        is_solved = solved < INF and solved // 12 <= end_pos
        self.ctl[11] += 1
        if err < INF and (err >> 8) // 12 <= end_pos and not (is_solved and solved < (err >> 8)):
            self.ctl[0] = 3
            return
        if is_solved:
            q, a = solved // 12 - c0, solved % 12
            before = int(gincl[q] - gpop[q]) + pop(int(self.gmask[q]) & ((1 << a) - 1))
            self.ctl[3] += q + 1
            self.ctl[1] = nodes_global + before
            self.ctl[4] = solved
            self.ctl[0] = 1
            return
        if len(self.states) + cl > self.node_cap:
            self.ctl[8] = max(int(self.ctl[8]), 2)
        else:
            base = int(self.ctl[2])
            todo = [w for w in self.winners if w[0] < 12 * (p_end + 1)]
            assert len(todo) == cl, (len(todo), cl)
            for tag, k0, k1, pref in todo:
                par, a = tag // 12, tag % 12
                gp = base + int(gincl[par] - gpop[par]) + pop(int(self.gmask[par]) & ((1 << a) - 1))
                self._add(k0, k1, a, pref, gp)
            self.ctl[5] = len(self.states)
        self.ctl[2] += cg
        self.ctl[1] = nodes_global + cg
        self.ctl[3] += p_end + 1
        if budget_hit:
            self.ctl[0] = 2

    def ctl_snapshot(self, slot):
        self.snap[slot] = self.ctl.copy()

    def ctl_wait(self, slot):
        return self.snap[slot]

    def fail_local(self):
        self.ctl[8] = max(int(self.ctl[8]), 4)

    def find(self, gpos):
        for nid in range(self.lvl_lo, self.lvl_hi):
            if self.gpos[nid] == gpos:
                return nid
        return -1

    def node_info(self, nid):
        return (-1 if self.prefs[nid] < 0 else self.acts[nid]), self.tlens[nid], self.prefs[nid]


class ThreadComm:
    """Ranks are threads of one process; collectives meet at a barrier."""

    class Shared:
        def __init__(self, world):
            self.world = world
            self.barrier = threading.Barrier(world)
            self.slots = [None] * world

    def __init__(self, shared, rank):
        self.s, self.rank, self.world = shared, rank, shared.world
        self.stats = {"all_to_all_calls": 0, "all_reduce_calls": 0}

    def _exchange(self, value):
        self.s.slots[self.rank] = value
        self.s.barrier.wait()
        vals = list(self.s.slots)
        self.s.barrier.wait()
        return vals

    def all_to_all_single(self, recv, send):
        assert recv.numel() == send.numel() and send.numel() % self.world == 0
        if send.is_cuda:
            torch.cuda.current_stream(send.device).synchronize()  # the other threads read this buffer on THEIR streams
        everyone = self._exchange(send)
        k = send.numel() // self.world
        for src in range(self.world):
            recv[src * k:(src + 1) * k].copy_(everyone[src][self.rank * k:(self.rank + 1) * k])
        if send.is_cuda:
            torch.cuda.current_stream(send.device).synchronize()
        self._exchange(None)  # nobody rewrites its send buffer before everybody has copied out of it
        self.stats["all_to_all_calls"] += 1

    def all_reduce(self, t, op):
        if t.is_cuda:
            torch.cuda.current_stream(t.device).synchronize()
        vals = self._exchange(t.clone())
        st = torch.stack([v.to(t.device) for v in vals])
        red = {"min": st.min(0)[0], "max": st.max(0)[0], "sum": st.sum(0)}[op]
        t.copy_(red)
        self.stats["all_reduce_calls"] += 1
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
