// acx_greedy_mega.h -- big buckets of a single greedy_search on the whole GPU (ac_solver/search/greedy.py:15-121).
//
// The persistent frontier of acx_greedy.h runs one search on ONE compute unit: right for the thousands of small batches a
// best-first search consists of (a round trip to the host per batch would cost more than the batch), wrong for the few
// hundred buckets that hold thousands of parents of one (total length, depth): on AK(3) with a 1e7-node budget those are
// 85 % of all expanded parents and 60 % of the kernel's time.  When the persistent kernel selects a bucket with at
// least GreedyDev::hand_min queued parents it parks its state (GreedyState) and returns GREEDY_HANDOFF; the host then
// runs that bucket here, in MEGA-BATCHES of up to kMegaParents parents spread over the whole chip, and relaunches the
// persistent kernel afterwards.  A mega-batch is the batch of acx_greedy.h step for step -- same tags, same decisions:
//
//   k_gm_rank      order the bucket by the signed state tuple: an all-pairs counting sort over the whole chip
//   k_gm_begin     scalars of the batch, empty in-batch table
//   k_gm_expand    one lane per tag t = 12 * parent + action: move, success / raising-move test, read-only probe of the
//                  visited table, in-batch dedup to the minimum tag (the key of a table occupant is REBUILT from its
//                  parent, never read from another workgroup's stores)
//   k_gm_mark      winners (minimum tag of an unseen key), winners per tile, first new child shorter than the bucket
//   k_gm_decide    tile prefix, the child that reaches the budget, the reference's cut / budget / success decision
//   k_gm_commit    winners below the cutoff become nodes (numbered in tag order), enter the visited table, count per length
//   k_gm_file      bucket records of (length, depth + 1): growth, bitmap, counts; the popped parents leave their bucket
//   k_gm_push      node ids into their buckets
#pragma once
#include "acx_greedy.h"

namespace acx {

constexpr uint32_t kMegaParents = 16384;
constexpr uint32_t kMegaTags = 12u * kMegaParents;
constexpr uint32_t kMegaTile = 1024;  // tags per workgroup of k_gm_mark / k_gm_commit
constexpr uint32_t kMegaTiles = kMegaTags / kMegaTile;
constexpr uint32_t kMegaSlots = 1u << 19;  // in-batch table, >= 2 * kMegaTags
constexpr uint32_t kNone = 0xFFFFFFFFu;

struct MegaScalars {
    unsigned long long err_tag;             // min (tag << 8 | error) of a move the reference raises on
    uint32_t solved, shorter, lo, seen;     // minimum tags (kNone: none) / smallest total length seen
    uint32_t committed, total;
    uint32_t lcnt[132], pushbase[132];
    // the decision (k_gm_decide) and the outcome (k_gm_file)
    uint32_t cutoff, p_end, is_solved, budget_hit, err, last_parent, last_child_len, solved_pid, solved_action;
    uint32_t status, cut, remaining, nodes0, base, cur_len, cur_depth;
    // chained mode (round 4; the host enqueues frontier kernel -> sort -> mega-batch -> frontier kernel ... without reading anything
    // in between): h_* is written by the frontier kernel when it hands a bucket off (pending, its queued parents, 1 = unsorted tail)
    // and kept up to date by k_gm_file; b_* is this mega-batch as k_gm_begin froze it (0 / np / 12 np)
    uint32_t h_pending, h_live, h_sort, b_active, b_np, b_m;
    uint32_t tbase[kMegaTiles + 1], tcnt[kMegaTiles + 1];
};

template <typename W> struct MegaDev {
    GreedyDev<W> g;
    W* ck0;          // [kMegaTags] candidate keys
    W* ck1;
    uint8_t* clen;   // [kMegaTags] total length of the candidate
    uint32_t* info;  // [kMegaTags] bit 0 active, bit 1 seen before (visited table, or the move changed nothing), bit 2 winner; in-batch slot << 4
    uint32_t* idv;   // [kMegaTags] node id of a committed winner (kNone otherwise)
    uint32_t* posv;  // [kMegaTags] its position among this batch's new nodes of the same total length
    uint32_t* mtab;  // [kMegaSlots] minimum tag per key
    uint32_t* rank;  // [g.rank_max] k_gm_rank: entries of the bucket that precede entry i (zero between sorts)
    MegaScalars* sc;
};

// ---- ordering a bucket ----------------------------------------------------------------------------------------------------
// A handed-off bucket of at most kMegaRankMax entries is ordered by COUNTING (round 4; a larger one the frontier kernel orders
// itself before it hands it off): the keys of a bucket are pairwise distinct, so the
// final position of an entry is the number of entries that precede it.  k_gm_rank spreads the n x n comparisons over the whole chip --
// a work item is 256 entries against a tile of 32, an LDS broadcast per comparison -- and k_gm_begin, the first kernel of the
// mega-batch behind it, writes the ids to their places: ~15 us for the average handed-off bucket (3 200 entries) where the sorted runs + rank merge took 41 + 15 us with three or
// four workgroups busy.  The work grows with n^2: beyond kMegaRankMax entries the runs are cheaper again.
// (a work item's tile: 128 entries 176.5-177.4 ms per 1e7-node search, 64: 173.4-174.1, 32: 171.1-172.5, 16: 171.4-172.2, 8: 173.1-174.2, 256: 183.2-184.0 --
// the bucket is a few thousand entries, so the chip runs about one wave per SIMD and a work item's length is instruction LATENCY)
#ifndef ACX_RANK_TILE
#define ACX_RANK_TILE 32
#endif
constexpr uint32_t kRankTile = ACX_RANK_TILE;
#ifndef ACX_RANK_GRID
#define ACX_RANK_GRID 1024
#endif
constexpr uint32_t kRankGrid = ACX_RANK_GRID;  // workgroups of k_gm_rank (a work item = 256 entries x a tile; a bucket of 3 200 entries: 1 300 items)
template <typename W> __global__ void __launch_bounds__(256) k_gm_rank(MegaDev<W> md, uint32_t n, uint32_t chained) {
    __shared__ W sj0[kRankTile];
    __shared__ W sj1[kRankTile];
    ACX_VGPR_PAD_W(W, "v47", "v63");
    const GreedyDev<W>& g = md.g;
    if (chained) {
        if (!md.sc->h_pending || !md.sc->h_sort) return;
        n = md.sc->h_live;
    }
    if (n > g.rank_max) return;
    const GreedyState* ps = g.state;
    const BucketRec r = g.bk[(size_t)ps->cur_len * kDepthCap + ps->cur_depth];
    const uint32_t base = r.off + r.head, tid = threadIdx.x;
    const uint32_t eb = (n + 255u) / 256u, jt_n = (n + kRankTile - 1u) / kRankTile;
    for (uint32_t w = blockIdx.x; w < eb * jt_n; w += gridDim.x) {
        const uint32_t ei = w / jt_n, jt = w - ei * jt_n;
        const uint32_t i = ei * 256u + tid, j = jt * kRankTile + tid;
        uint32_t id = 0;
        W m0 = 0, m1 = 0;
        if (i < n) {
            id = g.arena[base + i];
            const NodeKey<W> nk = g.nkeys[id];
            m0 = nk.k0;
            m1 = nk.k1;
        }
        if (tid < kRankTile && j < n) {
            const NodeKey<W> nk = g.nkeys[g.arena[base + j]];
            sj0[tid] = nk.k0;
            sj1[tid] = nk.k1;
        }
        __syncthreads();
        if (i < n) {
            const uint32_t jn = min(kRankTile, n - jt * kRankTile);
            uint32_t c = 0;
            for (uint32_t q = 0; q < jn; q++) c += key_less<W>(sj0[q], sj1[q], m0, m1) ? 1u : 0u;
            if (c) atomicAdd(&md.rank[i], c);
            if (jt == 0) g.gid[i] = id;
        }
        __syncthreads();
    }
}

// ---- one mega-batch -----------------------------------------------------------------------------------------------------------
// `sort_n`: the bucket has just been counted by k_gm_rank (sort_n entries; 0: no): the ids go to their places first
template <typename W> __global__ void __launch_bounds__(256) k_gm_begin(MegaDev<W> md, uint32_t slots, uint32_t sort_n, uint32_t chained) {
    ACX_VGPR_PAD("v23");
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (chained) {  // the batch is what the frontier kernel handed off / what the last batch left of it; nothing pending: every kernel of the batch returns
        const uint32_t pending = md.sc->h_pending, live = md.sc->h_live, np = min(live, kMegaParents);
        sort_n = pending && md.sc->h_sort ? live : 0u;  // (k_gm_file clears h_sort: only the first batch of a hand-off finds it set)
        if (i == 0) {
            md.sc->b_active = pending && np ? 1u : 0u;
            md.sc->b_np = np;
            md.sc->b_m = 12u * np;
        }
        if (!pending || !np) return;
        slots = 1024;
        while (slots < 24u * np) slots <<= 1;
    }
    if (sort_n) {
        const GreedyDev<W>& g = md.g;
        GreedyState* ps = g.state;
        BucketRec* rp = g.bk + (size_t)ps->cur_len * kDepthCap + ps->cur_depth;
        const BucketRec r = *rp;
        for (uint32_t k = i; k < sort_n; k += gridDim.x * blockDim.x) {
            const uint32_t rk = md.rank[k];
            md.rank[k] = 0;  // (zero again for the next bucket)
            g.arena[r.off + r.head + rk] = g.gid[k];
        }
        if (i == 0) {
            rp->sorted_end = r.cnt;
            ps->sorts++;
            if (sort_n > greedy_cfg<W>::kSortCap) ps->big_sorts++;
            ps->hist[min(15, 31 - __builtin_clz(sort_n))]++;
            if (sort_n > ps->max_bucket) ps->max_bucket = sort_n;
        }
    }
    for (uint32_t k = i; k < slots; k += gridDim.x * blockDim.x) md.mtab[k] = kNone;
    MegaScalars* sc = md.sc;
    if (i < 132) {
        sc->lcnt[i] = 0;
        sc->pushbase[i] = 0;
    }
    if (i == 0) {
        const GreedyState* ps = md.g.state;
        const BucketRec r = md.g.bk[(size_t)ps->cur_len * kDepthCap + ps->cur_depth];
        sc->err_tag = ~0ull;
        sc->solved = sc->shorter = sc->lo = sc->seen = kNone;
        sc->committed = sc->total = 0;
        sc->nodes0 = ps->nodes;
        sc->base = r.off + r.head;
        sc->cur_len = ps->cur_len;
        sc->cur_depth = ps->cur_depth;
        sc->status = GREEDY_RUNNING;
        sc->cut = 0;
        sc->remaining = 0;
    }
}

template <typename W> __device__ __forceinline__ void gm_child(const MegaDev<W>& md, uint32_t base, uint32_t t, W& c0, W& c1, uint32_t& tl, int& e, W& p0, W& p1) {
    const GreedyDev<W>& g = md.g;
    const uint32_t p = t / 12u;
    const NodeKey<W> nk = g.nkeys[g.arena[base + p]];
    p0 = nk.k0;
    p1 = nk.k1;
    Pres<W> s;
    key_to_pres<W>(p0, p1, s);
    e = g.nf ? apply_move_nf<W, kSearchSafeOf<W>>(s, (int)(t - 12u * p), g.d.L, g.d.cyclical != 0) : apply_move<W, kSearchSafeOf<W>>(s, (int)(t - 12u * p), g.d.L, g.d.cyclical != 0);
    c0 = keyops<W>::make(s.w0, s.n0);
    c1 = keyops<W>::make(s.w1, s.n1);
    tl = (uint32_t)(s.n0 + s.n1);
}

template <typename W> __global__ void __launch_bounds__(256) k_gm_expand(MegaDev<W> md, uint32_t m, uint32_t smask, uint32_t chained) {
    ACX_VGPR_PAD_W(W, "v71", "v103");
    const GreedyDev<W>& g = md.g;
    MegaScalars* sc = md.sc;
    if (chained) {
        if (!sc->b_active) return;
        m = sc->b_m;
        uint32_t slots = 1024;
        while (slots < 2u * m) slots <<= 1;
        smask = slots - 1;
    }
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= m) return;
    const uint32_t base = sc->base;
    W c0, c1, p0, p1;
    uint32_t tl;
    int e;
    gm_child<W>(md, base, t, c0, c1, tl, e, p0, p1);
    if (e) atomicMin(&sc->err_tag, ((unsigned long long)t << 8) | (unsigned long long)e);  // counts only if the reference gets this far
    if (tl == 2) atomicMin(&sc->solved, t);  // greedy.py:91, before the membership test
    md.ck0[t] = c0;
    md.ck1[t] = c1;
    md.clen[t] = (uint8_t)tl;
    const uint64_t h = greedy_hash(c0, c1);
    bool known = c0 == p0 && c1 == p1;  // an over-long product leaves the parent, which is in the visited set
    if (!known) {  // read-only probe of the visited table: pairs of 8-byte slots (node id | 32-bit fingerprint << 32)
        const uint32_t fp = (uint32_t)(h >> 32);
        uint32_t hv = (uint32_t)h & g.tmask & ~1u;
        for (;;) {
            const ulonglong2 sl = *(const ulonglong2*)(g.tab + hv);
            const unsigned long long s2[2] = {sl.x, sl.y};
            bool miss = false;
#pragma unroll
            for (int j = 0; j < 2; j++) {
                if (known || miss) continue;
                if (s2[j] == kTabEmpty) {
                    miss = true;
                } else if ((uint32_t)(s2[j] >> 32) == fp) {
                    const NodeKey<W> nk = g.nkeys[(uint32_t)s2[j]];
                    known = nk.k0 == c0 && nk.k1 == c1;
                }
            }
            if (known || miss) break;
            hv = (hv + 2) & g.tmask;
        }
    }
    uint32_t hs = 0;
    if (!known) {  // in-batch dedup: the minimum tag among equal keys keeps the slot
        hs = (uint32_t)(h >> 40) & smask;
        for (;;) {
            uint32_t v = md.mtab[hs];
            if (v == kNone) {
                v = atomicCAS(&md.mtab[hs], kNone, t);
                if (v == kNone) break;
            }
            W q0, q1, r0, r1;
            uint32_t ql;
            int qe;
            gm_child<W>(md, base, v, q0, q1, ql, qe, r0, r1);  // any holder of this slot has the same key as the first
            if (q0 == c0 && q1 == c1) {
                if (v > t) atomicMin(&md.mtab[hs], t);
                break;
            }
            hs = (hs + 1) & smask;
        }
    }
    md.info[t] = 1u | (known ? 2u : 0u) | (hs << 4);
}

template <typename W> __global__ void __launch_bounds__(kMegaTile) k_gm_mark(MegaDev<W> md, uint32_t m, uint32_t chained) {
    ACX_VGPR_PAD("v23");
    __shared__ uint32_t s_cnt;
    MegaScalars* sc = md.sc;
    if (chained) {
        if (!sc->b_active) return;
        m = sc->b_m;
        if (blockIdx.x * kMegaTile >= m) return;  // (a full-size grid over a short batch)
    }
    const uint32_t t = blockIdx.x * kMegaTile + threadIdx.x, lane = threadIdx.x & 63u;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    bool win = false;
    uint32_t inf = 0;
    if (t < m) {
        inf = md.info[t];
        win = (inf & 3u) == 1u && md.mtab[inf >> 4] == t;
        if (win) md.info[t] = inf | 4u;
    }
    {  // tags grow with the lane: the lowest flagged lane holds the wave's minimum
        const unsigned long long sb = __ballot(win && (uint32_t)md.clen[t < m ? t : 0] < sc->cur_len);
        if (sb && lane == (uint32_t)__builtin_ctzll(sb)) atomicMin(&sc->shorter, t);
    }
    const unsigned long long bal = __ballot(win);
    if (lane == 0 && bal) atomicAdd(&s_cnt, (uint32_t)__popcll(bal));
    __syncthreads();
    if (threadIdx.x == 0) sc->tcnt[blockIdx.x] = s_cnt;
}

// one workgroup: prefix of the tile counts, the winner that reaches the budget, the decision of greedy.py:71-119 for this batch
template <typename W> __global__ void __launch_bounds__(256) k_gm_decide(MegaDev<W> md, uint32_t np, uint32_t m, uint32_t chained) {
    ACX_VGPR_PAD("v31");
    __shared__ uint32_t s_x[256];
    __shared__ uint32_t s_lo, s_tile, s_tile_base;
    const GreedyDev<W>& g = md.g;
    MegaScalars* sc = md.sc;
    if (chained) {
        if (!sc->b_active) return;
        np = sc->b_np;
        m = sc->b_m;
    }
    const uint32_t tid = threadIdx.x;
    const uint32_t tiles = (m + kMegaTile - 1) / kMegaTile;
    const uint32_t mine = tid < tiles ? sc->tcnt[tid] : 0u;
    s_x[tid] = mine;
    if (tid == 0) s_lo = kNone;
    __syncthreads();
    for (uint32_t o = 1; o < 256; o <<= 1) {  // inclusive scan (kMegaTiles <= 256)
        const uint32_t y = tid >= o ? s_x[tid - o] : 0u;
        __syncthreads();
        s_x[tid] += y;
        __syncthreads();
    }
    const uint32_t excl = s_x[tid] - mine, total = s_x[255];
    if (tid < tiles) sc->tbase[tid] = excl;
    const long long nodes = (long long)sc->nodes0;
    const bool over = nodes < g.max_nodes && nodes + (long long)total >= g.max_nodes;
    __syncthreads();
    if (over) {  // the winner of rank max_nodes - nodes - 1 (in tag order) is the child that reaches the budget
        const uint32_t want = (uint32_t)(g.max_nodes - nodes - 1);
        if (tid < tiles && excl <= want && want < excl + mine) {
            s_tile = tid;
            s_tile_base = excl;
        }
        __syncthreads();
        const uint32_t tile = s_tile, t0 = tile * kMegaTile + tid * 4u;
        uint32_t w[4], cnt = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            w[k] = (t0 + k < m && (md.info[t0 + k] & 4u)) ? 1u : 0u;
            cnt += w[k];
        }
        __syncthreads();
        s_x[tid] = cnt;
        __syncthreads();
        for (uint32_t o = 1; o < 256; o <<= 1) {
            const uint32_t y = tid >= o ? s_x[tid - o] : 0u;
            __syncthreads();
            s_x[tid] += y;
            __syncthreads();
        }
        uint32_t r = s_tile_base + s_x[tid] - cnt;  // rank of my first tag
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (w[k]) {
                if (r == want) s_lo = t0 + k;
                r++;
            }
        }
        __syncthreads();
    }
    if (tid == 0) {
        const uint32_t base = sc->base;
        uint32_t p_end = np - 1, budget_hit = 0;
        if (sc->shorter != kNone) p_end = min(p_end, sc->shorter / 12u);
        if (nodes >= g.max_nodes) {
            p_end = 0;
            budget_hit = 1;
        } else if (over) {
            const uint32_t pb = s_lo / 12u;
            if (pb <= p_end) {
                p_end = pb;
                budget_hit = 1;
            }
        }
        const uint32_t solved = sc->solved;
        uint32_t is_solved = solved != kNone && solved / 12u <= p_end;
        uint32_t err = 0;
        const unsigned long long et = sc->err_tag;
        // a move on which the reference's ACMove raises: it does so when the move is executed before the search ends
        if (et != ~0ull && (uint32_t)((et >> 8) / 12u) <= p_end && !(is_solved && solved < (uint32_t)(et >> 8))) {
            err = (uint32_t)(et & 0xffu);
            is_solved = 0;
        }
        sc->cutoff = is_solved ? solved : 12u * (p_end + 1);
        p_end = is_solved ? solved / 12u : p_end;
        sc->p_end = p_end;
        sc->budget_hit = budget_hit;
        sc->is_solved = is_solved;
        sc->err = err;
        sc->total = total;
        sc->lo = s_lo;
        sc->last_parent = g.arena[base + p_end];
        sc->last_child_len = md.clen[12u * p_end + 11u];  // greedy.py:121
        sc->solved_pid = is_solved ? g.arena[base + solved / 12u] : 0u;
        sc->solved_action = is_solved ? solved % 12u : 0u;
    }
}

template <typename W> __global__ void __launch_bounds__(kMegaTile) k_gm_commit(MegaDev<W> md, uint32_t m, uint32_t chained) {
    ACX_VGPR_PAD_W(W, "v31", "v39");
    __shared__ uint32_t s_w[kMegaTile / 64];
    __shared__ uint32_t s_l[132], s_lb[132];
    __shared__ uint32_t s_cm;
    const GreedyDev<W>& g = md.g;
    const SearchDev<W>& d = g.d;
    MegaScalars* sc = md.sc;
    if (chained) {
        if (!sc->b_active) return;
        m = sc->b_m;
        if (blockIdx.x * kMegaTile >= m) return;
    }
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    const uint32_t t = blockIdx.x * kMegaTile + tid;
    const uint32_t cutoff = sc->cutoff, p_end = sc->p_end, nodes = sc->nodes0, base = sc->base;
    const uint32_t inf = t < m ? md.info[t] : 0u;
    const bool win = (inf & 4u) != 0;
    const unsigned long long bal = __ballot(win);
    if (tid == 0) s_cm = 0;
    if (tid < 132) s_l[tid] = 0;
    if (lane == 0) s_w[wv] = (uint32_t)__popcll(bal);
    __syncthreads();
    uint32_t off = sc->tbase[blockIdx.x];
    for (uint32_t k = 0; k < wv; k++) off += s_w[k];
    const uint32_t cpos = off + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
    const bool cm = win && t < cutoff;
    uint32_t tl = 0, id = kNone, pos = 0;
    if (cm) {
        id = nodes + cpos;
        const uint32_t p = t / 12u;
        tl = md.clen[t];
        NodeKey<W> nk;
        nk.k0 = md.ck0[t];
        nk.k1 = md.ck1[t];
        g.nkeys[id] = nk;
        d.parent[id] = g.arena[base + p];
        d.act[id] = (uint8_t)(t - 12u * p);
        d.tlen[id] = (uint8_t)tl;
        d.depth[id] = sc->cur_depth + 1;
        // visited table: the keys of a batch's winners are pairwise distinct and absent; first free slot of the probe sequence
        const uint64_t h = greedy_hash(nk.k0, nk.k1);
        const unsigned long long mine = (unsigned long long)id | ((h >> 32) << 32);
        uint32_t hv = (uint32_t)h & g.tmask & ~1u;
        while (atomicCAS(&g.tab[hv], kTabEmpty, mine) != kTabEmpty) hv = (hv + 1) & g.tmask;
    }
    {  // smallest total length among the children the reference generates before the batch ends
        uint32_t seen = (t < m && (inf & 1u) && t < 12u * (p_end + 1)) ? (uint32_t)md.clen[t] : kNone;
        for (int o = 32; o > 0; o >>= 1) seen = min(seen, (uint32_t)__shfl_xor((int)seen, o));
        if (lane == 0 && seen != kNone) atomicMin(&sc->seen, seen);
    }
    // position inside the target bucket: per (wave, total length) in LDS, then one global atomic per (tile, total length)
    unsigned long long cb = __ballot(cm);
    if (lane == 0 && cb) atomicAdd(&s_cm, (uint32_t)__popcll(cb));
    while (cb) {
        const uint32_t lead = (uint32_t)__builtin_ctzll(cb);
        const uint32_t v = (uint32_t)__shfl((int)tl, (int)lead);
        const unsigned long long same = __ballot(cm && tl == v) & cb;
        uint32_t b = 0;
        if (lane == lead) b = atomicAdd(&s_l[v], (uint32_t)__popcll(same));
        b = (uint32_t)__shfl((int)b, (int)lead);
        if (cm && tl == v) pos = b + (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
        cb &= ~same;
    }
    __syncthreads();
    if (tid < 132 && s_l[tid]) s_lb[tid] = atomicAdd(&sc->lcnt[tid], s_l[tid]);
    if (tid == 0 && s_cm) atomicAdd(&sc->committed, s_cm);
    __syncthreads();
    if (t < m) {
        md.idv[t] = id;
        md.posv[t] = cm ? s_lb[tl] + pos : 0u;
    }
}

// one workgroup: what the persistent kernel does between "commit" and the next batch (filing, tail)
template <typename W> __global__ void __launch_bounds__(256) k_gm_file(MegaDev<W> md, uint32_t np, uint32_t chained) {
    ACX_VGPR_PAD("v31");
    __shared__ uint32_t s_job[132 * 3];
    __shared__ uint32_t s_lc[132], s_hint[132];
    __shared__ uint32_t s_njobs, s_top, s_status, s_reason;
    const GreedyDev<W>& g = md.g;
    MegaScalars* sc = md.sc;
    if (chained) {
        if (!sc->b_active) return;
        np = sc->b_np;
    }
    GreedyState* ps = g.state;
    const uint32_t tid = threadIdx.x, nlen = g.nlen;
    if (tid < 132) {
        s_lc[tid] = ps->len_count[tid];
        s_hint[tid] = ps->hint[tid];
    }
    const uint32_t cur_len = sc->cur_len, cur_depth = sc->cur_depth, D1 = cur_depth + 1;
    const bool is_solved = sc->is_solved != 0;
    if (tid == 0) {
        s_njobs = 0;
        s_top = ps->arena_top;
        s_status = GREEDY_RUNNING;
        s_reason = 0;
    }
    __syncthreads();
    if (tid < nlen && sc->lcnt[tid] > 0 && !is_solved) {
        const uint32_t k = sc->lcnt[tid];
        if (D1 >= kDepthCap) {
            s_status = GREEDY_FALLBACK;
            s_reason = 1;
        } else {
            BucketRec r = g.bk[(size_t)tid * kDepthCap + D1];
            const uint32_t lv = r.cnt - r.head;
            bool ok = true;
            if (r.cap == 0 || r.cnt + k > r.cap) {
                uint32_t nc = 16;
                while (nc < 2 * (lv + k)) nc <<= 1;
                const uint32_t no = atomicAdd(&s_top, nc);
                if ((unsigned long long)no + nc > g.arena_cap) {
                    s_status = GREEDY_FALLBACK;
                    s_reason = 2;
                    ok = false;
                } else if (lv > 0) {
                    const uint32_t j = atomicAdd(&s_njobs, 1u);
                    s_job[3 * j] = r.off + r.head;
                    s_job[3 * j + 1] = no;
                    s_job[3 * j + 2] = lv;
                }
                r.sorted_end = r.sorted_end > r.head ? r.sorted_end - r.head : 0;
                r.head = 0;
                r.off = no;
                r.cap = nc;
                r.cnt = lv;
            }
            sc->pushbase[tid] = r.off + r.cnt;
            r.cnt += k;
            if (ok) {
                g.bk[(size_t)tid * kDepthCap + D1] = r;
                if (lv == 0) atomicOr(&g.bitmap[(size_t)tid * (kDepthCap / 32) + D1 / 32], 1u << (D1 & 31));
                s_lc[tid] += k;
                if (D1 < s_hint[tid]) s_hint[tid] = D1;
            }
        }
    }
    __syncthreads();
    for (uint32_t j = 0; j < s_njobs; j++) {  // grown buckets move to their new region
        const uint32_t src = s_job[3 * j], dst = s_job[3 * j + 1], cnt = s_job[3 * j + 2];
        for (uint32_t i = tid; i < cnt; i += 256) g.arena[dst + i] = g.arena[src + i];
    }
    if (tid == 0) {
        ps->arena_top = s_top;
        ps->mega_batches++;
        ps->mega_parents += np;
        if (sc->seen < ps->seen_min) ps->seen_min = sc->seen;
        uint32_t status = s_status;
        if (status == GREEDY_FALLBACK) {
            ps->reason = s_reason;
        } else {
            const uint32_t popped = sc->p_end + 1;
            ps->batches++;
            ps->expanded += popped;
            ps->nodes = sc->nodes0 + sc->committed;
            if (sc->committed && !is_solved && D1 > ps->depth_hi) ps->depth_hi = D1;
            ps->last_parent = sc->last_parent;
            ps->last_child_len = sc->last_child_len;
            ps->hist[16 + 31 - __builtin_clz(np)]++;
            if (sc->err) {
                ps->err = sc->err;
                status = GREEDY_MOVE_ERROR;
            } else if (is_solved) {
                ps->solved_pid = sc->solved_pid;
                ps->solved_action = sc->solved_action;
                status = GREEDY_SOLVED;
            } else {
                BucketRec* rp = g.bk + (size_t)cur_len * kDepthCap + cur_depth;
                BucketRec r = *rp;
                r.head += popped;
                s_lc[cur_len] -= popped;
                const bool empty = r.head == r.cnt;
                const bool cut = sc->shorter != kNone;
                if (empty) {
                    r.head = r.cnt = r.sorted_end = 0;
                    atomicAnd(&g.bitmap[(size_t)cur_len * (kDepthCap / 32) + cur_depth / 32], ~(1u << (cur_depth & 31)));
                }
                *rp = r;
                if (sc->budget_hit) status = GREEDY_BUDGET;
                sc->cut = cut ? 1u : 0u;
                sc->remaining = r.cnt - r.head;
                ps->np_cap = cut ? (uint32_t)(kGT / 12) : (uint32_t)(kSingleSortCap<W> / 12u);
            }
        }
        ps->status = status;
        sc->status = status;
        if (chained) {  // another mega-batch of this bucket (what the frontier kernel reads as GREEDY_MEGA_MORE), or the bucket is done with
            const bool more = status == GREEDY_RUNNING && !sc->cut && sc->remaining != 0;
            sc->h_live = more ? sc->remaining : 0u;
            sc->h_sort = 0;
            if (!more) sc->h_pending = 0;
        }
    }
    __syncthreads();
    if (tid < 132) {
        ps->len_count[tid] = s_lc[tid];
        ps->hint[tid] = s_hint[tid];
    }
}

template <typename W> __global__ void __launch_bounds__(256) k_gm_push(MegaDev<W> md, uint32_t m, uint32_t chained) {
    ACX_VGPR_PAD("v23");
    const MegaScalars* sc = md.sc;
    if (chained) {
        if (!sc->b_active) return;
        m = sc->b_m;
    }
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= m || sc->is_solved || sc->status == GREEDY_FALLBACK) return;
    const uint32_t id = md.idv[t];
    if (id != kNone) md.g.arena[sc->pushbase[md.clen[t]] + md.posv[t]] = id;
}

}  // namespace acx
