// acx_frontier.h -- what the search translation units share: packed keys, the visited tables, the batch kernels of
// the BFS / greedy frontiers (expand, insert with minimum-tag resolution, mark, commit, decide) and the device
// block pool.  Included by acx_search.hip (single-GPU searches) and acx_shard.hip (per-GPU engine of the sharded BFS).
#pragma once
#include <array>
#include <string.h>
#include <cstring>
#include <stdlib.h>
#include <algorithm>
#include <map>
#include <memory>
#include <mutex>
#include <vector>
#include "acx_common.h"
#include "acx_word.h"
#include "acx_keys.h"  // keyops: (word, length) <-> key; u128x: the key type of max_relator_length 62 .. 64

namespace acx {

constexpr uint32_t kEmpty = 0xFFFFFFFFu;
constexpr uint32_t kProv = 0x80000000u;  // provisional id = kProv | tag (candidate of the running batch)
constexpr uint64_t kNoTag = ~0ull;


ACX_HD uint64_t mix64(uint64_t x) {
    x ^= x >> 32;
    x *= 0xd6e8feb86659fd93ull;
    x ^= x >> 32;
    x *= 0xd6e8feb86659fd93ull;
    x ^= x >> 32;
    return x;
}
ACX_HD uint64_t fold(uint64_t k) { return k; }
ACX_HD uint64_t fold(u128 k) { return (uint64_t)k ^ ((uint64_t)(k >> 64) * 0x9e3779b97f4a7c15ull); }
template <typename W> ACX_HD uint64_t hash_key(W k0, W k1) { return mix64(fold(k0) * 0x9e3779b97f4a7c15ull + mix64(fold(k1))); }


template <typename W> struct SearchDev {
    unsigned long long* stab;  // fused single-GPU BFS: stamp table (acx_bfs.h); stmask = slots - 1
    uint32_t stmask;
    uint32_t epoch;            // of this search's stamps (1 .. 254): a slot with another epoch field is free
    // node arena (committed nodes, id order == the reference's insertion order)
    W* k0;
    W* k1;
    uint32_t* parent;
    uint8_t* act;
    uint8_t* tlen;
    uint32_t* depth;
    // visited table: node id / provisional id / kEmpty
    uint32_t* slots;
    uint32_t smask;
    // batch-local table (greedy) for the in-batch dedup
    uint32_t* bslots;
    uint32_t bmask;
    // candidates of the running batch, indexed by tag
    W* ck0;
    W* ck1;
    uint8_t* clen;
    uint32_t* cslot;
    uint32_t* cflag;  // 1 = winner / new
    uint32_t* cpos;   // exclusive scan of cflag
    uint8_t* cknown;  // greedy: already in the visited table
    uint8_t* btook;   // fused BFS (acx_bfs.h): byte-wide "took the slot" / "was replaced" flags per tag of the running batch
    uint8_t* brepl;   // (null on the greedy batch-per-launch path, which numbers its winners through cflag / cpos)
    // device scalars
    unsigned long long* solved_tag;   // min tag with total length 2
    unsigned long long* shorter_tag;  // greedy: min tag of a NEW child shorter than the bucket
    unsigned long long* err_tag;      // min (tag << 8 | code) of a move on which the reference's ACMove raises
    uint32_t* err;
    uint32_t* min_len;
    // verbose searches (acx_search_minima_enable): first_len[l] = smallest tag of the running batch whose child has total
    // length l, recorded for l < min_len_start (the minimum when the batch began); null when off
    unsigned long long* first_len;
    uint32_t min_len_start;
    int32_t L;
    int32_t cyclical;
};

// Shift flavour of the moves in the search kernels.  Search keys cap max_relator_length at 29 (u64) / 61 (u128), so every shift
// count stays below the word width and the plain hardware shifts (SAFE = false in acx_word.h) are enough.  Rounds 1 and 2 ran
// the range-checked shifts here: round 1's k_expand built from the plain ones (exactly 24 VGPRs) returned corrupted children,
// and the checked flavour happened to allocate differently.  Round 3 found the actual cause -- a 64-bit shift whose amount sits
// in the LAST allocated VGPR -- which tools/check_shift64.py now excludes for every kernel at build time,
// so the search kernels are back on the plain shifts: 3 % off the 1e8-node bfs (fused and sharded) and off greedy_search.
#ifndef ACX_SEARCH_SAFE
#define ACX_SEARCH_SAFE 0
#endif
// (u128x, max_relator_length 62 .. 64: a shift by 64 letters IS the word width -- those searches run the range-checked flavour)
template <typename W> constexpr bool kSearchSafeOf = ACX_SEARCH_SAFE != 0 || is_long_key<W>::value;

template <typename W> __device__ __forceinline__ void key_to_pres(W k0, W k1, Pres<W>& s) {
    keyops<W>::split(k0, s.w0, s.n0);
    keyops<W>::split(k1, s.w1, s.n1);
}

// one lane per (parent, action): tag t = 12 * p + a
template <typename W>
__global__ void __launch_bounds__(256) k_expand(SearchDev<W> d, const uint32_t* __restrict__ plist, uint32_t pbegin, uint32_t np) {
    ACX_VGPR_PAD_W(W, "v39", "v55");
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t tl = 0xFFFFFFFFu;
    if (t < 12u * np) {
        const uint32_t p = t / 12u, a = t - 12u * p;
        const uint32_t pid = plist ? plist[p] : pbegin + p;
        Pres<W> s;
        const W pk0 = d.k0[pid], pk1 = d.k1[pid];
        key_to_pres<W>(pk0, pk1, s);
        const int e = apply_move<W, kSearchSafeOf<W>>(s, (int)a, d.L, d.cyclical != 0);
        // the reference's ACMove raises here -- but only if it gets this far: the search raises when this move precedes its
        // termination (k_decide), so the FIRST such move of the batch is what matters
        if (e) atomicMin(d.err_tag, ((unsigned long long)t << 8) | (unsigned long long)e);
        const W c0 = keyops<W>::make(s.w0, s.n0), c1 = keyops<W>::make(s.w1, s.n1);
        d.ck0[t] = c0;
        d.ck1[t] = c1;
        d.cknown[t] = (c0 == pk0 && c1 == pk1) ? 1 : 0;  // an unchanged state is its (visited) parent: k_insert skips the probe
        d.cslot[t] = 0;
        tl = (uint32_t)(s.n0 + s.n1);
        d.clen[t] = (uint8_t)tl;
        if (d.first_len && tl < d.min_len_start) atomicMin(&d.first_len[tl], (unsigned long long)t);  // a candidate for "New minimal length found"
        if (tl == 2) atomicMin(d.solved_tag, (unsigned long long)t);  // breadth_first.py:84 / greedy.py:91
    }
    // wave-level min before the atomic keeps contention low (all 64 lanes take part)
    uint32_t m = tl;
    for (int o = 32; o > 0; o >>= 1) m = min(m, (uint32_t)__shfl_xor((int)m, o));
    // one contended address: only waves that would actually lower the minimum issue the atomic (a plain, possibly
    // stale read can only over-estimate the current minimum, so no update is lost)
    if ((threadIdx.x & 63) == 0 && m < *(volatile uint32_t*)d.min_len) atomicMin(d.min_len, m);
}

// A probe sequence longer than its table means the table is full or damaged: the kernels stop probing and report it through
// the sticky error word instead of spinning forever.
constexpr uint32_t kErrTableFull = 0x40000000u;

template <typename W> __device__ __forceinline__ bool key_equals(const SearchDev<W>& d, uint32_t id, W k0, W k1) {
    if (id & kProv) {
        const uint32_t t = id & ~kProv;
        return d.ck0[t] == k0 && d.ck1[t] == k1;
    }
    return d.k0[id] == k0 && d.k1[id] == k1;
}

// Insert candidate t into `slots` with min-tag resolution among equal keys.  cslot[t] = slot holding the key.
// Occupants may be committed node ids (always win) or provisional ids of this batch.
template <typename W>
__global__ void __launch_bounds__(256) k_insert(SearchDev<W> d, uint32_t* __restrict__ slots, uint32_t mask, uint32_t m, int skip_known) {
    ACX_VGPR_PAD_W(W, "v31", "v39");
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= m) return;
    if (skip_known && d.cknown[t]) {
        d.cslot[t] = kEmpty;
        return;
    }
    const W k0 = d.ck0[t], k1 = d.ck1[t];
    const uint32_t me = kProv | t;
    uint32_t h = (uint32_t)hash_key<W>(k0, k1) & mask, probes = 0;
    for (;;) {
        uint32_t v = slots[h];
        if (v == kEmpty) {
            v = atomicCAS(&slots[h], kEmpty, me);
            if (v == kEmpty) break;  // claimed
        }
        if (key_equals<W>(d, v, k0, k1)) {
            if ((v & kProv) && v > me) atomicMin(&slots[h], me);
            break;
        }
        h = (h + 1) & mask;
        if (++probes > mask) {
            atomicOr(d.err, kErrTableFull);
            h = kEmpty;
            break;
        }
    }
    d.cslot[t] = h;
}

// read-only membership test against the visited table (greedy: speculative batches must not touch it)
template <typename W> __global__ void __launch_bounds__(256) k_lookup(SearchDev<W> d, uint32_t m) {
    ACX_VGPR_PAD("v31");
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= m) return;
    const W k0 = d.ck0[t], k1 = d.ck1[t];
    uint32_t h = (uint32_t)hash_key<W>(k0, k1) & d.smask, probes = 0;
    uint8_t known = 0;
    for (;;) {
        const uint32_t v = d.slots[h];
        if (v == kEmpty) break;
        if (key_equals<W>(d, v, k0, k1)) {
            known = 1;
            break;
        }
        h = (h + 1) & d.smask;
        if (++probes > d.smask) {
            atomicOr(d.err, kErrTableFull);
            break;
        }
    }
    d.cknown[t] = known;
}

// cflag[t] = 1 iff candidate t is the first discoverer of a state not seen before
template <typename W>
__global__ void __launch_bounds__(256) k_mark(SearchDev<W> d, const uint32_t* __restrict__ slots, uint32_t m, int bucket_len) {
    ACX_VGPR_PAD("v15");
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= m) return;
    const uint32_t s = d.cslot[t];
    const uint32_t win = (s != kEmpty && slots[s] == (kProv | t)) ? 1u : 0u;
    d.cflag[t] = win;
    if (win && bucket_len >= 0 && (int)d.clen[t] < bucket_len) atomicMin(d.shorter_tag, (unsigned long long)t);
}

// What the reference does with this batch, decided on the device so that the host needs ONE read-back per batch.
struct Decision {
    uint32_t p_end;       // last parent of the batch that the reference pops
    uint32_t cutoff;      // candidates with tag < cutoff are committed
    uint32_t committed;   // number of winners below cutoff
    uint32_t total;       // winners in the whole batch
    uint32_t budget_hit;  // len(tree_nodes) >= max_nodes after parent p_end
    uint32_t solved;      // a child of total length 2 was generated at or before parent p_end
    uint32_t solved_tag;
    uint32_t last_child_len;  // total length of child (p_end, action 11): greedy.py:121
    uint32_t err;
    uint32_t min_len;
};

template <typename W>
__global__ void k_decide(SearchDev<W> d, uint32_t m, uint32_t np, unsigned long long nodes, long long max_nodes, int greedy, Decision* __restrict__ out) {
    ACX_VGPR_PAD("v23");
    const uint32_t total = d.cpos[m - 1] + d.cflag[m - 1];
    uint32_t p_end = np - 1, budget_hit = 0;
    const unsigned long long shorter = *d.shorter_tag, solved_tag = *d.solved_tag;
    if (greedy && shorter != kNoTag) p_end = min(p_end, (uint32_t)(shorter / 12));  // the shorter new child is the heap's next minimum
    if ((long long)nodes >= max_nodes) {  // only possible for the very first parent (budget <= 1)
        p_end = 0;
        budget_hit = 1;
    } else if ((long long)(nodes + total) >= max_nodes) {
        // first candidate whose inclusive winner count reaches `need` (cpos + cflag is non-decreasing in t)
        const uint32_t need = (uint32_t)(max_nodes - (long long)nodes);
        uint32_t lo = 0, hi = m - 1;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (d.cpos[mid] + d.cflag[mid] >= need) hi = mid;
            else lo = mid + 1;
        }
        const uint32_t pb = lo / 12;
        if (pb <= p_end) {
            p_end = pb;
            budget_hit = 1;
        }
    }
    const uint32_t is_solved = solved_tag != kNoTag && (uint32_t)(solved_tag / 12) <= p_end;
    const uint32_t cutoff = is_solved ? (uint32_t)solved_tag : 12u * (p_end + 1);  // on success only stats need the commit
    out->p_end = is_solved ? (uint32_t)(solved_tag / 12) : p_end;
    out->cutoff = cutoff;
    out->committed = cutoff >= m ? total : d.cpos[cutoff];
    out->total = total;
    out->budget_hit = budget_hit;
    out->solved = is_solved;
    out->solved_tag = (uint32_t)solved_tag;
    out->last_child_len = d.clen[12u * p_end + 11];
    // an erroring move counts when the reference executes it: its parent is popped (<= p_end) and no earlier child ended the search
    const unsigned long long et = *d.err_tag;
    const bool err_hit = et != kNoTag && (uint32_t)((et >> 8) / 12) <= p_end && !(is_solved && solved_tag < (et >> 8));
    out->err = err_hit ? (uint32_t)(et & 0xff) : 0u;
    if (err_hit) out->solved = 0;
    if (*d.err & kErrTableFull) out->err = 0xFE;  // not a move error: a probe sequence ran through its whole table
    out->min_len = *d.min_len;
}

// Winners below `cutoff` become nodes base + cpos[t].  BFS: their table slot (already claimed in the visited
// table) is rewritten to the final id.  Greedy: the key is inserted into the visited table now (it is known
// to be absent and the committed keys are pairwise distinct, so a plain CAS claim is enough).
template <typename W>
__global__ void __launch_bounds__(256) k_commit(SearchDev<W> d, const uint32_t* __restrict__ plist, uint32_t pbegin, const Decision* __restrict__ dec, uint32_t m,
                                                uint32_t base, int insert_now) {
    ACX_VGPR_PAD("v31");
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= m || t >= dec->cutoff || !d.cflag[t]) return;
    const uint32_t id = base + d.cpos[t];
    const uint32_t p = t / 12u;
    const uint32_t pid = plist ? plist[p] : pbegin + p;
    const W k0 = d.ck0[t], k1 = d.ck1[t];
    d.k0[id] = k0;
    d.k1[id] = k1;
    d.parent[id] = pid;
    d.act[id] = (uint8_t)(t - 12u * p);
    d.tlen[id] = d.clen[t];
    d.depth[id] = d.depth[pid] + 1;
    if (insert_now == 1) {  // batch-per-launch greedy: id table
        uint32_t h = (uint32_t)hash_key<W>(k0, k1) & d.smask, probes = 0;
        while (atomicCAS(&d.slots[h], kEmpty, id) != kEmpty) {
            h = (h + 1) & d.smask;
            if (++probes > d.smask) {
                atomicOr(d.err, kErrTableFull);
                break;
            }
        }
    }
}

// ---- BFS: winners -> nodes (acx_bfs.h: k_bfs_count + k_bfs_compact) work on tiles of kCompactTile consecutive candidates, 32 per lane
// (one winner bit each in a 32-bit word) ---------------------------------------------------------------------------------------------
#ifndef ACX_COMPACT_ITEMS
#define ACX_COMPACT_ITEMS 32
#endif
constexpr uint32_t kCompactItems = ACX_COMPACT_ITEMS;  // a multiple of 8, at most 32 (one flag bit each in a 32-bit word)
constexpr uint32_t kCompactTile = 256 * kCompactItems;

// k_decide for the one-pass BFS commit: the winners of the batch are nodes base .. base + total - 1 in tag order, so
// "winners before tag T" is a binary search over their (parent, action) and the budget-crossing candidate is winner need - 1.
// Device-resident batch cursor of the fused BFS (round 3).  Once the frontier holds a full batch the host stops reading every
// batch's decision back (a 25 us round trip per batch, 33 per 1e8-node search): it enqueues batch after batch with full-size
// grids, the kernels take "which parents, how many nodes" from this block, k_decide_tab advances it, and the host looks at a
// pinned snapshot two batches late.  A batch that ENDS the search (success, budget, a raising move, a full table) is not applied:
// its decision is parked in `term` and the host finishes it exactly as it finishes a batch it read back synchronously.
struct BfsCursor {
    uint32_t head, nodes;     // next FIFO position to expand / len(tree_nodes)
    uint32_t status;          // 0 running, 1 the batch described by term* ended the search (not applied), 3 queue exhausted (applied)
    uint32_t batches;
    unsigned long long expanded;
    uint32_t min_len, term_pbegin, term_np, pad_;
    Decision term;
};

__device__ __forceinline__ bool cursor_begin(const BfsCursor* cur, uint32_t& m, uint32_t& np, uint32_t& pbegin, uint32_t& base) {
    if (cur->status) return false;
    pbegin = cur->head;
    const uint32_t avail = cur->nodes - pbegin;
    np = avail < np ? avail : np;
    m = 12u * np;
    base = cur->nodes;
    return true;
}
__device__ __forceinline__ void cursor_advance(BfsCursor* cur, const Decision& dec, uint32_t pbegin, uint32_t np, uint32_t base) {
    cur->batches += 1;
    if (dec.min_len < cur->min_len) cur->min_len = dec.min_len;
    // (the status word is stored LAST, behind a fence: the host's snapshot copies may run while this kernel does, and whatever they
    // see with a non-zero status must be complete)
    if (dec.solved || dec.budget_hit || dec.err) {  // this batch ends the search: the host finishes it
        cur->term = dec;
        cur->term_pbegin = pbegin;
        cur->term_np = np;
        __threadfence();
        cur->status = 1;
    } else {
        const uint32_t head = pbegin + dec.p_end + 1, nodes = base + dec.committed;
        cur->head = head;
        cur->nodes = nodes;
        cur->expanded += (unsigned long long)dec.p_end + 1;
        if (head >= nodes) {  // queue exhausted (breadth_first.py:61)
            __threadfence();
            cur->status = 3;
        }
    }
}

template <typename W>
__device__ __forceinline__ void decide_tab_body(const SearchDev<W>& d, uint32_t m, uint32_t np, uint32_t pbegin, uint32_t base, uint32_t cap_nodes, long long max_nodes,
                                                const uint32_t* __restrict__ total_in, Decision* __restrict__ out, int reset_tags, BfsCursor* cur, BfsCursor* __restrict__ snap) {
    // run-ahead mode (acx_bfs.h): the batch is what the cursor says.  `snap`: this batch's own snapshot slot -- the cursor as this
    // kernel leaves it, written by this kernel alone and not touched again before the host has read it (the host copies the SLOT,
    // never the live cursor, which the next batch's kernels may be changing while the copy runs: no torn snapshot)
    if (cur && !cursor_begin(cur, m, np, pbegin, base)) {
        if (snap) *snap = *cur;  // the search has ended in an earlier batch: the final cursor, once more
        return;
    }
    const uint32_t total = *total_in;
    const unsigned long long nodes = base;
    uint32_t p_end = np - 1, budget_hit = 0;
    const unsigned long long solved_tag = *d.solved_tag;
    auto tag_of = [&](uint32_t j) { return 12u * (d.parent[base + j] - pbegin) + (uint32_t)d.act[base + j]; };
    if ((long long)nodes >= max_nodes) {
        p_end = 0;
        budget_hit = 1;
    } else if ((long long)(nodes + total) >= max_nodes) {
        const uint32_t need = (uint32_t)(max_nodes - (long long)nodes);
        const uint32_t pb = tag_of(need - 1) / 12u;
        if (pb <= p_end) {
            p_end = pb;
            budget_hit = 1;
        }
    }
    const uint32_t is_solved = solved_tag != kNoTag && (uint32_t)(solved_tag / 12) <= p_end;
    const uint32_t cutoff = is_solved ? (uint32_t)solved_tag : 12u * (p_end + 1);
    uint32_t committed = total;
    if (cutoff < m) {  // winners with a tag below the cutoff (only a terminating batch gets here)
        const unsigned long long room = (unsigned long long)cap_nodes - nodes;
        uint32_t lo = 0, hi = (unsigned long long)total < room ? total : (uint32_t)room;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (tag_of(mid) < cutoff) lo = mid + 1;
            else hi = mid;
        }
        committed = lo;
    }
    out->p_end = is_solved ? (uint32_t)(solved_tag / 12) : p_end;
    out->cutoff = cutoff;
    out->committed = committed;
    out->total = total;
    out->budget_hit = budget_hit;
    out->solved = is_solved;
    out->solved_tag = (uint32_t)solved_tag;
    out->last_child_len = d.clen ? d.clen[12u * p_end + 11] : 0u;  // only greedy_search returns it (greedy.py:121)
    const unsigned long long et = *d.err_tag;
    const bool err_hit = et != kNoTag && (uint32_t)((et >> 8) / 12) <= p_end && !(is_solved && solved_tag < (et >> 8));
    out->err = err_hit ? (uint32_t)(et & 0xff) : 0u;
    if (err_hit) out->solved = 0;
    if (*d.err & kErrTableFull) out->err = 0xFE;  // not a move error: a probe sequence ran through its whole table
    out->min_len = *d.min_len;
    if (reset_tags) {  // the batch's success / error tags back to "none" for the next batch (saves a memset launch per batch)
        *d.solved_tag = kNoTag;
        *d.shorter_tag = kNoTag;
        *d.err_tag = kNoTag;
    }
    if (cur) {
        cursor_advance(cur, *out, pbegin, np, base);
        if (snap) *snap = *cur;
    }
}
template <typename W>
__global__ void k_decide_tab(SearchDev<W> d, uint32_t m, uint32_t np, uint32_t pbegin, uint32_t base, uint32_t cap_nodes, long long max_nodes,
                             const uint32_t* __restrict__ total_in, Decision* __restrict__ out, int reset_tags = 0, BfsCursor* cur = nullptr,
                             BfsCursor* __restrict__ snap = nullptr) {
    ACX_VGPR_PAD("v23");
    decide_tab_body<W>(d, m, np, pbegin, base, cap_nodes, max_nodes, total_in, out, reset_tags, cur, snap);
}

// root node: id 0
template <typename W> __global__ void k_root(SearchDev<W> d, W k0, W k1, uint32_t tl) {
    ACX_VGPR_PAD("v15");
    d.k0[0] = k0;
    d.k1[0] = k1;
    d.parent[0] = kEmpty;
    d.act[0] = 0xff;
    d.tlen[0] = (uint8_t)tl;
    d.depth[0] = 0;
    d.slots[(uint32_t)hash_key<W>(k0, k1) & d.smask] = 0;
}

// path of node `id` from the root, written root first: out_act / out_len [depth + 1]
template <typename W> __global__ void k_path(SearchDev<W> d, uint32_t id, int32_t* out_act, int32_t* out_len, int64_t cap) {
    ACX_VGPR_PAD("v15");
    const uint32_t dep = d.depth[id];
    for (uint32_t v = id, k = dep;; k--) {
        if ((int64_t)k < cap) {
            out_act[k] = d.act[v] == 0xff ? -1 : (int32_t)d.act[v];
            out_len[k] = d.tlen[v];
        }
        if (k == 0) break;
        v = d.parent[v];
    }
}

// Order a heap bucket by the signed state tuple (greedy.py:104-113 heap key, third field): rank sort.
// Thread i counts the bucket entries that sort before its own; states inside a bucket are pairwise
// distinct (they passed the visited set), so ranks are a permutation.  Keys are staged through LDS in
// tiles of 256 so that every comparison reads one broadcast LDS row.  O(n^2), buckets are small.
template <typename W>
__global__ void __launch_bounds__(256) k_rank_sort(SearchDev<W> d, const uint32_t* __restrict__ in, uint32_t n, uint32_t* __restrict__ out) {
    ACX_VGPR_PAD_W(W, "v39", "v47");
    __shared__ W t0[256];
    __shared__ W t1[256];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t mine = i < n ? in[i] : 0;
    Pres<W> me;
    key_to_pres<W>(d.k0[mine], d.k1[mine], me);
    uint32_t rank = 0;
    for (uint32_t base = 0; base < n; base += 256) {
        const uint32_t j = base + threadIdx.x;
        if (j < n) {
            const uint32_t id = in[j];
            t0[threadIdx.x] = d.k0[id];
            t1[threadIdx.x] = d.k1[id];
        }
        __syncthreads();
        const uint32_t cnt = n - base < 256 ? n - base : 256;
        for (uint32_t q = 0; q < cnt; q++) {
            Pres<W> o;
            key_to_pres<W>(t0[q], t1[q], o);
            rank += compare_pres<W>(o, me) < 0 ? 1u : 0u;
        }
        __syncthreads();
    }
    if (i < n) out[rank] = mine;
}


// exclusive prefix sum of n 32-bit flags (the batch-per-launch paths: greedy fallback, simplex graph).  tmp == nullptr: only the size
// of the temporary storage is returned in *tmp_bytes.  Defined once, in acx_search.hip (three small kernels of the library's own).
int scan_u32_exclusive(void* tmp, size_t* tmp_bytes, const uint32_t* in, uint32_t* out, size_t n, hipStream_t st);

// ---------------------------------------------------------------------------------------- host ---
// Device blocks of finished searches are kept in one process-wide pool and handed to the next search: hipMalloc / hipFree
// cost milliseconds per call -- erratically up to seconds for the 10 GB arenas of a group of searches -- and hipFree
// synchronises the whole device, which would serialise the overlapped searches of acx_search_many.  (Round 1 kept a pool per
// host thread; worker threads then had to return their blocks before they ended, and every acx_search_many paid the
// allocations again.)  Blocks are keyed by the device they live on (a process may search on several GPUs: thread ranks of the
// sharded engine, acx_search after torch.cuda.set_device); per device the pool holds at most 60 % of the memory and stops
// caching once less than an eighth of the device is free; acx_release_cached_memory empties it.
struct BlockPool {
    static constexpr size_t kMaxCachedBlock = 48ull << 30;
    static constexpr size_t kMaxBlocks = 8192;
    static constexpr int kMaxDevices = 16;
    struct Block {
        void* p;
        size_t bytes;
        int dev;  // the device the block was allocated on: a block is only ever handed to a caller whose current device is that one
        uint32_t stamp_epoch;  // != 0: the block was the stamp table of a finished BFS whose stamps carry epochs <= this (StampBuf)
        uint64_t clean_tag;    // != 0: the block holds greedy slots in the layout this names, every one of them as a search expects it (GreedySlots)
    };
    std::mutex mu;
    std::vector<Block> blocks;
    size_t cached[kMaxDevices] = {};
    size_t max_cached[kMaxDevices] = {};  // 60 % of a device's memory (173 GB of an MI355X's 288), set at first use: the two sweeps of
                                          // bench.py (bfs, then greedy over the 1190 presentations) keep ~150 GB of arenas between runs
    static int current_device() {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) dev = 0;
        return dev;
    }
    // `stamp_epoch` (StampBuf): a former stamp table is preferred and its epoch returned; every other caller gets an untagged block
    // if there is one (a tagged block it takes loses its tag: its content becomes the caller's)
    // `clean_tag` (GreedySlots): in: the layout wanted, out: that tag if the block carries it (its content is then valid), else 0.
    void* take(size_t bytes, size_t* got, int dev, uint32_t* stamp_epoch = nullptr, uint64_t* clean_tag = nullptr) {
        std::lock_guard<std::mutex> lock(mu);
        size_t best = blocks.size();
        const uint64_t want_clean = clean_tag ? *clean_tag : 0;
        auto wanted = [&](size_t k) {  // a block whose content this caller can use
            return stamp_epoch ? blocks[k].stamp_epoch != 0 : (want_clean ? blocks[k].clean_tag == want_clean : false);
        };
        auto tagged = [&](size_t k) { return blocks[k].stamp_epoch != 0 || blocks[k].clean_tag != 0; };
        auto better = [&](size_t k) {
            if (best == blocks.size()) return true;
            if (wanted(k) != wanted(best)) return wanted(k);
            if (!wanted(k) && tagged(k) != tagged(best)) return !tagged(k);  // (somebody else's content is the last thing to overwrite)
            return blocks[k].bytes < blocks[best].bytes;
        };
        for (size_t k = 0; k < blocks.size(); k++)
            if (blocks[k].dev == dev && blocks[k].bytes >= bytes && blocks[k].bytes <= bytes + bytes / 2 + 4096 && better(k)) best = k;
        if (best == blocks.size()) {
            if (clean_tag) *clean_tag = 0;
            return nullptr;
        }
        void* p = blocks[best].p;
        *got = blocks[best].bytes;
        if (stamp_epoch) *stamp_epoch = blocks[best].stamp_epoch;
        if (clean_tag) *clean_tag = blocks[best].clean_tag == want_clean ? want_clean : 0;
        cached[dev] -= blocks[best].bytes;
        blocks[best] = blocks.back();
        blocks.pop_back();
        return p;
    }
    void give(void* p, size_t bytes, int dev, uint32_t stamp_epoch = 0, uint64_t clean_tag = 0) {
        {
            std::lock_guard<std::mutex> lock(mu);
            size_t free_b = 0, total_b = 0;
            const bool info = hipMemGetInfo(&free_b, &total_b) == hipSuccess;  // (of the CURRENT device: only used when that is `dev`)
            const bool mine = current_device() == dev;
            if (!max_cached[dev]) max_cached[dev] = info && mine ? total_b / 10 * 6 : (96ull << 30);
            // other users of the device (torch's allocator: a PPO run after a sweep) must not starve behind this cache: once less
            // than an eighth of the device is free, blocks go back to the driver instead of into the pool
            const bool roomy = !info || !mine || free_b > total_b / 8;
            if (roomy && bytes <= kMaxCachedBlock && cached[dev] + bytes <= max_cached[dev] && blocks.size() < kMaxBlocks) {
                blocks.push_back({p, bytes, dev, stamp_epoch, clean_tag});
                cached[dev] += bytes;
                return;
            }
        }
        (void)hipFree(p);
    }
    size_t cached_on(int dev) {
        std::lock_guard<std::mutex> lock(mu);
        return cached[dev];
    }
    void trim() {
        std::vector<Block> old;
        {
            std::lock_guard<std::mutex> lock(mu);
            old.swap(blocks);
            for (size_t& c : cached) c = 0;
        }
        for (auto& b : old) (void)hipFree(b.p);
    }
    // no destructor work: what the pool still holds at process exit is released with the context (calling hipFree during
    // runtime teardown can block)
};
inline BlockPool& block_pool() {
    static BlockPool* pool = new BlockPool();  // never destroyed (see above)
    return *pool;
}

// Bytes one group of searches (acx_search_many) may allocate: `want`, but never more than an eighth of what the device can
// still give (free memory + this pool's cached blocks) -- up to seven callers run groups side by side (the seven relator
// widths of the Miller-Schupp sweep, ac_solver/search/_common.py:run_search_groups).
inline double group_byte_budget(double want) {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return want;
    const double avail = (double)free_b + (double)block_pool().cached_on(BlockPool::current_device());
    return std::max(256e6, std::min(want, avail / 8.0));
}

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    int dev = 0;
    uint64_t clean_tag = 0;  // what the block goes back to the pool as (GreedySlots: set once every slot in it is clean again)
    // `want_clean`: in: a content tag; out: that tag if the block already carries it, else 0
    int alloc(size_t b, uint64_t* want_clean = nullptr) {
        const size_t want = b ? b : 1;
        dev = BlockPool::current_device();
        p = block_pool().take(want, &bytes, dev, nullptr, want_clean);
        if (p) return ACX_OK;
        bytes = want;
        if (hipMalloc(&p, bytes) != hipSuccess) {
            block_pool().trim();  // give cached blocks back and retry once
            if (hipMalloc(&p, bytes) != hipSuccess) {
                p = nullptr;
                return fail(ACX_E_NOMEM, "hipMalloc(%zu) failed", bytes);
            }
        }
        return ACX_OK;
    }
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;  // (an owner: a copy would hand the block back twice)
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p(o.p), bytes(o.bytes), dev(o.dev), clean_tag(o.clean_tag) { o.p = nullptr; }
    DevBuf& operator=(DevBuf&& o) noexcept {
        if (this != &o) {
            release();
            p = o.p, bytes = o.bytes, dev = o.dev, clean_tag = o.clean_tag;
            o.p = nullptr;
        }
        return *this;
    }
    // back to the pool now (the buffer can be allocated again)
    void release() {
        if (p) block_pool().give(p, bytes, dev, 0, clean_tag);
        p = nullptr;
        bytes = 0;
        clean_tag = 0;
    }
    ~DevBuf() { release(); }
};

// The stamp table(s) of a BFS (acx_bfs.h).  A stamp carries its search's EPOCH, and a slot with another epoch is free: the table of
// a finished search goes back to the pool tagged with its epoch, and the next search that gets it runs with the next epoch WITHOUT
// refilling it (round 4: a 2.1 GB fill per 1e8-node search = 0.4 of its 9 ms; 39.5 GB per Miller-Schupp sweep).  Epochs 1 .. 254;
// a fresh block, or one whose epochs are used up, is filled with 0xFF (epoch field 255 = never a search's) on `st` and starts at 1.
struct StampBuf {
    void* p = nullptr;
    size_t bytes = 0;
    int dev = 0;
    uint32_t epoch = 0;
    int alloc(size_t b, hipStream_t st) {
        const size_t want = b ? b : 1;
        dev = BlockPool::current_device();
        uint32_t last = 0;
        p = block_pool().take(want, &bytes, dev, &last);
        if (!p) {
            bytes = want;
            if (hipMalloc(&p, bytes) != hipSuccess) {
                block_pool().trim();
                if (hipMalloc(&p, bytes) != hipSuccess) {
                    p = nullptr;
                    return fail(ACX_E_NOMEM, "hipMalloc(%zu) failed", bytes);
                }
            }
        }
        if (last >= 1 && last < 254) {
            epoch = last + 1;
            return ACX_OK;
        }
        // the WHOLE block, not only the `want` bytes in use: a later, larger table that gets this block with its epoch tag must not find
        // bytes that were never filled (or stamps from before an epoch wrap) beyond this table's end
        // (the tag only once the fill is QUEUED: a block whose fill failed goes back untagged -- epoch 0 -- and is filled by whoever takes it next)
        ACX_HIP_TRY(hipMemsetAsync(p, 0xff, bytes, st));
        epoch = 1;
        filled_on = st;
        fresh = true;
        return ACX_OK;
    }
    hipStream_t filled_on = nullptr;
    bool fresh = false;  // this owner queued the fill: a give-back before that stream has drained (an error return right behind alloc) must not
                         // hand the block to another stream as "clean"
    ~StampBuf() {
        if (!p) return;
        if (fresh && hipStreamQuery(filled_on) != hipSuccess) (void)hipStreamSynchronize(filled_on);
        block_pool().give(p, bytes, dev, epoch);
    }
};

// two timing events, handed back on every exit path; finished pairs are kept for the next search (per device: an event belongs
// to the device it was created on)
struct EventPair {
    hipEvent_t a = nullptr, b = nullptr;
    int dev = -1;
    struct Pool {
        std::mutex mu;
        std::vector<EventPair*> unused;  // (owned copies)
        std::vector<std::array<void*, 3>> free_;  // a, b, dev
    };
    static Pool& pool() {
        static Pool* p = new Pool();  // never destroyed: no HIP calls at process exit
        return *p;
    }
    hipError_t create() {
        if (hipGetDevice(&dev) != hipSuccess) dev = 0;
        {
            Pool& P = pool();
            std::lock_guard<std::mutex> lock(P.mu);
            for (size_t i = 0; i < P.free_.size(); i++)
                if ((int)(intptr_t)P.free_[i][2] == dev) {
                    a = (hipEvent_t)P.free_[i][0];
                    b = (hipEvent_t)P.free_[i][1];
                    P.free_[i] = P.free_.back();
                    P.free_.pop_back();
                    return hipSuccess;
                }
        }
        hipError_t e = hipEventCreate(&a);
        return e != hipSuccess ? e : hipEventCreate(&b);
    }
    ~EventPair() {
        if (a && b) {
            Pool& P = pool();
            std::lock_guard<std::mutex> lock(P.mu);
            P.free_.push_back({(void*)a, (void*)b, (void*)(intptr_t)dev});
            return;
        }
        if (a) (void)hipEventDestroy(a);
        if (b) (void)hipEventDestroy(b);
    }
};

// pinned host staging, one grow-only buffer per host thread (hipHostMalloc is as slow as hipMalloc)
inline uint8_t* pinned_staging(size_t bytes) {
    static thread_local uint8_t* buf = nullptr;
    static thread_local size_t cap = 0;
    if (bytes <= cap) return buf;
    if (buf) (void)hipHostFree(buf);
    buf = nullptr;
    cap = 0;
    const size_t want = bytes + bytes / 2 + 4096;
    if (hipHostMalloc((void**)&buf, want, hipHostMallocDefault) != hipSuccess) return nullptr;
    cap = want;
    return buf;
}

}  // namespace acx
