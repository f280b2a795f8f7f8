// acx_search.hip -- device-resident BFS / greedy frontier over the AC graph (single GPU).
//
// Replaces bfs (ac_solver/search/breadth_first.py:15-97) and greedy_search (search/greedy.py:15-121)
// with the SAME visiting order and therefore the same (solved, path) result:
//
//   * a search advances in BATCHES of parents that the reference would pop consecutively
//       bfs     a slice of the current FIFO level (node ids are FIFO order)
//       greedy  a prefix of the heap's minimal (total length, depth) bucket, in signed state order
//   * every (parent, action) pair of the batch is one lane of k_expand: tag = 12 * parent_pos + action is
//     exactly the order in which the reference generates children
//   * duplicates are resolved to the MINIMUM tag (the reference's "first discoverer wins",
//     breadth_first.py:87-89): the open-addressed table stores node ids; a candidate claims an empty slot
//     with a 32-bit CAS on a provisional id (2^31 + tag) and equal keys fold with atomicMin, the full
//     key of the occupant is always compared (exact set, no fingerprints)
//   * winners are numbered by an exclusive scan in tag order, which reproduces the reference's insertion
//     order; the per-parent budget test (breadth_first.py:91-95) becomes "first parent whose cumulative
//     winner count reaches the budget"; the solved test (:84-85) "minimum tag with total length 2"
//   * greedy additionally stops a batch right after the first parent that inserts a NEW child shorter than
//     the bucket (that child is the heap's next minimum); later parents stay queued
//
// Keys: a relator word and its length share one machine word (length in the top 6 bits):
// W = u64 for L <= 29, u128 for L <= 61.  Roofline: HBM (random table probes); see DESIGN.md.
#include <string.h>

#include <cstring>

#include <rocprim/rocprim.hpp>

#include <stdlib.h>

#include <algorithm>
#include <map>
#include <memory>
#include <vector>

#include "acx_common.h"
#include "acx_word.h"

namespace acx {

constexpr uint32_t kEmpty = 0xFFFFFFFFu;
constexpr uint32_t kProv = 0x80000000u;  // provisional id = kProv | tag (candidate of the running batch)
constexpr uint64_t kNoTag = ~0ull;

template <typename W> struct keyops {
    static constexpr int kShift = wtraits<W>::kBits - 6;
    static ACX_HD W make(W w, int n) { return w | ((W)n << kShift); }
    static ACX_HD int len(W k) { return (int)(uint32_t)(k >> kShift); }
    static ACX_HD W word(W k) { return k & (((W)1 << kShift) - 1); }
};

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

// Visited set of the BFS frontiers: open addressing with the FULL key inline, one entry per 32-byte (u64 keys) /
// 64-byte (u128 keys) sector, so a probe is ONE random memory access whether the slot is empty, holds another key
// or holds this key.  `stamp` = epoch << 32 | tag of the candidate that claimed the entry; an entry whose epoch is not
// the running batch's is a committed state, an entry of the running epoch is provisional and folds to the minimum tag
// among equal keys (64-bit atomicMin).  The table never stores node ids: BFS only asks "seen before?".
template <typename W> struct TabEntry;
template <> struct alignas(32) TabEntry<uint64_t> {
    uint64_t k0, k1;
    unsigned long long stamp;
    uint64_t pad;
};
template <> struct alignas(64) TabEntry<u128> {
    u128 k0, k1;
    unsigned long long stamp;
    uint64_t pad[3];
};
constexpr unsigned long long kStampEmpty = ~0ull;

template <typename W> struct SearchDev {
    TabEntry<W>* tab;  // BFS visited table (inline keys); tmask = entries - 1
    uint32_t tmask;
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
    // device scalars
    unsigned long long* solved_tag;   // min tag with total length 2
    unsigned long long* shorter_tag;  // greedy: min tag of a NEW child shorter than the bucket
    unsigned long long* rank_tag;     // tag of the r-th winner (budget crossing)
    uint32_t* err;
    uint32_t* min_len;
    int32_t L;
    int32_t cyclical;
};

template <typename W> __device__ __forceinline__ void key_to_pres(W k0, W k1, Pres<W>& s) {
    s.w0 = keyops<W>::word(k0);
    s.n0 = keyops<W>::len(k0);
    s.w1 = keyops<W>::word(k1);
    s.n1 = keyops<W>::len(k1);
}

// one lane per (parent, action): tag t = 12 * p + a
template <typename W>
__global__ void __launch_bounds__(256) k_expand(SearchDev<W> d, const uint32_t* __restrict__ plist, uint32_t pbegin, uint32_t np) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t tl = 0xFFFFFFFFu;
    if (t < 12u * np) {
        const uint32_t p = t / 12u, a = t - 12u * p;
        const uint32_t pid = plist ? plist[p] : pbegin + p;
        Pres<W> s;
        const W pk0 = d.k0[pid], pk1 = d.k1[pid];
        key_to_pres<W>(pk0, pk1, s);
        const int e = apply_move<W, true>(s, (int)a, d.L, d.cyclical != 0);
        if (e) atomicOr(d.err, (uint32_t)e);  // the reference's ACMove raises: the whole search raises
        const W c0 = keyops<W>::make(s.w0, s.n0), c1 = keyops<W>::make(s.w1, s.n1);
        d.ck0[t] = c0;
        d.ck1[t] = c1;
        d.cknown[t] = (c0 == pk0 && c1 == pk1) ? 1 : 0;  // an unchanged state is its (visited) parent: k_insert skips the probe
        tl = (uint32_t)(s.n0 + s.n1);
        d.clen[t] = (uint8_t)tl;
        if (tl == 2) atomicMin(d.solved_tag, (unsigned long long)t);  // breadth_first.py:84 / greedy.py:91
    }
    // wave-level min before the atomic keeps contention low (all 64 lanes take part)
    uint32_t m = tl;
    for (int o = 32; o > 0; o >>= 1) m = min(m, (uint32_t)__shfl_xor((int)m, o));
    // one contended address: only waves that would actually lower the minimum issue the atomic (a plain, possibly
    // stale read can only over-estimate the current minimum, so no update is lost)
    if ((threadIdx.x & 63) == 0 && m < *(volatile uint32_t*)d.min_len) atomicMin(d.min_len, m);
}

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
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= m) return;
    if (skip_known && d.cknown[t]) {
        d.cslot[t] = kEmpty;
        return;
    }
    const W k0 = d.ck0[t], k1 = d.ck1[t];
    const uint32_t me = kProv | t;
    uint32_t h = (uint32_t)hash_key<W>(k0, k1) & mask;
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
    }
    d.cslot[t] = h;
}

// read-only membership test against the visited table (greedy: speculative batches must not touch it)
template <typename W> __global__ void __launch_bounds__(256) k_lookup(SearchDev<W> d, uint32_t m) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= m) return;
    const W k0 = d.ck0[t], k1 = d.ck1[t];
    uint32_t h = (uint32_t)hash_key<W>(k0, k1) & d.smask;
    uint8_t known = 0;
    for (;;) {
        const uint32_t v = d.slots[h];
        if (v == kEmpty) break;
        if (key_equals<W>(d, v, k0, k1)) {
            known = 1;
            break;
        }
        h = (h + 1) & d.smask;
    }
    d.cknown[t] = known;
}

// cflag[t] = 1 iff candidate t is the first discoverer of a state not seen before
template <typename W>
__global__ void __launch_bounds__(256) k_mark(SearchDev<W> d, const uint32_t* __restrict__ slots, uint32_t m, int bucket_len) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= m) return;
    const uint32_t s = d.cslot[t];
    const uint32_t win = (s != kEmpty && slots[s] == (kProv | t)) ? 1u : 0u;
    d.cflag[t] = win;
    if (win && bucket_len >= 0 && (int)d.clen[t] < bucket_len) atomicMin(d.shorter_tag, (unsigned long long)t);
}

// BFS: insert candidate t into the inline-key table (see TabEntry).  cslot[t] = entry that holds the key, kEmpty when
// the candidate was skipped (a child equal to its parent).
template <typename W>
__global__ void __launch_bounds__(256) k_insert_tab(SearchDev<W> d, uint32_t m, uint32_t epoch, int skip_known) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= m) return;
    if (skip_known && d.cknown[t]) {
        d.cslot[t] = kEmpty;
        return;
    }
    const W k0 = d.ck0[t], k1 = d.ck1[t];
    const unsigned long long me = ((unsigned long long)epoch << 32) | t;
    uint32_t h = (uint32_t)hash_key<W>(k0, k1) & d.tmask;
    for (;;) {
        TabEntry<W>* e = d.tab + h;
        const W e0 = e->k0, e1 = e->k1;  // one sector together with the stamp
        unsigned long long st = e->stamp;
        if (st == kStampEmpty) {
            st = atomicCAS(&e->stamp, kStampEmpty, me);
            if (st == kStampEmpty) {  // claimed: the key moves in (readers of this batch compare through the candidate arena)
                e->k0 = k0;
                e->k1 = k1;
                break;
            }
            // lost the race: `st` is a stamp of the running epoch now
        }
        if ((uint32_t)(st >> 32) == epoch) {
            const uint32_t o = (uint32_t)st;
            if (d.ck0[o] == k0 && d.ck1[o] == k1) {
                if (st > me) atomicMin(&e->stamp, me);
                break;
            }
        } else if (e0 == k0 && e1 == k1) {
            break;  // a committed state
        }
        h = (h + 1) & d.tmask;
    }
    d.cslot[t] = h;
}

template <typename W> __global__ void __launch_bounds__(256) k_mark_tab(SearchDev<W> d, uint32_t m, uint32_t epoch) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= m) return;
    const uint32_t s = d.cslot[t];
    d.cflag[t] = (s != kEmpty && d.tab[s].stamp == (((unsigned long long)epoch << 32) | t)) ? 1u : 0u;
}

template <typename W> __global__ void k_root_tab(SearchDev<W> d, W k0, W k1, uint32_t tl) {
    d.k0[0] = k0;
    d.k1[0] = k1;
    d.parent[0] = kEmpty;
    d.act[0] = 0xff;
    d.tlen[0] = (uint8_t)tl;
    d.depth[0] = 0;
    TabEntry<W>* e = d.tab + ((uint32_t)hash_key<W>(k0, k1) & d.tmask);
    e->k0 = k0;
    e->k1 = k1;
    e->stamp = 0;  // epoch 0 is never a running batch
}

// tag of the winner with 1-based rank r (exists and is unique)
template <typename W> __global__ void __launch_bounds__(256) k_find_rank(SearchDev<W> d, uint32_t m, uint32_t r) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= m) return;
    if (d.cflag[t] && d.cpos[t] + 1 == r) *d.rank_tag = t;
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
    out->err = *d.err;
    out->min_len = *d.min_len;
}

// Winners below `cutoff` become nodes base + cpos[t].  BFS: their table slot (already claimed in the visited
// table) is rewritten to the final id.  Greedy: the key is inserted into the visited table now (it is known
// to be absent and the committed keys are pairwise distinct, so a plain CAS claim is enough).
template <typename W>
__global__ void __launch_bounds__(256) k_commit(SearchDev<W> d, const uint32_t* __restrict__ plist, uint32_t pbegin, const Decision* __restrict__ dec, uint32_t m,
                                                uint32_t base, int insert_now) {
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
        uint32_t h = (uint32_t)hash_key<W>(k0, k1) & d.smask;
        while (atomicCAS(&d.slots[h], kEmpty, id) != kEmpty) h = (h + 1) & d.smask;
    }
    // BFS (insert_now == 0): the inline-key table already holds the key under this batch's epoch; nothing to rewrite
}

// root node: id 0
template <typename W> __global__ void k_root(SearchDev<W> d, W k0, W k1, uint32_t tl) {
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

}  // namespace acx
#include "acx_greedy.h"
namespace acx {

// ---------------------------------------------------------------------------------------- host ---
// Device blocks of finished searches are kept per host thread and handed to the next search of that thread:
// hipMalloc / hipFree cost milliseconds and hipFree synchronises the whole device, which would serialise the
// overlapped searches of acx_search_many.  Blocks above kMaxCachedBlock go back to the driver at once.
struct BlockPool {
    static constexpr size_t kMaxCachedBlock = 16ull << 30;  // (a thread's cache is trimmed when acx_search_many returns)
    static constexpr size_t kMaxBlocks = 8192;
    std::vector<std::pair<void*, size_t>> blocks;
    void* take(size_t bytes, size_t* got) {
        size_t best = blocks.size();
        for (size_t k = 0; k < blocks.size(); k++)
            if (blocks[k].second >= bytes && blocks[k].second <= bytes + bytes / 2 + 4096 && (best == blocks.size() || blocks[k].second < blocks[best].second)) best = k;
        if (best == blocks.size()) return nullptr;
        void* p = blocks[best].first;
        *got = blocks[best].second;
        blocks[best] = blocks.back();
        blocks.pop_back();
        return p;
    }
    void give(void* p, size_t bytes) {
        if (bytes > kMaxCachedBlock || blocks.size() >= kMaxBlocks) (void)hipFree(p);
        else blocks.emplace_back(p, bytes);
    }
    void trim() {
        for (auto& b : blocks) (void)hipFree(b.first);
        blocks.clear();
    }
    // no destructor work: a worker thread trims explicitly before it ends; what the main thread still holds at process
    // exit is released with the context (calling hipFree during runtime teardown can block)
};
static BlockPool& block_pool() {
    static thread_local BlockPool pool;
    return pool;
}

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    int alloc(size_t b) {
        const size_t want = b ? b : 1;
        p = block_pool().take(want, &bytes);
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
    ~DevBuf() {
        if (p) block_pool().give(p, bytes);
    }
};

// pinned host staging, one grow-only buffer per host thread (hipHostMalloc is as slow as hipMalloc)
static uint8_t* pinned_staging(size_t bytes) {
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

struct Scalars {
    unsigned long long solved_tag, shorter_tag, rank_tag;
    uint32_t err, min_len;
};

template <typename W> struct Searcher {
    SearchDev<W> d;
    DevBuf arena_nodes, arena_cand, arena_tab, arena_btab, arena_scal, arena_tmp, arena_list, arena_path;
    size_t tmp_bytes = 0;
    uint64_t cap_nodes = 0, cap_cand = 0, n_slots = 0, n_bslots = 0;
    hipStream_t st = nullptr;
    Decision* d_dec = nullptr;   // device
    uint8_t* h_pin = nullptr;    // pinned host staging: Decision followed by the total lengths of the new nodes
    size_t h_pin_bytes = 0;

    ~Searcher() {
        if (st) (void)hipStreamDestroy(st);
    }

    // inline_tab: BFS visited table with inline keys (TabEntry); otherwise the id table of the greedy paths
    int init(int L, int cyclical, int64_t max_nodes, uint32_t batch_parents, bool greedy, bool inline_tab = false) {
        memset(&d, 0, sizeof(d));
        // every search owns a stream, so that searches driven from different host threads overlap on the GPU
        ACX_HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        d.L = L;
        d.cyclical = cyclical;
        cap_nodes = (uint64_t)max_nodes + 64;
        cap_cand = 12ull * batch_parents;
        n_slots = 1024;
        while (n_slots < 2 * (cap_nodes + cap_cand)) n_slots <<= 1;
        if (n_slots > (1ull << 31)) return fail(ACX_E_INVAL, "acx_search: budget too large for 32-bit node ids");
        n_bslots = 1024;
        while (greedy && n_bslots < 2 * cap_cand) n_bslots <<= 1;
        size_t o = 0;
        auto take = [&](uint8_t* base, size_t bytes) {
            uint8_t* p = base ? base + o : nullptr;
            o += (bytes + 255) / 256 * 256;
            return p;
        };
        for (int pass = 0; pass < 2; pass++) {
            uint8_t* b = (uint8_t*)arena_nodes.p;
            o = 0;
            d.k0 = (W*)take(b, cap_nodes * sizeof(W));
            d.k1 = (W*)take(b, cap_nodes * sizeof(W));
            d.parent = (uint32_t*)take(b, cap_nodes * 4);
            d.depth = (uint32_t*)take(b, cap_nodes * 4);
            d.act = (uint8_t*)take(b, cap_nodes);
            d.tlen = (uint8_t*)take(b, cap_nodes);
            if (pass == 0 && arena_nodes.alloc(o)) return ACX_E_NOMEM;
        }
        for (int pass = 0; pass < 2; pass++) {
            uint8_t* b = (uint8_t*)arena_cand.p;
            o = 0;
            d.ck0 = (W*)take(b, cap_cand * sizeof(W));
            d.ck1 = (W*)take(b, cap_cand * sizeof(W));
            d.cslot = (uint32_t*)take(b, cap_cand * 4);
            d.cflag = (uint32_t*)take(b, cap_cand * 4);
            d.cpos = (uint32_t*)take(b, cap_cand * 4);
            d.clen = (uint8_t*)take(b, cap_cand);
            d.cknown = (uint8_t*)take(b, cap_cand);
            if (pass == 0 && arena_cand.alloc(o)) return ACX_E_NOMEM;
        }
        if (inline_tab) {
            if (arena_tab.alloc(n_slots * sizeof(TabEntry<W>))) return ACX_E_NOMEM;
            d.tab = (TabEntry<W>*)arena_tab.p;
            d.tmask = (uint32_t)(n_slots - 1);
        } else {
            if (arena_tab.alloc(n_slots * 4)) return ACX_E_NOMEM;
            d.slots = (uint32_t*)arena_tab.p;
            d.smask = (uint32_t)(n_slots - 1);
        }
        if (greedy) {
            if (arena_btab.alloc(n_bslots * 4)) return ACX_E_NOMEM;
            d.bslots = (uint32_t*)arena_btab.p;
            d.bmask = (uint32_t)(n_bslots - 1);
        }
        if (arena_scal.alloc(256)) return ACX_E_NOMEM;
        uint8_t* sc = (uint8_t*)arena_scal.p;
        d_dec = (Decision*)(sc + 64);
        h_pin_bytes = sizeof(Decision) + 64 + cap_cand;
        h_pin = pinned_staging(h_pin_bytes);
        if (!h_pin) return fail(ACX_E_NOMEM, "hipHostMalloc(%zu) failed", h_pin_bytes);
        d.solved_tag = (unsigned long long*)(sc + 0);
        d.shorter_tag = (unsigned long long*)(sc + 8);
        d.rank_tag = (unsigned long long*)(sc + 16);
        d.err = (uint32_t*)(sc + 24);
        d.min_len = (uint32_t*)(sc + 28);
        if (arena_list.alloc(std::max<uint64_t>(batch_parents, 1024) * 4 * 2)) return ACX_E_NOMEM;
        if (arena_path.alloc(8)) return ACX_E_NOMEM;
        // rocprim temporary storage for the scan over one batch
        size_t need = 0;
        if (rocprim::exclusive_scan(nullptr, need, d.cflag, d.cpos, 0u, cap_cand, rocprim::plus<uint32_t>(), st) != hipSuccess)
            return fail(ACX_E_NODEVICE, "rocprim::exclusive_scan sizing failed");
        tmp_bytes = need + 256;
        if (arena_tmp.alloc(tmp_bytes)) return ACX_E_NOMEM;
        ACX_HIP_TRY(hipMemsetAsync(arena_tab.p, 0xff, n_slots * (inline_tab ? sizeof(TabEntry<W>) : 4), st));
        ACX_HIP_TRY(hipMemsetAsync(arena_scal.p, 0xff, 256, st));
        ACX_HIP_TRY(hipMemsetAsync(d.err, 0, 4, st));
        return ACX_OK;
    }

    int read_scalars(Scalars& s) {
        uint8_t h[32];
        ACX_HIP_TRY(hipMemcpyAsync(h, arena_scal.p, 32, hipMemcpyDeviceToHost, st));
        ACX_HIP_TRY(hipStreamSynchronize(st));
        memcpy(&s.solved_tag, h + 0, 8);
        memcpy(&s.shorter_tag, h + 8, 8);
        memcpy(&s.rank_tag, h + 16, 8);
        memcpy(&s.err, h + 24, 4);
        memcpy(&s.min_len, h + 28, 4);
        return ACX_OK;
    }

    int reset_batch_scalars() {  // solved / shorter / rank tags back to "none"; err and min_len are sticky
        ACX_HIP_TRY(hipMemsetAsync(arena_scal.p, 0xff, 24, st));
        return ACX_OK;
    }

    int scan(uint32_t m, uint32_t& total) {
        size_t tb = tmp_bytes;
        if (rocprim::exclusive_scan(arena_tmp.p, tb, d.cflag, d.cpos, 0u, m, rocprim::plus<uint32_t>(), st) != hipSuccess)
            return fail(ACX_E_NODEVICE, "rocprim::exclusive_scan failed");
        uint32_t last[2];
        ACX_HIP_TRY(hipMemcpyAsync(&last[0], d.cpos + (m - 1), 4, hipMemcpyDeviceToHost, st));
        ACX_HIP_TRY(hipMemcpyAsync(&last[1], d.cflag + (m - 1), 4, hipMemcpyDeviceToHost, st));
        ACX_HIP_TRY(hipStreamSynchronize(st));
        total = last[0] + last[1];
        return ACX_OK;
    }

    // number of winners with tag < cutoff
    int winners_below(uint32_t cutoff, uint32_t m, uint32_t total, uint32_t& out) {
        if (cutoff >= m) {
            out = total;
            return ACX_OK;
        }
        ACX_HIP_TRY(hipMemcpyAsync(&out, d.cpos + cutoff, 4, hipMemcpyDeviceToHost, st));
        ACX_HIP_TRY(hipStreamSynchronize(st));
        return ACX_OK;
    }

    // first parent position whose cumulative winner count reaches `need` (need >= 1 and <= total)
    int budget_parent(uint32_t m, uint32_t need, uint32_t& parent_pos) {
        hipLaunchKernelGGL(k_find_rank<W>, dim3((m + 255) / 256), dim3(256), 0, st, d, m, need);
        unsigned long long t = 0;
        ACX_HIP_TRY(hipMemcpyAsync(&t, d.rank_tag, 8, hipMemcpyDeviceToHost, st));
        ACX_HIP_TRY(hipStreamSynchronize(st));
        parent_pos = (uint32_t)(t / 12);
        return ACX_OK;
    }

    int path_of(uint32_t id, uint32_t depth, int32_t* pa, int32_t* pl, int64_t cap, int64_t* n) {
        const int64_t len = (int64_t)depth + 1;
        *n = len;
        const int64_t w = std::min<int64_t>(len, cap);
        if (w <= 0) return ACX_OK;
        DevBuf buf;
        if (buf.alloc((size_t)w * 8)) return ACX_E_NOMEM;
        int32_t* da = (int32_t*)buf.p;
        int32_t* dl = da + w;
        hipLaunchKernelGGL(k_path<W>, dim3(1), dim3(1), 0, st, d, id, da, dl, w);
        ACX_HIP_TRY(hipMemcpyAsync(pa, da, w * 4, hipMemcpyDeviceToHost, st));
        ACX_HIP_TRY(hipMemcpyAsync(pl, dl, w * 4, hipMemcpyDeviceToHost, st));
        ACX_HIP_TRY(hipStreamSynchronize(st));
        return ACX_OK;
    }

    int node_field(uint32_t id, uint32_t& parent, uint32_t& depth) {
        ACX_HIP_TRY(hipMemcpyAsync(&parent, d.parent + id, 4, hipMemcpyDeviceToHost, st));
        ACX_HIP_TRY(hipMemcpyAsync(&depth, d.depth + id, 4, hipMemcpyDeviceToHost, st));
        ACX_HIP_TRY(hipStreamSynchronize(st));
        return ACX_OK;
    }
};

static int err_to_rc(uint32_t e) {
    return fail(ACX_E_ROWERR, "a move emptied a relator during the search: the reference raises %s here",
                (e & ACX_ERR_INDEX) && !(e & ACX_ERR_ASSERT) ? "IndexError" : "AssertionError");
}

// Device buffers of one greedy search on the persistent frontier
template <typename W> struct GreedySearch {
    Searcher<W> S;
    DevBuf bk, bitmap, arena, gk0, gk1, gid, fpb;
    GreedyDev<W> g;
    // `st`: stream for the bucket-table memsets (nullptr = the search's own stream, S.st)
    int setup(const Pres<W>& root, int L, int64_t max_nodes, int cyclical, hipStream_t st) {
        int rc = S.init(L, cyclical, max_nodes, 1024, false);
        if (rc) return rc;
        if (!st) st = S.st;
        g.d = S.d;
        g.nlen = (uint32_t)(2 * L + 1);
        g.max_nodes = (long long)max_nodes;
        g.root_len = (uint32_t)(root.n0 + root.n1);
        const uint64_t arena_entries = std::min<uint64_t>(8ull * (uint64_t)std::max<int64_t>(max_nodes, 1) + (1ull << 20), 1ull << 31);
        g.arena_cap = (uint32_t)arena_entries;
        if (fpb.alloc((S.n_slots + 8) * 2)) return ACX_E_NOMEM;  // never read behind an empty slot, so no initialisation
        g.fp = (uint16_t*)fpb.p;
        g.root_k0 = keyops<W>::make(root.w0, root.n0);
        g.root_k1 = keyops<W>::make(root.w1, root.n1);
        const size_t sort_cap = (size_t)std::max<int64_t>(max_nodes, 1) + 64;  // a bucket never holds more than all nodes
        if (gk0.alloc(sort_cap * sizeof(W)) || gk1.alloc(sort_cap * sizeof(W)) || gid.alloc(sort_cap * 4)) return ACX_E_NOMEM;
        g.gk0 = (W*)gk0.p;
        g.gk1 = (W*)gk1.p;
        g.gid = (uint32_t*)gid.p;
        const size_t bk_bytes = (size_t)g.nlen * kDepthCap * sizeof(BucketRec), bm_bytes = (size_t)g.nlen * (kDepthCap / 32) * 4;
        if (bk.alloc(bk_bytes) || bitmap.alloc(bm_bytes) || arena.alloc(arena_entries * 4)) return ACX_E_NOMEM;
        g.bk = (BucketRec*)bk.p;
        g.bitmap = (uint32_t*)bitmap.p;
        g.arena = (uint32_t*)arena.p;
        ACX_HIP_TRY(hipMemsetAsync(bk.p, 0, bk_bytes, st));
        ACX_HIP_TRY(hipMemsetAsync(bitmap.p, 0, bm_bytes, st));
        return ACX_OK;
    }
};

// A group of independent greedy searches in one launch of k_greedy_multi (one workgroup each).  rc_out[k] = ACX_OK /
// ACX_E_CAPACITY (path buffer) / ACX_E_ROWERR; need_rerun[k] = 1 when search k must be repeated on the single-search path
// (a capacity of the persistent kernel was exceeded).
template <typename W>
static int run_greedy_group(const int8_t* rows, int64_t n, int L, int64_t max_nodes, int cyclical, int32_t* solved, int32_t* path_action, int32_t* path_len,
                            int64_t path_cap, int64_t* path_n, acx_search_stats* stats, int32_t* rc_out, uint8_t* need_rerun) {
    // ONE device allocation for the whole group (thousands of hipMalloc calls would cost more than the searches):
    // per search the node arrays, the id table + fingerprints, the bucket table / bitmap / arena and the sort scratch;
    // the regions that need initialising (tables, bucket records, bitmaps) are contiguous over the group
    const uint64_t cap_nodes = (uint64_t)max_nodes + 64 + 12 * 1024;
    uint64_t n_slots = 1024;
    while (n_slots < 2 * (cap_nodes + 12288)) n_slots <<= 1;
    if (n_slots > (1ull << 31)) return fail(ACX_E_INVAL, "acx_search_many: budget too large for 32-bit node ids");
    const uint32_t nlen = (uint32_t)(2 * L + 1);
    const uint64_t arena_entries = std::min<uint64_t>(8ull * (uint64_t)std::max<int64_t>(max_nodes, 1) + (1ull << 20), 1ull << 31);
    const uint64_t sort_cap = (uint64_t)std::max<int64_t>(max_nodes, 1) + 64;
    auto up = [](uint64_t b) { return (b + 255) / 256 * 256; };
    const uint64_t b_slots = up(n_slots * 4), b_bk = up((uint64_t)nlen * kDepthCap * sizeof(BucketRec)), b_bm = up((uint64_t)nlen * (kDepthCap / 32) * 4);
    const uint64_t b_fp = up((n_slots + 8) * 2), b_arena = up(arena_entries * 4), b_key = up(cap_nodes * sizeof(W)), b_u32 = up(cap_nodes * 4), b_u8 = up(cap_nodes);
    const uint64_t b_gk = up(sort_cap * sizeof(W)), b_gid = up(sort_cap * 4);
    const uint64_t per_rest = b_fp + b_arena + 2 * b_key + 2 * b_u32 + 2 * b_u8 + 2 * b_gk + b_gid;
    const uint64_t total = (uint64_t)n * (b_slots + b_bk + b_bm + per_rest);
    DevBuf big;
    if (big.alloc(total)) return ACX_E_NOMEM;
    uint8_t* base = (uint8_t*)big.p;
    uint8_t* p_slots = base;
    uint8_t* p_bk = p_slots + (uint64_t)n * b_slots;
    uint8_t* p_bm = p_bk + (uint64_t)n * b_bk;
    uint8_t* p_rest = p_bm + (uint64_t)n * b_bm;
    std::vector<GreedyDev<W>> hdev((size_t)n);
    hipStream_t st = nullptr;
    ACX_HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    struct StreamGuard {
        hipStream_t s;
        ~StreamGuard() { (void)hipStreamDestroy(s); }
    } guard{st};
    ACX_HIP_TRY(hipMemsetAsync(p_slots, 0xff, (uint64_t)n * b_slots, st));
    ACX_HIP_TRY(hipMemsetAsync(p_bk, 0, (uint64_t)n * (b_bk + b_bm), st));
    for (int64_t k = 0; k < n; k++) {
        need_rerun[k] = 0;
        rc_out[k] = ACX_OK;
        solved[k] = 0;
        path_n[k] = 0;
        Pres<W> root;
        bool ok = pack_relator<W>(rows + k * 2 * L, L, root.w0, root.n0);
        ok = pack_relator<W>(rows + k * 2 * L + L, L, root.w1, root.n1) && ok;
        if (!ok) return fail(ACX_E_ROWERR, "acx_search_many: presentation %lld is not a zero-padded word pair over {+-1,+-2}", (long long)k);
        GreedyDev<W>& g = hdev[k];
        memset(&g, 0, sizeof(g));
        uint8_t* q = p_rest + (uint64_t)k * per_rest;
        auto take = [&](uint64_t bytes) {
            uint8_t* r = q;
            q += bytes;
            return r;
        };
        g.d.slots = (uint32_t*)(p_slots + (uint64_t)k * b_slots);
        g.d.smask = (uint32_t)(n_slots - 1);
        g.bk = (BucketRec*)(p_bk + (uint64_t)k * b_bk);
        g.bitmap = (uint32_t*)(p_bm + (uint64_t)k * b_bm);
        g.fp = (uint16_t*)take(b_fp);
        g.arena = (uint32_t*)take(b_arena);
        g.d.k0 = (W*)take(b_key);
        g.d.k1 = (W*)take(b_key);
        g.d.parent = (uint32_t*)take(b_u32);
        g.d.depth = (uint32_t*)take(b_u32);
        g.d.act = take(b_u8);
        g.d.tlen = take(b_u8);
        g.gk0 = (W*)take(b_gk);
        g.gk1 = (W*)take(b_gk);
        g.gid = (uint32_t*)take(b_gid);
        g.d.L = L;
        g.d.cyclical = cyclical;
        g.arena_cap = (uint32_t)arena_entries;
        g.nlen = nlen;
        g.max_nodes = (long long)max_nodes;
        g.root_len = (uint32_t)(root.n0 + root.n1);
        g.root_k0 = keyops<W>::make(root.w0, root.n0);
        g.root_k1 = keyops<W>::make(root.w1, root.n1);
    }
    const int64_t pc = std::max<int64_t>(path_cap, 1);
    DevBuf ddev, douts, dpa, dpl;
    if (ddev.alloc((size_t)n * sizeof(GreedyDev<W>)) || douts.alloc((size_t)n * sizeof(GreedyOut)) || dpa.alloc((size_t)n * pc * 4) || dpl.alloc((size_t)n * pc * 4))
        return ACX_E_NOMEM;
    ACX_HIP_TRY(hipMemcpyAsync(ddev.p, hdev.data(), (size_t)n * sizeof(GreedyDev<W>), hipMemcpyHostToDevice, st));
    ACX_HIP_TRY(hipMemsetAsync(douts.p, 0, (size_t)n * sizeof(GreedyOut), st));
    hipEvent_t ev0, ev1;
    ACX_HIP_TRY(hipEventCreate(&ev0));
    ACX_HIP_TRY(hipEventCreate(&ev1));
    ACX_HIP_TRY(hipEventRecord(ev0, st));
    hipLaunchKernelGGL(k_greedy_multi<W>, dim3((unsigned)n), dim3(kGT), 0, st, (const GreedyDev<W>*)ddev.p, (GreedyOut*)douts.p, (int32_t*)dpa.p, (int32_t*)dpl.p,
                       (long long)pc);
    ACX_HIP_TRY(hipGetLastError());
    ACX_HIP_TRY(hipEventRecord(ev1, st));
    std::vector<GreedyOut> o((size_t)n);
    std::vector<int32_t> pa((size_t)n * pc), pl((size_t)n * pc);
    ACX_HIP_TRY(hipMemcpyAsync(o.data(), douts.p, (size_t)n * sizeof(GreedyOut), hipMemcpyDeviceToHost, st));
    ACX_HIP_TRY(hipMemcpyAsync(pa.data(), dpa.p, (size_t)n * pc * 4, hipMemcpyDeviceToHost, st));
    ACX_HIP_TRY(hipMemcpyAsync(pl.data(), dpl.p, (size_t)n * pc * 4, hipMemcpyDeviceToHost, st));
    ACX_HIP_TRY(hipStreamSynchronize(st));
    float ms = 0;
    ACX_HIP_TRY(hipEventElapsedTime(&ms, ev0, ev1));
    (void)hipEventDestroy(ev0);
    (void)hipEventDestroy(ev1);
    for (int64_t k = 0; k < n; k++) {
        const GreedyOut& r = o[k];
        if (r.status == GREEDY_FALLBACK) {
            need_rerun[k] = 1;
            continue;
        }
        if (r.status == GREEDY_MOVE_ERROR) {
            rc_out[k] = err_to_rc(r.err);
            continue;
        }
        if (r.status != GREEDY_SOLVED && r.status != GREEDY_BUDGET && r.status != GREEDY_EXHAUSTED) {
            rc_out[k] = fail(ACX_E_NODEVICE, "greedy frontier kernel ended in state %u", r.status);
            continue;
        }
        solved[k] = r.status == GREEDY_SOLVED ? 1 : 0;
        path_n[k] = r.path_n;
        if ((int64_t)r.path_n > path_cap) {
            rc_out[k] = fail(ACX_E_CAPACITY, "path has %u entries, buffer holds %lld", r.path_n, (long long)path_cap);
        } else if (path_action && path_len) {
            memcpy(path_action + k * path_cap, pa.data() + k * pc, (size_t)r.path_n * 4);
            memcpy(path_len + k * path_cap, pl.data() + k * pc, (size_t)r.path_n * 4);
        }
        if (stats) {
            stats[k].nodes = (int64_t)r.nodes;
            stats[k].expanded = (int64_t)r.expanded;
            stats[k].children = (int64_t)r.expanded * 12;
            stats[k].levels = (int64_t)r.batches;
            stats[k].min_len = (int32_t)r.min_len;
            stats[k].seconds = ms * 1e-3;  // of the whole group launch
        }
    }
    return ACX_OK;
}

// greedy_search on the device-resident priority frontier (acx_greedy.h).  *handled = false when the persistent
// kernel ran out of one of its capacities: the caller then reruns the search on the batch-per-launch path.
template <typename W>
static int run_greedy_device(const Pres<W>& root, int L, int64_t max_nodes, int cyclical, int32_t* solved, int32_t* path_action, int32_t* path_len,
                             int64_t path_cap, int64_t* path_n, acx_search_stats* stats, bool* handled) {
    *handled = false;
    GreedySearch<W> G;
    int rc = G.setup(root, L, max_nodes, cyclical, nullptr);
    if (rc) return rc;
    Searcher<W>& S = G.S;
    GreedyDev<W>& g = G.g;
    hipStream_t st = S.st;
    DevBuf outb;
    if (outb.alloc(sizeof(GreedyOut))) return ACX_E_NOMEM;
    ACX_HIP_TRY(hipMemsetAsync(outb.p, 0, sizeof(GreedyOut), st));
    hipEvent_t ev0, ev1;
    ACX_HIP_TRY(hipEventCreate(&ev0));
    ACX_HIP_TRY(hipEventCreate(&ev1));
    ACX_HIP_TRY(hipEventRecord(ev0, st));
    hipLaunchKernelGGL(k_greedy_persistent<W>, dim3(1), dim3(kGT), 0, st, g, (GreedyOut*)outb.p);
    ACX_HIP_TRY(hipGetLastError());
    ACX_HIP_TRY(hipEventRecord(ev1, st));
    GreedyOut o;
    ACX_HIP_TRY(hipMemcpyAsync(&o, outb.p, sizeof(o), hipMemcpyDeviceToHost, st));
    ACX_HIP_TRY(hipStreamSynchronize(st));
    float ms = 0;
    ACX_HIP_TRY(hipEventElapsedTime(&ms, ev0, ev1));
    (void)hipEventDestroy(ev0);
    (void)hipEventDestroy(ev1);
    if (getenv("ACX_DEBUG"))
        fprintf(stderr, "[acx_greedy] status=%u nodes=%u batches=%llu expanded=%llu sorts=%llu big_sorts=%llu max_bucket=%u reason=%u %.3f ms\n", o.status, o.nodes,
                o.batches, o.expanded, o.sorts, o.big_sorts, o.max_bucket, o.fallback_reason, ms);
    if (getenv("ACX_DEBUG")) {
        fprintf(stderr, "[acx_greedy] sorts by log2(n):");
        for (int k = 0; k < 16; k++) fprintf(stderr, " %u", o.hist_sort[k]);
        fprintf(stderr, "\n[acx_greedy] batches by log2(parents):");
        for (int k = 0; k < 10; k++) fprintf(stderr, " %u", o.hist_np[k]);
        fprintf(stderr, "\n");
        unsigned long long tot = 0;
        for (int k = 0; k < 8; k++) tot += o.t_phase[k];
        if (tot) fprintf(stderr, "[acx_greedy] cycles%%: select %.1f sort %.1f expand %.1f probe %.1f scan %.1f commit %.1f file %.1f tail %.1f (total %.3e cycles)\n",
                100.0 * o.t_phase[0] / tot, 100.0 * o.t_phase[1] / tot, 100.0 * o.t_phase[2] / tot, 100.0 * o.t_phase[3] / tot, 100.0 * o.t_phase[4] / tot,
                100.0 * o.t_phase[5] / tot, 100.0 * o.t_phase[6] / tot, 100.0 * o.t_phase[7] / tot, (double)tot);
    }
    if (o.status == GREEDY_FALLBACK) return ACX_OK;  // *handled stays false
    *handled = true;
    if (o.status == GREEDY_MOVE_ERROR) return err_to_rc(o.err);
    if (o.status != GREEDY_SOLVED && o.status != GREEDY_BUDGET && o.status != GREEDY_EXHAUSTED)
        return fail(ACX_E_NODEVICE, "greedy frontier kernel ended in state %u", o.status);
    *solved = o.status == GREEDY_SOLVED ? 1 : 0;
    // greedy.py:93 (success) / :121 (failure): path of a popped node + one more (action, length) entry
    const uint32_t tail_node = *solved ? o.solved_parent : o.last_parent;
    uint32_t par, dep;
    rc = S.node_field(tail_node, par, dep);
    if (rc) return rc;
    int64_t n = 0;
    rc = S.path_of(tail_node, dep, path_action, path_len, path_cap, &n);
    if (rc) return rc;
    if (n < path_cap) {
        path_action[n] = *solved ? (int32_t)o.solved_action : 11;
        path_len[n] = *solved ? 2 : (int32_t)o.last_child_len;
    }
    *path_n = n + 1;
    if (stats) {
        stats->nodes = (int64_t)o.nodes;
        stats->expanded = (int64_t)o.expanded;
        stats->children = (int64_t)o.expanded * 12;
        stats->levels = (int64_t)o.batches;
        stats->min_len = (int32_t)o.min_len;
        stats->seconds = ms * 1e-3;
    }
    if (*path_n > path_cap) return fail(ACX_E_CAPACITY, "path has %lld entries, buffer holds %lld", (long long)*path_n, (long long)path_cap);
    return ACX_OK;
}

template <typename W>
static int run_search(int kind, const int8_t* pres, int L, int64_t max_nodes, int cyclical, int32_t* solved, int32_t* path_action,
                      int32_t* path_len, int64_t path_cap, int64_t* path_n, acx_search_stats* stats) {
    Pres<W> root;
    bool ok = pack_relator<W>(pres, L, root.w0, root.n0);
    ok = pack_relator<W>(pres + L, L, root.w1, root.n1) && ok;
    if (!ok) return fail(ACX_E_ROWERR, "acx_search: the presentation is not a zero-padded word pair over {+-1,+-2}");
    const bool greedy = kind == ACX_SEARCH_GREEDY;
    *solved = 0;
    *path_n = 0;
    if (greedy && !getenv("ACX_GREEDY_HOST")) {  // device-resident priority frontier; falls through when it hits a capacity
        bool handled = false;
        const int grc = run_greedy_device<W>(root, L, max_nodes, cyclical, solved, path_action, path_len, path_cap, path_n, stats, &handled);
        if (grc != ACX_OK || handled) return grc;
        *solved = 0;
        *path_n = 0;
    }
    // parents per batch: a BFS batch far larger than the remaining budget only inflates the candidate arena and the table
    // (every new key of a batch claims an entry, committed or not), so it is tied to the budget
    const uint32_t bmax = greedy ? (uint32_t)std::min<int64_t>(std::max<int64_t>(max_nodes, 1024), 1 << 14)
                                 : (uint32_t)std::min<int64_t>(std::max<int64_t>(max_nodes / 4, 1024), 1 << 20);
    Searcher<W> S;
    int rc = S.init(L, cyclical, max_nodes, bmax, greedy, !greedy);
    if (rc) return rc;
    SearchDev<W>& d = S.d;
    hipStream_t st = S.st;
    hipEvent_t ev0, ev1;
    ACX_HIP_TRY(hipEventCreate(&ev0));
    ACX_HIP_TRY(hipEventCreate(&ev1));
    ACX_HIP_TRY(hipEventRecord(ev0, st));

    const uint32_t tl0 = (uint32_t)(root.n0 + root.n1);
    if (greedy) hipLaunchKernelGGL(k_root<W>, dim3(1), dim3(1), 0, st, d, keyops<W>::make(root.w0, root.n0), keyops<W>::make(root.w1, root.n1), tl0);
    else hipLaunchKernelGGL(k_root_tab<W>, dim3(1), dim3(1), 0, st, d, keyops<W>::make(root.w0, root.n0), keyops<W>::make(root.w1, root.n1), tl0);
    uint64_t nodes = 1, expanded = 0, batches = 0;
    uint32_t min_len = tl0;
    *solved = 0;
    *path_n = 0;
    uint32_t last_parent = 0, last_child_len = 0;  // greedy.py:121 return value
    bool done = false;

    // greedy: heap buckets keyed by (total length, depth) -> node ids (host side; sorted on the device when popped)
    struct Bucket {
        std::vector<uint32_t> ids;
        size_t head = 0;  // ids[head..] are still queued
    };
    std::map<std::pair<uint32_t, uint32_t>, Bucket> buckets;
    std::pair<uint32_t, uint32_t> sorted_key(0, 0);
    bool have_sorted = false;
    if (greedy) buckets[{tl0, 0u}].ids.push_back(0);
    uint32_t bfs_head = 0;  // next FIFO position to expand
    uint32_t* dlist = (uint32_t*)S.arena_list.p;
    std::vector<uint32_t> hlist;
    std::vector<uint8_t> hlen;
    const bool debug = getenv("ACX_DEBUG") != nullptr;
    uint32_t adaptive = 64;  // greedy batch size: grows while buckets are consumed without a cut

    while (!done) {
        // ---- choose the batch of parents -------------------------------------------------------------
        uint32_t np = 0;
        const uint32_t* plist = nullptr;
        uint32_t pbegin = 0;
        int bucket_len = -1;
        uint32_t bucket_depth = 0;
        if (!greedy) {
            if (bfs_head >= nodes) break;  // queue exhausted (breadth_first.py:61)
            np = (uint32_t)std::min<uint64_t>(nodes - bfs_head, bmax);
            pbegin = bfs_head;
        } else {
            if (buckets.empty()) break;  // heap exhausted (greedy.py:71)
            auto it = buckets.begin();
            Bucket& bk = it->second;
            const size_t live = bk.ids.size() - bk.head;
            uint32_t* ids = bk.ids.data() + bk.head;
            bucket_len = (int)it->first.first;
            bucket_depth = it->first.second;
            if (!(have_sorted && sorted_key == it->first) && live > 1) {
                // order the bucket by signed state tuple on the device (merge sort on node ids)
                DevBuf big;
                uint32_t* buf = dlist;
                const size_t list_cap = std::max<uint32_t>(bmax, 1024);
                if (live > list_cap) {
                    if (big.alloc(live * 8)) return ACX_E_NOMEM;
                    buf = (uint32_t*)big.p;
                }
                uint32_t* out = buf + std::max(live, list_cap);
                if (live > list_cap) out = buf + live;
                ACX_HIP_TRY(hipMemcpyAsync(buf, ids, live * 4, hipMemcpyHostToDevice, st));
                hipLaunchKernelGGL(k_rank_sort<W>, dim3((unsigned)((live + 255) / 256)), dim3(256), 0, st, d, buf, (uint32_t)live, out);
                ACX_HIP_TRY(hipGetLastError());
                ACX_HIP_TRY(hipMemcpyAsync(ids, out, live * 4, hipMemcpyDeviceToHost, st));
                ACX_HIP_TRY(hipStreamSynchronize(st));
            }
            have_sorted = true;
            sorted_key = it->first;
            np = (uint32_t)std::min<size_t>(live, std::min<uint32_t>(adaptive, bmax));
            hlist.assign(ids, ids + np);
            ACX_HIP_TRY(hipMemcpyAsync(dlist, hlist.data(), (size_t)np * 4, hipMemcpyHostToDevice, st));
            plist = dlist;
        }
        const uint32_t m = 12u * np;
        const dim3 grid((m + 255) / 256), block(256);
        batches++;
        if (debug) fprintf(stderr, "[acx_search] batch %llu: np=%u nodes=%llu bucket=(%d,%u) buckets=%zu\n", (unsigned long long)batches, np,
                           (unsigned long long)nodes, bucket_len, bucket_depth, buckets.size());

        // ---- expand, dedup with min-tag resolution, number the winners, decide -- all on the stream --------
        rc = S.reset_batch_scalars();
        if (rc) return rc;
        hipLaunchKernelGGL(k_expand<W>, grid, block, 0, st, d, plist, pbegin, np);
        if (greedy) {
            // batch-local table sized for this batch (only its used prefix is cleared)
            uint32_t bs = 1024;
            while (bs < 2 * m) bs <<= 1;
            hipLaunchKernelGGL(k_lookup<W>, grid, block, 0, st, d, m);
            ACX_HIP_TRY(hipMemsetAsync(d.bslots, 0xff, (size_t)bs * 4, st));
            hipLaunchKernelGGL(k_insert<W>, grid, block, 0, st, d, d.bslots, bs - 1, m, 1);
            hipLaunchKernelGGL(k_mark<W>, grid, block, 0, st, d, d.bslots, m, bucket_len);
        } else {
            hipLaunchKernelGGL(k_insert_tab<W>, grid, block, 0, st, d, m, (uint32_t)batches, 1);  // epoch = batch number (>= 1)
            hipLaunchKernelGGL(k_mark_tab<W>, grid, block, 0, st, d, m, (uint32_t)batches);
        }
        {
            size_t tb = S.tmp_bytes;
            if (rocprim::exclusive_scan(S.arena_tmp.p, tb, d.cflag, d.cpos, 0u, m, rocprim::plus<uint32_t>(), st) != hipSuccess)
                return fail(ACX_E_NODEVICE, "rocprim::exclusive_scan failed");
        }
        hipLaunchKernelGGL(k_decide<W>, dim3(1), dim3(1), 0, st, d, m, np, (unsigned long long)nodes, (long long)max_nodes, greedy ? 1 : 0, S.d_dec);
        hipLaunchKernelGGL(k_commit<W>, grid, block, 0, st, d, plist, pbegin, S.d_dec, m, (uint32_t)nodes, greedy ? 1 : 0);
        ACX_HIP_TRY(hipGetLastError());
        // one read-back: the decision and (greedy) the total lengths of the nodes this batch may have created
        Decision* dec = (Decision*)S.h_pin;
        uint8_t* hl = S.h_pin + sizeof(Decision) + 64 - (sizeof(Decision) % 64);
        ACX_HIP_TRY(hipMemcpyAsync(dec, S.d_dec, sizeof(Decision), hipMemcpyDeviceToHost, st));
        if (greedy) ACX_HIP_TRY(hipMemcpyAsync(hl, d.tlen + nodes, std::min<uint64_t>(m, S.cap_nodes - nodes), hipMemcpyDeviceToHost, st));
        ACX_HIP_TRY(hipStreamSynchronize(st));
        if (dec->err) return err_to_rc(dec->err);
        if (debug) fprintf(stderr, "[acx_search]   total=%u p_end=%u committed=%u solved=%u budget_hit=%u\n", dec->total, dec->p_end, dec->committed, dec->solved, dec->budget_hit);
        const uint32_t p_end = dec->p_end, committed = dec->committed;
        min_len = std::min<uint32_t>(min_len, dec->min_len);

        if (dec->solved) {
            // success: path of the parent + (action, 2); checked before dedup and before the budget test
            const uint32_t ps = dec->solved_tag / 12, as = dec->solved_tag % 12;
            const uint32_t pid = plist ? hlist[ps] : pbegin + ps;
            uint32_t par, dep;
            rc = S.node_field(pid, par, dep);
            if (rc) return rc;
            int64_t n = 0;
            rc = S.path_of(pid, dep, path_action, path_len, path_cap, &n);
            if (rc) return rc;
            if (n < path_cap) {
                path_action[n] = (int32_t)as;
                path_len[n] = 2;
            }
            *path_n = n + 1;
            *solved = 1;
            expanded += ps + 1;
            nodes += committed;  // nodes inserted before the solving child (stats only)
            min_len = 2;
            done = true;
            break;
        }
        expanded += p_end + 1;

        if (greedy) {
            // file the new nodes into their heap buckets and drop the popped parents
            last_parent = hlist[p_end];
            last_child_len = dec->last_child_len;
            auto it = buckets.begin();
            it->second.head += p_end + 1;
            const bool cut = p_end + 1 < np;
            if (it->second.head == it->second.ids.size()) {
                buckets.erase(it);
                have_sorted = false;
            }
            for (uint32_t k = 0; k < committed; k++) {
                const std::pair<uint32_t, uint32_t> key(hl[k], bucket_depth + 1);
                if (have_sorted && key == sorted_key) have_sorted = false;  // cannot happen (depth differs); kept for safety
                buckets[key].ids.push_back((uint32_t)nodes + k);
            }
            adaptive = cut ? 64 : std::min<uint32_t>(adaptive * 2, bmax);
        } else {
            bfs_head += p_end + 1;
        }
        nodes += committed;
        if (dec->budget_hit) break;  // breadth_first.py:91-95 / greedy.py:115-119
    }

    if (!*solved && greedy) {  // greedy.py:121: path of the last popped node + (11, length of its last child)
        uint32_t par, dep;
        rc = S.node_field(last_parent, par, dep);
        if (rc) return rc;
        int64_t n = 0;
        rc = S.path_of(last_parent, dep, path_action, path_len, path_cap, &n);
        if (rc) return rc;
        if (n < path_cap) {
            path_action[n] = 11;
            path_len[n] = (int32_t)last_child_len;
        }
        *path_n = n + 1;
    }
    ACX_HIP_TRY(hipEventRecord(ev1, st));
    ACX_HIP_TRY(hipEventSynchronize(ev1));
    float ms = 0;
    ACX_HIP_TRY(hipEventElapsedTime(&ms, ev0, ev1));
    (void)hipEventDestroy(ev0);
    (void)hipEventDestroy(ev1);
    if (stats) {
        stats->nodes = (int64_t)nodes;
        stats->expanded = (int64_t)expanded;
        stats->children = (int64_t)expanded * 12;
        stats->levels = (int64_t)batches;
        stats->min_len = (int32_t)min_len;
        stats->seconds = ms * 1e-3;
    }
    if (*path_n > path_cap) return fail(ACX_E_CAPACITY, "path has %lld entries, buffer holds %lld", (long long)*path_n, (long long)path_cap);
    return ACX_OK;
}


// =============================================================================================
// Sharded frontier: one engine per GPU, states partitioned by hash(key) mod world.  The host side
// (ac_solver/search/sharded.py) moves candidate records between ranks with an RCCL all-to-all and
// broadcasts winner tags; everything per rank happens in the kernels below.  A record is KW+2 int64:
// the key words, tag = 12 * global_parent_position + action, parent_ref = rank << 40 | local id.
// =============================================================================================
template <typename W> struct recio;
template <> struct recio<uint64_t> {
    static constexpr int KW = 2;
    static ACX_HD void put(int64_t* r, uint64_t k0, uint64_t k1) { r[0] = (int64_t)k0; r[1] = (int64_t)k1; }
    static ACX_HD void get(const int64_t* r, uint64_t& k0, uint64_t& k1) { k0 = (uint64_t)r[0]; k1 = (uint64_t)r[1]; }
};
template <> struct recio<u128> {
    static constexpr int KW = 4;
    static ACX_HD void put(int64_t* r, u128 k0, u128 k1) {
        r[0] = (int64_t)(uint64_t)k0; r[1] = (int64_t)(uint64_t)(k0 >> 64);
        r[2] = (int64_t)(uint64_t)k1; r[3] = (int64_t)(uint64_t)(k1 >> 64);
    }
    static ACX_HD void get(const int64_t* r, u128& k0, u128& k1) {
        k0 = ((u128)(uint64_t)r[1] << 64) | (uint64_t)r[0];
        k1 = ((u128)(uint64_t)r[3] << 64) | (uint64_t)r[2];
    }
};

template <typename W>
__global__ void __launch_bounds__(256) k_shard_expand(SearchDev<W> d, const int64_t* __restrict__ ids, const int64_t* __restrict__ gpos, int64_t np,
                                                      int64_t pref_hi, int64_t* __restrict__ rec, unsigned long long* __restrict__ solved) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 12 * np) return;
    const int64_t p = t / 12;
    const int a = (int)(t - 12 * p);
    const int64_t id = ids[p];
    Pres<W> s;
    key_to_pres<W>(d.k0[id], d.k1[id], s);
    const int e = apply_move<W, true>(s, a, d.L, d.cyclical != 0);
    if (e) atomicOr(d.err, (uint32_t)e);
    int64_t* r = rec + t * (recio<W>::KW + 2);
    recio<W>::put(r, keyops<W>::make(s.w0, s.n0), keyops<W>::make(s.w1, s.n1));
    const int64_t tag = 12 * gpos[p] + a;
    r[recio<W>::KW] = tag;
    r[recio<W>::KW + 1] = pref_hi | id;
    if (s.n0 + s.n1 == 2) atomicMin(solved, (unsigned long long)tag);
    if ((uint32_t)(s.n0 + s.n1) < *(volatile uint32_t*)d.min_len) atomicMin(d.min_len, (uint32_t)(s.n0 + s.n1));
}

// Owner rank of a packed key: the arithmetic of ac_solver/search/sharded.py:owner_of on the key's int64 words.
ACX_HD uint64_t owner_mix(uint64_t h, uint64_t w) {
    h = (h ^ w) * 0x9E3779B97F4A7C15ull;
    return h ^ (h >> 29);
}
ACX_HD uint32_t owner_of_key(uint64_t k0, uint64_t k1, uint32_t world) {
    const uint64_t h = owner_mix(owner_mix(0, k0), k1);
    return (uint32_t)((h & 0x7FFFFFFFFFFFFFFFull) % world);
}
ACX_HD uint32_t owner_of_key(u128 k0, u128 k1, uint32_t world) {
    uint64_t h = owner_mix(owner_mix(0, (uint64_t)k0), (uint64_t)(k0 >> 64));
    h = owner_mix(owner_mix(h, (uint64_t)k1), (uint64_t)(k1 >> 64));
    return (uint32_t)((h & 0x7FFFFFFFFFFFFFFFull) % world);
}

// k_shard_expand + routing: the record of a child goes straight into the send region of the rank that owns the
// child's key (region o = rec[o * region_cap ...], filled through a wave-aggregated cursor counts[o]), so the
// all-to-all can leave without a sort by owner.  The order inside a region is arbitrary (the receiver orders by tag).
template <typename W>
__global__ void __launch_bounds__(1024) k_shard_expand_routed(SearchDev<W> d, const int64_t* __restrict__ ids, const int64_t* __restrict__ gpos, int64_t np,
                                                             int64_t pref_hi, uint32_t world, int64_t* __restrict__ rec, int64_t region_cap,
                                                             unsigned long long* __restrict__ counts, unsigned long long* __restrict__ solved) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = t < 12 * np;
    const uint32_t lane = threadIdx.x & 63;
    W k0 = 0, k1 = 0;
    int64_t tag = 0, pref = 0;
    uint32_t owner = 0xFFFFFFFFu;
    if (active) {
        const int64_t p = t / 12;
        const int a = (int)(t - 12 * p);
        const int64_t id = ids[p];
        Pres<W> s;
        const W pk0 = d.k0[id], pk1 = d.k1[id];
        key_to_pres<W>(pk0, pk1, s);
        const int e = apply_move<W, true>(s, a, d.L, d.cyclical != 0);
        if (e) atomicOr(d.err, (uint32_t)e);
        k0 = keyops<W>::make(s.w0, s.n0);
        k1 = keyops<W>::make(s.w1, s.n1);
        tag = 12 * gpos[p] + a;
        pref = pref_hi | id;
        // a move that leaves the state unchanged (over-long product: ac_moves.py:64, :126) yields the parent itself, which
        // is in the visited set already: such a child can never be new, so it is not sent at all
        if (k0 != pk0 || k1 != pk1) owner = owner_of_key(k0, k1, world);
        if (s.n0 + s.n1 == 2) atomicMin(solved, (unsigned long long)tag);
        if ((uint32_t)(s.n0 + s.n1) < *(volatile uint32_t*)d.min_len) atomicMin(d.min_len, (uint32_t)(s.n0 + s.n1));
    }
    // position inside the destination region: wave-aggregated LDS counters per owner, then ONE global atomicAdd per
    // (workgroup, owner) -- per-wave global atomics on `world` addresses serialise (1.6 ms per 12 M children)
    __shared__ uint32_t s_cnt[64];
    __shared__ unsigned long long s_base[64];
    if (threadIdx.x < 64) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    uint32_t pos_in_block = 0;
    for (uint32_t o = 0; o < world; o++) {
        const unsigned long long m = __ballot(owner == o);
        if (!m) continue;
        const uint32_t lead = (uint32_t)__builtin_ctzll(m);
        uint32_t base = 0;
        if (lane == lead) base = atomicAdd(&s_cnt[o], (uint32_t)__popcll(m));
        base = (uint32_t)__shfl((int)base, (int)lead);
        if (owner == o) pos_in_block = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    }
    __syncthreads();
    if (threadIdx.x < world && s_cnt[threadIdx.x]) s_base[threadIdx.x] = atomicAdd(&counts[threadIdx.x], (unsigned long long)s_cnt[threadIdx.x]);
    __syncthreads();
    if (owner != 0xFFFFFFFFu) {
        const int64_t pos = (int64_t)s_base[owner] + pos_in_block;
        if (pos < region_cap) {
            int64_t* r = rec + ((int64_t)owner * region_cap + pos) * (recio<W>::KW + 2);
            recio<W>::put(r, k0, k1);
            r[recio<W>::KW] = tag;
            r[recio<W>::KW + 1] = pref;
        }  // an overflow shows in counts[o] > region_cap; the host reports it
    }
}

template <typename W> __global__ void __launch_bounds__(256) k_shard_tags(const int64_t* __restrict__ rec, int64_t n, uint64_t* __restrict__ tags, uint32_t* __restrict__ idx) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    tags[i] = (uint64_t)rec[i * (recio<W>::KW + 2) + recio<W>::KW];
    idx[i] = (uint32_t)i;
}

// candidate arena in tag order: j-th smallest tag -> slot j
template <typename W>
__global__ void __launch_bounds__(256) k_shard_gather(SearchDev<W> d, const int64_t* __restrict__ rec, const uint64_t* __restrict__ tags_sorted,
                                                      const uint32_t* __restrict__ idx_sorted, int64_t n, int64_t* __restrict__ ctag, int64_t* __restrict__ cpref) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const int64_t* r = rec + (int64_t)idx_sorted[j] * (recio<W>::KW + 2);
    W k0, k1;
    recio<W>::get(r, k0, k1);
    d.ck0[j] = k0;
    d.ck1[j] = k1;
    d.clen[j] = (uint8_t)(keyops<W>::len(k0) + keyops<W>::len(k1));
    ctag[j] = (int64_t)tags_sorted[j];
    cpref[j] = r[recio<W>::KW + 1];
}

template <typename W>
__global__ void __launch_bounds__(256) k_shard_win_tags(SearchDev<W> d, const int64_t* __restrict__ ctag, int64_t n, int64_t* __restrict__ out) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n || !d.cflag[j]) return;
    out[d.cpos[j]] = ctag[j];
}

template <typename W>
__global__ void __launch_bounds__(256) k_shard_commit(SearchDev<W> d, const int64_t* __restrict__ ctag, const int64_t* __restrict__ cpref, int64_t n,
                                                      int64_t cutoff, uint32_t base, int64_t* __restrict__ node_pref) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n || !d.cflag[j] || ctag[j] >= cutoff) return;
    const uint32_t id = base + d.cpos[j];
    d.k0[id] = d.ck0[j];
    d.k1[id] = d.ck1[j];
    d.act[id] = (uint8_t)(ctag[j] % 12);
    d.tlen[id] = d.clen[j];
    node_pref[id] = cpref[j];
}

// number of winners with tag < cutoff: candidates are in tag order, so it is cpos at the first tag >= cutoff
template <typename W> __global__ void k_shard_count(SearchDev<W> d, const int64_t* __restrict__ ctag, int64_t n, int64_t cutoff, uint32_t* __restrict__ count) {
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (ctag[mid] < cutoff) lo = mid + 1;
        else hi = mid;
    }
    *count = lo >= n ? d.cpos[n - 1] + d.cflag[n - 1] : d.cpos[lo];
}

template <typename W> struct ShardEngine {
    SearchDev<W> d;
    DevBuf nodes_buf, cand_buf, tab_buf, scal_buf, tmp_buf, sort_buf;
    int64_t* node_pref = nullptr;  // [cap] parent_ref of every local node
    int64_t* ctag = nullptr;
    int64_t* cpref = nullptr;
    uint64_t* tags_in = nullptr;
    uint64_t* tags_sorted = nullptr;
    uint32_t* idx_in = nullptr;
    uint32_t* idx_sorted = nullptr;
    uint32_t* commit_count = nullptr;
    size_t scan_tmp = 0, sort_tmp = 0;
    uint64_t cap_nodes = 0, cap_cand = 0, n_slots = 0;
    uint64_t nodes = 0;    // committed local nodes
    uint32_t epoch = 0;    // insert calls so far (stamps of the inline-key table)
    int64_t pending = 0;   // candidates of the last insert (awaiting commit)
    int rank = 0, world = 1;

    int init(int L, int cyclical, int64_t node_cap, int64_t batch_cap, int rank_, int world_) {
        memset(&d, 0, sizeof(d));
        d.L = L;
        d.cyclical = cyclical;
        rank = rank_;
        world = world_;
        cap_nodes = (uint64_t)node_cap + 64;
        cap_cand = (uint64_t)std::max<int64_t>(batch_cap, 1024);
        n_slots = 1024;
        while (n_slots < 2 * (cap_nodes + cap_cand)) n_slots <<= 1;
        if (n_slots > (1ull << 31)) return fail(ACX_E_INVAL, "acx_shard: capacity too large for 32-bit node ids");
        size_t o = 0;
        auto take = [&](uint8_t* base, size_t bytes) {
            uint8_t* p = base ? base + o : nullptr;
            o += (bytes + 255) / 256 * 256;
            return p;
        };
        for (int pass = 0; pass < 2; pass++) {
            uint8_t* b = (uint8_t*)nodes_buf.p;
            o = 0;
            d.k0 = (W*)take(b, cap_nodes * sizeof(W));
            d.k1 = (W*)take(b, cap_nodes * sizeof(W));
            node_pref = (int64_t*)take(b, cap_nodes * 8);
            d.act = (uint8_t*)take(b, cap_nodes);
            d.tlen = (uint8_t*)take(b, cap_nodes);
            if (pass == 0 && nodes_buf.alloc(o)) return ACX_E_NOMEM;
        }
        for (int pass = 0; pass < 2; pass++) {
            uint8_t* b = (uint8_t*)cand_buf.p;
            o = 0;
            d.ck0 = (W*)take(b, cap_cand * sizeof(W));
            d.ck1 = (W*)take(b, cap_cand * sizeof(W));
            ctag = (int64_t*)take(b, cap_cand * 8);
            cpref = (int64_t*)take(b, cap_cand * 8);
            tags_in = (uint64_t*)take(b, cap_cand * 8);
            tags_sorted = (uint64_t*)take(b, cap_cand * 8);
            idx_in = (uint32_t*)take(b, cap_cand * 4);
            idx_sorted = (uint32_t*)take(b, cap_cand * 4);
            d.cslot = (uint32_t*)take(b, cap_cand * 4);
            d.cflag = (uint32_t*)take(b, cap_cand * 4);
            d.cpos = (uint32_t*)take(b, cap_cand * 4);
            d.clen = (uint8_t*)take(b, cap_cand);
            if (pass == 0 && cand_buf.alloc(o)) return ACX_E_NOMEM;
        }
        if (tab_buf.alloc(n_slots * sizeof(TabEntry<W>))) return ACX_E_NOMEM;
        d.tab = (TabEntry<W>*)tab_buf.p;
        d.tmask = (uint32_t)(n_slots - 1);
        if (scal_buf.alloc(256)) return ACX_E_NOMEM;
        uint8_t* sc = (uint8_t*)scal_buf.p;
        d.err = (uint32_t*)(sc + 24);
        d.min_len = (uint32_t*)(sc + 28);
        commit_count = (uint32_t*)(sc + 32);
        if (rocprim::exclusive_scan(nullptr, scan_tmp, d.cflag, d.cpos, 0u, cap_cand, rocprim::plus<uint32_t>(), (hipStream_t) nullptr) != hipSuccess)
            return fail(ACX_E_NODEVICE, "rocprim::exclusive_scan sizing failed");
        if (rocprim::radix_sort_pairs(nullptr, sort_tmp, tags_in, tags_sorted, idx_in, idx_sorted, cap_cand, 0, 64, (hipStream_t) nullptr) != hipSuccess)
            return fail(ACX_E_NODEVICE, "rocprim::radix_sort_pairs sizing failed");
        if (tmp_buf.alloc(std::max(scan_tmp, sort_tmp) + 256)) return ACX_E_NOMEM;
        ACX_HIP_TRY(hipMemset(d.tab, 0xff, n_slots * sizeof(TabEntry<W>)));
        ACX_HIP_TRY(hipMemset(scal_buf.p, 0xff, 256));
        ACX_HIP_TRY(hipMemset(d.err, 0, 4));
        return ACX_OK;
    }
};

struct ShardAny {
    bool wide;
    ShardEngine<uint64_t>* e64 = nullptr;
    ShardEngine<u128>* e128 = nullptr;
};

#define ACX_SHARD_DISPATCH(h, ...)                   \
    do {                                             \
        if ((h)->wide) {                             \
            typedef u128 W;                          \
            auto& E = *(h)->e128;                    \
            (void)sizeof(W);                         \
            __VA_ARGS__;                             \
        } else {                                     \
            typedef uint64_t W;                      \
            auto& E = *(h)->e64;                     \
            (void)sizeof(W);                         \
            __VA_ARGS__;                             \
        }                                            \
    } while (0)

template <typename W> static int shard_root(ShardEngine<W>& E, const int8_t* pres, int64_t* rec) {
    Pres<W> root;
    bool ok = pack_relator<W>(pres, E.d.L, root.w0, root.n0);
    ok = pack_relator<W>(pres + E.d.L, E.d.L, root.w1, root.n1) && ok;
    if (!ok) return fail(ACX_E_ROWERR, "acx_shard: the presentation is not a zero-padded word pair over {+-1,+-2}");
    recio<W>::put(rec, keyops<W>::make(root.w0, root.n0), keyops<W>::make(root.w1, root.n1));
    rec[recio<W>::KW] = 0;
    rec[recio<W>::KW + 1] = -1;
    return ACX_OK;
}

template <typename W> static int shard_expand(ShardEngine<W>& E, const int64_t* ids, const int64_t* gpos, int64_t np, int64_t* rec, int64_t* solved, hipStream_t st) {
    if (np <= 0) return ACX_OK;
    const int64_t m = 12 * np;
    hipLaunchKernelGGL(k_shard_expand<W>, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, E.d, ids, gpos, np, (int64_t)E.rank << 40, rec,
                       (unsigned long long*)solved);
    ACX_HIP_TRY(hipGetLastError());
    return ACX_OK;
}

template <typename W>
static int shard_expand_routed(ShardEngine<W>& E, const int64_t* ids, const int64_t* gpos, int64_t np, int64_t* rec, int64_t region_cap, int64_t* counts,
                               int64_t* solved, hipStream_t st) {
    ACX_HIP_TRY(hipMemsetAsync(counts, 0, (size_t)E.world * 8, st));
    if (np <= 0) return ACX_OK;
    const int64_t m = 12 * np;
    if (E.world > 64) return fail(ACX_E_INVAL, "acx_shard_expand_routed handles world <= 64");
    hipLaunchKernelGGL(k_shard_expand_routed<W>, dim3((unsigned)((m + 1023) / 1024)), dim3(1024), 0, st, E.d, ids, gpos, np, (int64_t)E.rank << 40, (uint32_t)E.world,
                       rec, region_cap, (unsigned long long*)counts, (unsigned long long*)solved);
    ACX_HIP_TRY(hipGetLastError());
    return ACX_OK;
}

template <typename W> static int shard_insert(ShardEngine<W>& E, const int64_t* rec, int64_t n, int tag_bits, int64_t* win_tags, int64_t* n_win, hipStream_t st) {
    *n_win = 0;
    E.pending = n;
    if (n <= 0) return ACX_OK;
    if ((uint64_t)n > E.cap_cand) return fail(ACX_E_CAPACITY, "acx_shard_insert: %lld records exceed the batch capacity %llu", (long long)n, (unsigned long long)E.cap_cand);
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    hipLaunchKernelGGL(k_shard_tags<W>, grid, block, 0, st, rec, n, E.tags_in, E.idx_in);
    size_t tb = E.sort_tmp;
    const unsigned end_bit = (unsigned)(tag_bits < 1 ? 64 : (tag_bits > 64 ? 64 : tag_bits));  // tags < 2^tag_bits: fewer radix passes
    if (rocprim::radix_sort_pairs(E.tmp_buf.p, tb, E.tags_in, E.tags_sorted, E.idx_in, E.idx_sorted, (size_t)n, 0, end_bit, st) != hipSuccess)
        return fail(ACX_E_NODEVICE, "rocprim::radix_sort_pairs failed");
    hipLaunchKernelGGL(k_shard_gather<W>, grid, block, 0, st, E.d, rec, E.tags_sorted, E.idx_sorted, n, E.ctag, E.cpref);
    E.epoch++;
    hipLaunchKernelGGL(k_insert_tab<W>, grid, block, 0, st, E.d, (uint32_t)n, E.epoch, 0);
    hipLaunchKernelGGL(k_mark_tab<W>, grid, block, 0, st, E.d, (uint32_t)n, E.epoch);
    tb = E.scan_tmp;
    if (rocprim::exclusive_scan(E.tmp_buf.p, tb, E.d.cflag, E.d.cpos, 0u, (size_t)n, rocprim::plus<uint32_t>(), st) != hipSuccess)
        return fail(ACX_E_NODEVICE, "rocprim::exclusive_scan failed");
    hipLaunchKernelGGL(k_shard_win_tags<W>, grid, block, 0, st, E.d, E.ctag, n, win_tags);
    ACX_HIP_TRY(hipGetLastError());
    uint32_t last[2];
    ACX_HIP_TRY(hipMemcpyAsync(&last[0], E.d.cpos + (n - 1), 4, hipMemcpyDeviceToHost, st));
    ACX_HIP_TRY(hipMemcpyAsync(&last[1], E.d.cflag + (n - 1), 4, hipMemcpyDeviceToHost, st));
    ACX_HIP_TRY(hipStreamSynchronize(st));
    *n_win = (int64_t)last[0] + last[1];
    return ACX_OK;
}

template <typename W> static int shard_commit(ShardEngine<W>& E, int64_t cutoff, int64_t* first_id, int64_t* n_committed, hipStream_t st) {
    *first_id = (int64_t)E.nodes;
    *n_committed = 0;
    const int64_t n = E.pending;
    E.pending = 0;
    if (n <= 0) return ACX_OK;
    hipLaunchKernelGGL(k_shard_count<W>, dim3(1), dim3(1), 0, st, E.d, E.ctag, n, cutoff, E.commit_count);
    hipLaunchKernelGGL(k_shard_commit<W>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, E.d, E.ctag, E.cpref, n, cutoff, (uint32_t)E.nodes,
                       E.node_pref);
    ACX_HIP_TRY(hipGetLastError());
    uint32_t c = 0;
    ACX_HIP_TRY(hipMemcpyAsync(&c, E.commit_count, 4, hipMemcpyDeviceToHost, st));
    ACX_HIP_TRY(hipStreamSynchronize(st));
    if (E.nodes + c > E.cap_nodes) return fail(ACX_E_CAPACITY, "acx_shard_commit: node capacity exceeded");
    E.nodes += c;
    *n_committed = c;
    return ACX_OK;
}

template <typename W> static int shard_node_info(ShardEngine<W>& E, int64_t id, int64_t* info) {
    if (id < 0 || (uint64_t)id >= E.nodes) return fail(ACX_E_INVAL, "acx_shard_node_info: id out of range");
    uint8_t a = 0, l = 0;
    int64_t pr = 0;
    ACX_HIP_TRY(hipDeviceSynchronize());
    ACX_HIP_TRY(hipMemcpy(&a, E.d.act + id, 1, hipMemcpyDeviceToHost));
    ACX_HIP_TRY(hipMemcpy(&l, E.d.tlen + id, 1, hipMemcpyDeviceToHost));
    ACX_HIP_TRY(hipMemcpy(&pr, E.node_pref + id, 8, hipMemcpyDeviceToHost));
    info[0] = pr < 0 ? -1 : (int64_t)a;
    info[1] = l;
    info[2] = pr;
    return ACX_OK;
}

}  // namespace acx

using namespace acx;

extern "C" int acx_search(int kind, const int8_t* h_presentation, int L, int64_t max_nodes, int cyclical, int32_t* solved,
                          int32_t* path_action, int32_t* path_len, int64_t path_cap, int64_t* path_n, acx_search_stats* stats) {
    if (!have_device()) return ACX_E_NODEVICE;
    if ((kind != ACX_SEARCH_BFS && kind != ACX_SEARCH_GREEDY) || !h_presentation || L < 1 || !solved || !path_n || path_cap < 0 ||
        (path_cap > 0 && (!path_action || !path_len)))
        return fail(ACX_E_INVAL, "acx_search: bad argument");
    if (L > 61) return fail(ACX_E_INVAL, "acx_search handles max_relator_length <= 61, got %d", L);
    if (max_nodes < 0) max_nodes = 0;
    if (L <= 29) return run_search<uint64_t>(kind, h_presentation, L, max_nodes, cyclical, solved, path_action, path_len, path_cap, path_n, stats);
    return run_search<u128>(kind, h_presentation, L, max_nodes, cyclical, solved, path_action, path_len, path_cap, path_n, stats);
}

extern "C" int acx_release_cached_memory(void) {
    block_pool().trim();
    return ACX_OK;
}

// ------------------------------------------------------------------ many independent searches ----
#include <atomic>
#include <thread>

extern "C" int acx_search_many(int kind, const int8_t* h_presentations, int64_t n, int L, int64_t max_nodes, int cyclical, int n_threads,
                               int32_t* solved, int32_t* path_action, int32_t* path_len, int64_t path_cap, int64_t* path_n,
                               acx_search_stats* stats, int32_t* rc_out) {
    if (!have_device()) return ACX_E_NODEVICE;
    if (n < 0 || !h_presentations || !solved || !path_n || !rc_out || path_cap < 0) return fail(ACX_E_INVAL, "acx_search_many: bad argument");
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 64) n_threads = 64;
    if (kind == ACX_SEARCH_GREEDY && n > 1 && L >= 1 && L <= 61 && !getenv("ACX_GREEDY_HOST")) {
        // greedy: groups of searches in ONE launch of the persistent frontier kernel, one workgroup per search
        if (max_nodes < 0) max_nodes = 0;
        const double per_search = 160.0 * (double)std::max<int64_t>(max_nodes, 1) + 96e6;  // bytes, generous (run_greedy_group computes the exact figure)
        const int64_t group = (int64_t)std::max(1.0, std::min(256.0, 12e9 / per_search));
        std::vector<uint8_t> rerun((size_t)n, 0);
        for (int64_t k0 = 0; k0 < n; k0 += group) {
            const int64_t m = std::min<int64_t>(group, n - k0);
            int32_t* pa = path_action ? path_action + k0 * path_cap : nullptr;
            int32_t* pl = path_len ? path_len + k0 * path_cap : nullptr;
            acx_search_stats* ps = stats ? stats + k0 : nullptr;
            const int rc = L <= 29 ? run_greedy_group<uint64_t>(h_presentations + k0 * 2 * L, m, L, max_nodes, cyclical, solved + k0, pa, pl, path_cap, path_n + k0,
                                                                ps, rc_out + k0, rerun.data() + k0)
                                   : run_greedy_group<u128>(h_presentations + k0 * 2 * L, m, L, max_nodes, cyclical, solved + k0, pa, pl, path_cap, path_n + k0, ps,
                                                            rc_out + k0, rerun.data() + k0);
            if (rc != ACX_OK) return rc;
        }
        for (int64_t k = 0; k < n; k++)
            if (rerun[k])
                rc_out[k] = acx_search(kind, h_presentations + k * 2 * L, L, max_nodes, cyclical, solved + k, path_action ? path_action + k * path_cap : nullptr,
                                       path_len ? path_len + k * path_cap : nullptr, path_cap, path_n + k, stats ? stats + k : nullptr);
        block_pool().trim();
        for (int64_t k = 0; k < n; k++)
            if (rc_out[k] != ACX_OK && rc_out[k] != ACX_E_CAPACITY) return fail(ACX_E_ROWERR, "acx_search_many: search %lld failed with code %d", (long long)k, rc_out[k]);
        return ACX_OK;
    }
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::atomic<int64_t> next(0);
    auto work = [&]() {
        (void)hipSetDevice(dev);
        for (;;) {
            const int64_t k = next.fetch_add(1);
            if (k >= n) break;
            rc_out[k] = acx_search(kind, h_presentations + k * 2 * L, L, max_nodes, cyclical, solved + k, path_action ? path_action + k * path_cap : nullptr,
                                   path_len ? path_len + k * path_cap : nullptr, path_cap, path_n + k, stats ? stats + k : nullptr);
        }
        block_pool().trim();  // the blocks this thread cached go back before it ends
    };
    std::vector<std::thread> pool;
    for (int t = 0; t < n_threads; t++) pool.emplace_back(work);
    for (auto& t : pool) t.join();
    for (int64_t k = 0; k < n; k++)
        if (rc_out[k] != ACX_OK && rc_out[k] != ACX_E_CAPACITY) return fail(ACX_E_ROWERR, "acx_search_many: search %lld failed with code %d", (long long)k, rc_out[k]);
    return ACX_OK;
}

// ------------------------------------------------------------------ sharded frontier: C ABI ----
struct acx_shard {
    acx::ShardAny any;
};

extern "C" {

int acx_shard_key_words(int L) { return L <= 29 ? 2 : 4; }

acx_shard* acx_shard_create(int L, int cyclical, int64_t node_cap, int64_t batch_cap, int rank, int world) {
    if (!have_device()) return nullptr;
    if (L < 1 || L > 61 || node_cap < 1 || batch_cap < 1 || world < 1 || rank < 0 || rank >= world) {
        fail(ACX_E_INVAL, "acx_shard_create: bad argument (1 <= L <= 61)");
        return nullptr;
    }
    acx_shard* h = new (std::nothrow) acx_shard();
    if (!h) return nullptr;
    h->any.wide = L > 29;
    int rc;
    if (h->any.wide) {
        h->any.e128 = new ShardEngine<u128>();
        rc = h->any.e128->init(L, cyclical, node_cap, batch_cap, rank, world);
    } else {
        h->any.e64 = new ShardEngine<uint64_t>();
        rc = h->any.e64->init(L, cyclical, node_cap, batch_cap, rank, world);
    }
    if (rc != ACX_OK) {
        delete h->any.e64;
        delete h->any.e128;
        delete h;
        return nullptr;
    }
    return h;
}

void acx_shard_destroy(acx_shard* h) {
    if (!h) return;
    delete h->any.e64;
    delete h->any.e128;
    delete h;
}

int acx_shard_root_record(acx_shard* h, const int8_t* h_presentation, int64_t* h_record) {
    if (!h || !h_presentation || !h_record) return fail(ACX_E_INVAL, "acx_shard_root_record: bad argument");
    ACX_SHARD_DISPATCH(&h->any, return shard_root<W>(E, h_presentation, h_record));
}

int acx_shard_expand(acx_shard* h, const int64_t* d_ids, const int64_t* d_gpos, int64_t np, int64_t* d_records, int64_t* d_solved, void* stream) {
    if (!h || np < 0 || (np > 0 && (!d_ids || !d_gpos || !d_records || !d_solved))) return fail(ACX_E_INVAL, "acx_shard_expand: bad argument");
    ACX_SHARD_DISPATCH(&h->any, return shard_expand<W>(E, d_ids, d_gpos, np, d_records, d_solved, (hipStream_t)stream));
}

int acx_shard_expand_routed(acx_shard* h, const int64_t* d_ids, const int64_t* d_gpos, int64_t np, int64_t* d_records, int64_t region_cap,
                            int64_t* d_counts, int64_t* d_solved, void* stream) {
    if (!h || np < 0 || region_cap < 0 || !d_counts || (np > 0 && (!d_ids || !d_gpos || !d_records || !d_solved)))
        return fail(ACX_E_INVAL, "acx_shard_expand_routed: bad argument");
    ACX_SHARD_DISPATCH(&h->any, return shard_expand_routed<W>(E, d_ids, d_gpos, np, d_records, region_cap, d_counts, d_solved, (hipStream_t)stream));
}

int acx_shard_insert(acx_shard* h, const int64_t* d_records, int64_t n, int tag_bits, int64_t* d_win_tags, int64_t* n_win, void* stream) {
    if (!h || n < 0 || !n_win || (n > 0 && (!d_records || !d_win_tags))) return fail(ACX_E_INVAL, "acx_shard_insert: bad argument");
    ACX_SHARD_DISPATCH(&h->any, return shard_insert<W>(E, d_records, n, tag_bits, d_win_tags, n_win, (hipStream_t)stream));
}

int acx_shard_commit(acx_shard* h, int64_t cutoff_tag, int64_t* first_id, int64_t* n_committed, void* stream) {
    if (!h || !first_id || !n_committed) return fail(ACX_E_INVAL, "acx_shard_commit: bad argument");
    ACX_SHARD_DISPATCH(&h->any, return shard_commit<W>(E, cutoff_tag, first_id, n_committed, (hipStream_t)stream));
}

int acx_shard_node_info(acx_shard* h, int64_t id, int64_t* h_info3) {
    if (!h || !h_info3) return fail(ACX_E_INVAL, "acx_shard_node_info: bad argument");
    ACX_SHARD_DISPATCH(&h->any, return shard_node_info<W>(E, id, h_info3));
}

int64_t acx_shard_node_count(acx_shard* h) {
    if (!h) return 0;
    return h->any.wide ? (int64_t)h->any.e128->nodes : (int64_t)h->any.e64->nodes;
}

int acx_shard_status(acx_shard* h, int32_t* err, int32_t* min_len) {
    if (!h || !err || !min_len) return fail(ACX_E_INVAL, "acx_shard_status: bad argument");
    uint32_t v[2];
    ACX_HIP_TRY(hipDeviceSynchronize());
    ACX_SHARD_DISPATCH(&h->any, ACX_HIP_TRY(hipMemcpy(v, E.d.err, 8, hipMemcpyDeviceToHost)));
    *err = (int32_t)v[0];
    *min_len = (int32_t)v[1];
    return ACX_OK;
}

}  // extern "C"
