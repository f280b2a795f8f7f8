// acx_shard.hip -- per-GPU engine of the sharded BFS frontier (multi-GPU form of bfs, breadth_first.py:15-97) and its
// C ABI (acx_shard_*).  Orchestration across ranks: ac-solver_amd/ac_solver/search/sharded.py.
//
// Round 2: the engine keeps its frontier on the device (the nodes committed during a level ARE the next level, a
// contiguous id range in global FIFO order), dedups what it receives in an 8-byte stamp table (bucketed probing, one CAS
// per new state; the keys of occupants are read from the received records / the node arena) and turns winners into nodes
// through per-parent child masks -- the same 12-bit masks the ranks all-reduce -- so nothing is ever sorted and a chunk
// costs the host two read-backs (send counts, the decision scalars).
#include "acx_bfs.h"  // search_move: the shorter move code for searches whose root is in normal form

namespace acx {

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
    // the whole 32-byte record as two 16-byte stores (records are 32-byte aligned)
    static __device__ __forceinline__ void put_all(int64_t* r, uint64_t k0, uint64_t k1, int64_t tag, int64_t pref) {
        ((ulonglong2*)r)[0] = make_ulonglong2(k0, k1);
        ((ulonglong2*)r)[1] = make_ulonglong2((unsigned long long)tag, (unsigned long long)pref);
    }
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
    static __device__ __forceinline__ void put_all(int64_t* r, u128 k0, u128 k1, int64_t tag, int64_t pref) {  // 48 bytes, 16-byte aligned
        ((ulonglong2*)r)[0] = make_ulonglong2((uint64_t)k0, (uint64_t)(k0 >> 64));
        ((ulonglong2*)r)[1] = make_ulonglong2((uint64_t)k1, (uint64_t)(k1 >> 64));
        ((ulonglong2*)r)[2] = make_ulonglong2((unsigned long long)tag, (unsigned long long)pref);
    }
};

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


// ---- local frontier ------------------------------------------------------------------------------------------------------
// node arena: k0 / k1 (packed key), node_pref (parent_ref), act, tlen, gpos (d.depth: global FIFO position inside its level)
// The nodes [lvl_lo, lvl_hi) are the current level, ascending in gpos.

// first node of [lo, hi) whose gpos is >= c
__device__ __forceinline__ uint32_t lower_gpos(const uint32_t* __restrict__ gpos, uint32_t lo, uint32_t hi, uint32_t c) {
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (gpos[mid] < c) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

// [lo, hi) = the local frontier nodes with gpos in [c0, c1): found ONCE per chunk by one lane (two binary searches of ~25
// dependent loads each: done at the top of every workgroup of the expansion they cost it 8 of its 11 ms on a 1e8-node search)
template <typename W> __global__ void k_shard_bounds(SearchDev<W> d, uint32_t lvl_lo, uint32_t lvl_hi, uint32_t c0, uint32_t c1, uint32_t* __restrict__ out) {
    ACX_VGPR_PAD("v23");
    out[0] = lower_gpos(d.depth, lvl_lo, lvl_hi, c0);
    out[1] = lower_gpos(d.depth, lvl_lo, lvl_hi, c1);
}

// Children of the local frontier nodes with gpos in [c0, c1), each written straight into the send region of the rank
// that owns its key (region o = rec[o * region_cap ...]), so the all-to-all can leave without a sort by owner.  The order
// inside a region is arbitrary.  A workgroup expands kRouteItems x 1024 children and reserves its share of every region
// with ONE atomicAdd per owner: the cursors are single words, and one word takes only ~90 returning atomics per
// microsecond (a reservation per 1024 children cost 3.7 of the kernel's 8.3 ms on a 1e8-node search).
#ifndef ACX_ROUTE_ITEMS
#define ACX_ROUTE_ITEMS 4
#endif
constexpr int kRouteItems = ACX_ROUTE_ITEMS;

template <typename W, int MODE>
__global__ void __launch_bounds__(1024) k_shard_expand_routed(SearchDev<W> d, const uint32_t* __restrict__ bounds, int64_t pref_hi,
                                                             uint32_t world, int64_t* __restrict__ rec, int64_t region_cap, unsigned long long* __restrict__ counts,
                                                             unsigned long long* __restrict__ solved) {
    ACX_VGPR_PAD_W(W, "v71", "v103");
    __shared__ uint32_t s_cnt[64];
    __shared__ unsigned long long s_base[64];
    const uint32_t s_lo = bounds[0], s_hi = bounds[1];
    if (threadIdx.x < 64) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const int64_t np = (int64_t)s_hi - s_lo;
    const uint32_t lane = threadIdx.x & 63;
    W k0[kRouteItems], k1[kRouteItems];
    int64_t tag[kRouteItems];
    uint32_t owner[kRouteItems], pos_in_block[kRouteItems], ids[kRouteItems];
#pragma unroll
    for (int it = 0; it < kRouteItems; it++) {
        const int64_t t = ((int64_t)blockIdx.x * kRouteItems + it) * 1024 + threadIdx.x;
        owner[it] = 0xFFFFFFFFu;
        pos_in_block[it] = 0;
        k0[it] = k1[it] = 0;
        tag[it] = 0;
        ids[it] = 0;
        if (t < 12 * np) {
            const int64_t p = t / 12;
            const int a = (int)(t - 12 * p);
            const uint32_t id = s_lo + (uint32_t)p;
            Pres<W> s;
            const W pk0 = d.k0[id], pk1 = d.k1[id];
            key_to_pres<W>(pk0, pk1, s);
            const int e = search_move<W, MODE>(s, a, d.L, d.cyclical != 0);
            tag[it] = 12 * (int64_t)d.depth[id] + a;
            ids[it] = id;
            if (e) atomicMin(solved + 1, ((unsigned long long)tag[it] << 8) | (unsigned long long)e);  // first erroring move (global tag)
            k0[it] = keyops<W>::make(s.w0, s.n0);
            k1[it] = keyops<W>::make(s.w1, s.n1);
            // a move that leaves the state unchanged (over-long product: ac_moves.py:64, :126) yields the parent itself, which
            // is in the visited set already: such a child can never be new, so it is not sent at all
            if (k0[it] != pk0 || k1[it] != pk1) owner[it] = owner_of_key(k0[it], k1[it], world);
            if (s.n0 + s.n1 == 2) atomicMin(solved, (unsigned long long)tag[it]);
            if ((uint32_t)(s.n0 + s.n1) < *(volatile uint32_t*)d.min_len) atomicMin(d.min_len, (uint32_t)(s.n0 + s.n1));
        }
        // position inside the workgroup's share of the destination region: wave-aggregated LDS counters per owner
        for (uint32_t o = 0; o < world; o++) {
            const unsigned long long m = __ballot(owner[it] == o);
            if (!m) continue;
            const uint32_t lead = (uint32_t)__builtin_ctzll(m);
            uint32_t base = 0;
            if (lane == lead) base = atomicAdd(&s_cnt[o], (uint32_t)__popcll(m));
            base = (uint32_t)__shfl((int)base, (int)lead);
            if (owner[it] == o) pos_in_block[it] = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        }
    }
    __syncthreads();
    if (threadIdx.x < world && s_cnt[threadIdx.x]) s_base[threadIdx.x] = atomicAdd(&counts[threadIdx.x], (unsigned long long)s_cnt[threadIdx.x]);
    __syncthreads();
#pragma unroll
    for (int it = 0; it < kRouteItems; it++) {
        if (owner[it] == 0xFFFFFFFFu) continue;
        const int64_t pos = (int64_t)s_base[owner[it]] + pos_in_block[it];
        if (pos < region_cap)  // an overflow shows in counts[o] > region_cap; the host reports it
            recio<W>::put_all(rec + ((int64_t)owner[it] * region_cap + pos) * (recio<W>::KW + 2), k0[it], k1[it], tag[it], pref_hi | ids[it]);
    }
}

// ---- dedup of the received records -----------------------------------------------------------------------------------------
// Stamp table (cf. acx_bfs.h): 8-byte slots probed four at a time (one 32-byte sector), all ones = free.
//   fingerprint(27) | provisional(1) | payload(36)
// provisional: payload = index of a record of the running chunk (key and tag in rec[payload]); committed: payload = local
// node id (key in the node arena).  Among equal keys the smaller tag takes the slot with a CAS (retried when another
// record got there first).  Two byte flags per tag of the chunk, both zero on entry: btook[tag - tag0] is set by a record
// that takes a slot, brepl[tag - tag0] by the record that pushes it out again -- "took and was not replaced" does not
// depend on the order in which the two stores land.  k_shard_pack folds them into one 12-bit child mask per parent.
constexpr unsigned long long kShardFree = ~0ull;
constexpr unsigned long long kShardProv = 1ull << 36;
constexpr unsigned long long kShardPayload = kShardProv - 1;

template <typename W>
__global__ void __launch_bounds__(256) k_shard_insert(SearchDev<W> d, const int64_t* __restrict__ rec, int64_t n, int64_t tag0, uint32_t* __restrict__ cslot,
                                                      uint8_t* __restrict__ took_i) {
    ACX_VGPR_PAD("v63");
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t* r = rec + i * (recio<W>::KW + 2);
    W c0, c1;
    recio<W>::get(r, c0, c1);
    const int64_t tag = r[recio<W>::KW];
    const uint64_t hk = hash_key<W>(c0, c1);
    const unsigned long long fp = hk & ~((1ull << 37) - 1);
    const unsigned long long me = fp | kShardProv | (unsigned long long)i;
    uint32_t base = (uint32_t)hk & d.stmask & ~3u, probes = 0, took = 0;
    bool open = true;
    while (open) {
        const ulonglong2 lo = *(const ulonglong2*)(d.stab + base), hi = *(const ulonglong2*)(d.stab + base + 2);
        const unsigned long long v0 = lo.x, v1 = lo.y, v2 = hi.x, v3 = hi.y;
        auto hot = [&](unsigned long long v) { return v == kShardFree || (v >> 37) == (me >> 37); };
        uint32_t cand = (hot(v0) ? 1u : 0u) | (hot(v1) ? 2u : 0u) | (hot(v2) ? 4u : 0u) | (hot(v3) ? 8u : 0u);
        while (cand && open) {
            const uint32_t j = (uint32_t)__builtin_ctz(cand);
            cand &= cand - 1;
            unsigned long long st = j == 0 ? v0 : (j == 1 ? v1 : (j == 2 ? v2 : v3));
            unsigned long long* slot = d.stab + base + j;
            for (;;) {  // until this slot is decided for me (it changes only among records of MY key once it holds my key)
                if (st == kShardFree) {
                    const unsigned long long old = atomicCAS(slot, kShardFree, me);
                    if (old == kShardFree) {
                        took = 1;
                        open = false;
                        break;
                    }
                    st = old;
                }
                if ((st >> 37) != (me >> 37)) break;  // another key's fingerprint: next slot
                W q0, q1;
                int64_t qtag = -1;  // committed states beat every record
                if (st & kShardProv) {
                    const int64_t* h = rec + (int64_t)(st & kShardPayload) * (recio<W>::KW + 2);
                    recio<W>::get(h, q0, q1);
                    qtag = h[recio<W>::KW];
                } else {
                    const uint32_t id = (uint32_t)(st & kShardPayload);
                    q0 = d.k0[id];
                    q1 = d.k1[id];
                }
                if (q0 != c0 || q1 != c1) break;  // same fingerprint, other key: next slot
                if (qtag < 0 || qtag < tag) {     // seen before, or a record of this chunk with a smaller tag holds it
                    open = false;
                    break;
                }
                const unsigned long long old = atomicCAS(slot, st, me);  // push the larger tag out
                if (old == st) {
                    took = 1;
                    d.brepl[qtag - tag0] = 1;  // no longer the first discoverer
                    open = false;
                    break;
                }
                st = old;  // somebody else replaced it meanwhile: look again
            }
            if (took) cslot[i] = base + j;
        }
        base = (base + 4) & d.stmask;
        if (open && ++probes > d.stmask / 4) {
            atomicOr(d.err, kErrTableFull);
            open = false;
        }
    }
    if (took) d.btook[tag - tag0] = 1;
    took_i[i] = (uint8_t)took;  // the same flag by record index (coalesced): k_shard_commit skips the records that never took a slot
}

// one 12-bit mask per parent of the chunk: bit a set when child (parent, a) is a new state of this rank
template <typename W> __global__ void __launch_bounds__(256) k_shard_pack(SearchDev<W> d, int64_t n_parents, int32_t* __restrict__ lmask) {
    ACX_VGPR_PAD("v23");
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_parents) return;
    const uint32_t* t = (const uint32_t*)(d.btook + 12 * p);  // 12 bytes, 4-byte aligned
    const uint32_t* r = (const uint32_t*)(d.brepl + 12 * p);
    int32_t m = 0;
#pragma unroll
    for (int w = 0; w < 3; w++) {
        const uint32_t v = t[w] & ~r[w];  // bytes are 0 / 1
        m |= (int32_t)(((v & 1u) | ((v >> 7) & 2u) | ((v >> 14) & 4u) | ((v >> 21) & 8u)) << (4 * w));
    }
    lmask[p] = m;
}

// Winners with a tag below the cutoff become local nodes: id = base + (winners of this rank with a smaller tag), which the
// per-parent masks give without a sort; gpos likewise from the all-reduced masks.  Their slot is rewritten to the node id.
template <typename W>
__global__ void __launch_bounds__(256) k_shard_commit(SearchDev<W> d, const int64_t* __restrict__ rec, int64_t n, int64_t tag0, int64_t cutoff,
                                                      const int32_t* __restrict__ lmask, const int64_t* __restrict__ lprefix, const int32_t* __restrict__ gmask,
                                                      const int64_t* __restrict__ gprefix, uint32_t base, int64_t gpos_base, const uint32_t* __restrict__ cslot,
                                                      const uint8_t* __restrict__ took_i, int64_t* __restrict__ node_pref, uint32_t cap_nodes) {
    ACX_VGPR_PAD_W(W, "v39", "v47");
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !took_i[i]) return;
    const int64_t* r = rec + i * (recio<W>::KW + 2);
    const int64_t tag = r[recio<W>::KW];
    const int64_t rel = tag - tag0, par = rel / 12;
    if (tag >= cutoff || d.brepl[rel]) return;
    const uint32_t below = (1u << (uint32_t)(rel % 12)) - 1u;
    const uint32_t id = base + (uint32_t)lprefix[par] + (uint32_t)__popc((uint32_t)lmask[par] & below);
    if (id >= cap_nodes) return;  // the host has refused this commit already (capacity): never reached
    W k0, k1;
    recio<W>::get(r, k0, k1);
    d.k0[id] = k0;
    d.k1[id] = k1;
    d.act[id] = (uint8_t)(tag % 12);
    d.tlen[id] = (uint8_t)(keyops<W>::len(k0) + keyops<W>::len(k1));
    d.depth[id] = (uint32_t)(gpos_base + gprefix[par] + __popc((uint32_t)gmask[par] & below));
    node_pref[id] = r[recio<W>::KW + 1];
    const uint64_t hk = hash_key<W>(k0, k1);
    d.stab[cslot[i]] = (hk & ~((1ull << 37) - 1)) | (unsigned long long)id;
}

template <typename W> __global__ void k_shard_seed(SearchDev<W> d, W k0, W k1, int64_t* __restrict__ node_pref) {
    ACX_VGPR_PAD("v23");
    d.k0[0] = k0;
    d.k1[0] = k1;
    d.act[0] = 0xff;
    d.tlen[0] = (uint8_t)(keyops<W>::len(k0) + keyops<W>::len(k1));
    d.depth[0] = 0;
    node_pref[0] = -1;
    const uint64_t hk = hash_key<W>(k0, k1);
    d.stab[(uint32_t)hk & d.stmask & ~3u] = hk & ~((1ull << 37) - 1);  // committed stamp of node 0, first slot of its bucket
}

template <typename W> __global__ void k_shard_find(SearchDev<W> d, uint32_t lvl_lo, uint32_t lvl_hi, uint32_t gpos, int64_t* __restrict__ out) {
    ACX_VGPR_PAD("v23");
    const uint32_t k = lower_gpos(d.depth, lvl_lo, lvl_hi, gpos);
    *out = (k < lvl_hi && d.depth[k] == gpos) ? (int64_t)k : -1;
}

template <typename W> struct ShardEngine {
    SearchDev<W> d;
    DevBuf nodes_buf, cand_buf, tab_buf, scal_buf;
    int64_t* node_pref = nullptr;  // [cap] parent_ref of every local node
    uint32_t* cslot = nullptr;     // [cap_cand] slot a record took
    uint8_t* took_i = nullptr;     // [cap_cand] "took a slot", by record index
    int64_t* d_find = nullptr;
    uint32_t* d_bounds = nullptr;  // [2] frontier slice of the running chunk
    uint64_t cap_nodes = 0, cap_cand = 0, n_slots = 0, chunk_tags = 0;
    uint64_t nodes = 0;            // committed local nodes
    uint64_t lvl_lo = 0, lvl_hi = 0;
    int64_t pending = 0;           // records of the last insert (awaiting commit)
    const int64_t* pending_rec = nullptr;
    int64_t pending_tag0 = 0;
    int rank = 0, world = 1;
    int move_mode = kMoveGeneral;  // acx_bfs.h: set from the root (acx_shard_root_record, which every rank calls)

    int init(int L, int cyclical, int64_t node_cap, int64_t batch_cap, int64_t chunk_parents, int rank_, int world_) {
        memset(&d, 0, sizeof(d));
        d.L = L;
        d.cyclical = cyclical;
        rank = rank_;
        world = world_;
        cap_nodes = (uint64_t)node_cap + 64;
        cap_cand = (uint64_t)std::max<int64_t>(batch_cap, 1024);
        chunk_tags = 12ull * (uint64_t)std::max<int64_t>(chunk_parents, 1);
        n_slots = 1024;
        while (n_slots < 2 * (cap_nodes + cap_cand)) n_slots <<= 1;
        if (n_slots > (1ull << 31) || cap_nodes > (1ull << 31)) return fail(ACX_E_INVAL, "acx_shard: capacity too large for 32-bit node ids");
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
            d.depth = (uint32_t*)take(b, cap_nodes * 4);
            d.act = (uint8_t*)take(b, cap_nodes);
            d.tlen = (uint8_t*)take(b, cap_nodes);
            if (pass == 0 && nodes_buf.alloc(o)) return ACX_E_NOMEM;
        }
        for (int pass = 0; pass < 2; pass++) {
            uint8_t* b = (uint8_t*)cand_buf.p;
            o = 0;
            cslot = (uint32_t*)take(b, cap_cand * 4);
            took_i = take(b, cap_cand);
            d.btook = take(b, chunk_tags);  // one byte per tag of a chunk
            d.brepl = take(b, chunk_tags);
            if (pass == 0 && cand_buf.alloc(o)) return ACX_E_NOMEM;
        }
        if (tab_buf.alloc(n_slots * 8)) return ACX_E_NOMEM;
        d.stab = (unsigned long long*)tab_buf.p;
        d.stmask = (uint32_t)(n_slots - 1);
        if (scal_buf.alloc(256)) return ACX_E_NOMEM;
        uint8_t* sc = (uint8_t*)scal_buf.p;
        d.err = (uint32_t*)(sc + 24);
        d.min_len = (uint32_t*)(sc + 28);
        d_find = (int64_t*)(sc + 64);
        d_bounds = (uint32_t*)(sc + 96);
        ACX_HIP_TRY(hipMemset(d.stab, 0xff, n_slots * 8));
        ACX_HIP_TRY(hipMemset(scal_buf.p, 0xff, 256));
        ACX_HIP_TRY(hipMemset(d.err, 0, 4));
        ACX_HIP_TRY(hipDeviceSynchronize());  // the fills run on the null stream; the engine's calls arrive on the caller's (possibly non-blocking) stream
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
    // a root in normal form keeps the whole search in normal form (acx_bfs.h)
    E.move_mode = !is_normal_form<W>(root, E.d.cyclical != 0) ? kMoveGeneral : (E.d.cyclical ? kMoveNfCyclical : kMoveNf);
    recio<W>::put(rec, keyops<W>::make(root.w0, root.n0), keyops<W>::make(root.w1, root.n1));
    rec[recio<W>::KW] = 0;
    rec[recio<W>::KW + 1] = -1;
    return ACX_OK;
}

template <typename W> static int shard_seed(ShardEngine<W>& E, const int64_t* rec, hipStream_t st) {
    if (E.nodes) return fail(ACX_E_INVAL, "acx_shard_seed: the engine already holds nodes");
    if (rec) {
        W k0, k1;
        recio<W>::get(rec, k0, k1);
        hipLaunchKernelGGL(k_shard_seed<W>, dim3(1), dim3(1), 0, st, E.d, k0, k1, E.node_pref);
        ACX_HIP_TRY(hipGetLastError());
        E.nodes = 1;
    }
    return ACX_OK;
}

template <typename W>
static int shard_expand_routed(ShardEngine<W>& E, int64_t c0, int64_t c1, int64_t* rec, int64_t region_cap, int64_t* counts, int64_t* solved, hipStream_t st) {
    ACX_HIP_TRY(hipMemsetAsync(counts, 0, (size_t)E.world * 8, st));
    if (E.world > 64) return fail(ACX_E_INVAL, "acx_shard_expand_routed handles world <= 64");
    const int64_t np_max = std::min<int64_t>(c1 - c0, (int64_t)(E.lvl_hi - E.lvl_lo));
    if (np_max <= 0) return ACX_OK;
    const int64_t m = 12 * np_max;
    hipLaunchKernelGGL(k_shard_bounds<W>, dim3(1), dim3(1), 0, st, E.d, (uint32_t)E.lvl_lo, (uint32_t)E.lvl_hi, (uint32_t)c0, (uint32_t)c1, E.d_bounds);
    const dim3 grid((unsigned)((m + 1024 * kRouteItems - 1) / (1024 * kRouteItems)));
#define ACX_SHARD_EXPAND(MODE)                                                                                                                           \
    hipLaunchKernelGGL((k_shard_expand_routed<W, MODE>), grid, dim3(1024), 0, st, E.d, E.d_bounds, (int64_t)E.rank << 40, (uint32_t)E.world, rec, region_cap, \
                       (unsigned long long*)counts, (unsigned long long*)solved)
    if (E.move_mode == kMoveNf) {
        ACX_SHARD_EXPAND(kMoveNf);
    } else if (E.move_mode == kMoveNfCyclical) {
        ACX_SHARD_EXPAND(kMoveNfCyclical);
    } else {
        ACX_SHARD_EXPAND(kMoveGeneral);
    }
#undef ACX_SHARD_EXPAND
    ACX_HIP_TRY(hipGetLastError());
    return ACX_OK;
}

template <typename W> static int shard_insert(ShardEngine<W>& E, const int64_t* rec, int64_t n, int64_t c0, int64_t n_parents, int32_t* lmask, hipStream_t st) {
    E.pending = n;
    E.pending_rec = rec;
    E.pending_tag0 = 12 * c0;
    if (n_parents < 0 || 12ull * (uint64_t)n_parents > E.chunk_tags)
        return fail(ACX_E_CAPACITY, "acx_shard_insert: a chunk of %lld parents exceeds the engine's %llu", (long long)n_parents, (unsigned long long)(E.chunk_tags / 12));
    if ((uint64_t)n > E.cap_cand) return fail(ACX_E_CAPACITY, "acx_shard_insert: %lld records exceed the batch capacity %llu", (long long)n, (unsigned long long)E.cap_cand);
    if (n_parents <= 0) return ACX_OK;
    const dim3 block(256);
    ACX_HIP_TRY(hipMemsetAsync(E.d.btook, 0, (size_t)n_parents * 12, st));
    ACX_HIP_TRY(hipMemsetAsync(E.d.brepl, 0, (size_t)n_parents * 12, st));
    if (n > 0) hipLaunchKernelGGL(k_shard_insert<W>, dim3((unsigned)((n + 255) / 256)), block, 0, st, E.d, rec, n, E.pending_tag0, E.cslot, E.took_i);
    hipLaunchKernelGGL(k_shard_pack<W>, dim3((unsigned)((n_parents + 255) / 256)), block, 0, st, E.d, n_parents, lmask);
    ACX_HIP_TRY(hipGetLastError());
    return ACX_OK;
}

template <typename W>
static int shard_commit(ShardEngine<W>& E, int64_t cutoff, const int32_t* lmask, const int64_t* lprefix, const int32_t* gmask, const int64_t* gprefix,
                        int64_t gpos_base, int64_t n_commit, hipStream_t st) {
    const int64_t n = E.pending;
    E.pending = 0;
    if (n_commit < 0) return fail(ACX_E_INVAL, "acx_shard_commit: negative count");
    // checked BEFORE anything is written: an overfull rank must not touch memory behind its node arena
    if (E.nodes + (uint64_t)n_commit > E.cap_nodes) return fail(ACX_E_CAPACITY, "acx_shard_commit: node capacity exceeded (%llu + %lld > %llu)",
                                                                (unsigned long long)E.nodes, (long long)n_commit, (unsigned long long)E.cap_nodes);
    if (n > 0 && n_commit > 0) {
        hipLaunchKernelGGL(k_shard_commit<W>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, E.d, E.pending_rec, n, E.pending_tag0, cutoff, lmask, lprefix, gmask,
                           gprefix, (uint32_t)E.nodes, gpos_base, E.cslot, E.took_i, E.node_pref, (uint32_t)E.cap_nodes);
        ACX_HIP_TRY(hipGetLastError());
    }
    E.nodes += (uint64_t)n_commit;
    return ACX_OK;
}

template <typename W> static int shard_find(ShardEngine<W>& E, int64_t gpos, int64_t* id, hipStream_t st) {
    *id = -1;
    if (E.lvl_hi == E.lvl_lo || gpos < 0) return ACX_OK;
    hipLaunchKernelGGL(k_shard_find<W>, dim3(1), dim3(1), 0, st, E.d, (uint32_t)E.lvl_lo, (uint32_t)E.lvl_hi, (uint32_t)gpos, E.d_find);
    ACX_HIP_TRY(hipMemcpyAsync(id, E.d_find, 8, hipMemcpyDeviceToHost, st));
    ACX_HIP_TRY(hipStreamSynchronize(st));
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

// ------------------------------------------------------------------ sharded frontier: C ABI ----
struct acx_shard {
    acx::ShardAny any;
};

extern "C" {

int acx_shard_key_words(int L) { return L <= 29 ? 2 : 4; }

acx_shard* acx_shard_create(int L, int cyclical, int64_t node_cap, int64_t batch_cap, int64_t chunk_parents, int rank, int world) {
    if (!have_device()) return nullptr;
    if (L < 1 || L > 61 || node_cap < 1 || batch_cap < 1 || chunk_parents < 1 || world < 1 || rank < 0 || rank >= world) {
        fail(ACX_E_INVAL, "acx_shard_create: bad argument (1 <= L <= 61)");
        return nullptr;
    }
    acx_shard* h = new (std::nothrow) acx_shard();
    if (!h) return nullptr;
    h->any.wide = L > 29;
    int rc;
    if (h->any.wide) {
        h->any.e128 = new ShardEngine<u128>();
        rc = h->any.e128->init(L, cyclical, node_cap, batch_cap, chunk_parents, rank, world);
    } else {
        h->any.e64 = new ShardEngine<uint64_t>();
        rc = h->any.e64->init(L, cyclical, node_cap, batch_cap, chunk_parents, rank, world);
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

int acx_shard_seed(acx_shard* h, const int64_t* h_record, void* stream) {
    if (!h) return fail(ACX_E_INVAL, "acx_shard_seed: bad argument");
    ACX_SHARD_DISPATCH(&h->any, return shard_seed<W>(E, h_record, (hipStream_t)stream));
}

int acx_shard_level_begin(acx_shard* h, int64_t* n_local) {
    if (!h) return fail(ACX_E_INVAL, "acx_shard_level_begin: bad argument");
    ACX_SHARD_DISPATCH(&h->any, {
        E.lvl_lo = E.lvl_hi;
        E.lvl_hi = E.nodes;
        if (n_local) *n_local = (int64_t)(E.lvl_hi - E.lvl_lo);
    });
    return ACX_OK;
}

int acx_shard_expand_routed(acx_shard* h, int64_t c0, int64_t c1, int64_t* d_records, int64_t region_cap, int64_t* d_counts, int64_t* d_solved, void* stream) {
    if (!h || c0 < 0 || c1 < c0 || region_cap < 0 || !d_counts || !d_solved || (region_cap > 0 && !d_records))
        return fail(ACX_E_INVAL, "acx_shard_expand_routed: bad argument");
    ACX_SHARD_DISPATCH(&h->any, return shard_expand_routed<W>(E, c0, c1, d_records, region_cap, d_counts, d_solved, (hipStream_t)stream));
}

int acx_shard_insert(acx_shard* h, const int64_t* d_records, int64_t n, int64_t c0, int64_t n_parents, int32_t* d_child_mask, void* stream) {
    if (!h || n < 0 || c0 < 0 || (n > 0 && !d_records) || (n_parents > 0 && !d_child_mask)) return fail(ACX_E_INVAL, "acx_shard_insert: bad argument");
    ACX_SHARD_DISPATCH(&h->any, return shard_insert<W>(E, d_records, n, c0, n_parents, d_child_mask, (hipStream_t)stream));
}

int acx_shard_commit(acx_shard* h, int64_t cutoff_tag, const int32_t* d_local_mask, const int64_t* d_local_prefix, const int32_t* d_global_mask,
                     const int64_t* d_global_prefix, int64_t gpos_base, int64_t n_commit, void* stream) {
    if (!h || (n_commit > 0 && (!d_local_mask || !d_local_prefix || !d_global_mask || !d_global_prefix))) return fail(ACX_E_INVAL, "acx_shard_commit: bad argument");
    ACX_SHARD_DISPATCH(&h->any, return shard_commit<W>(E, cutoff_tag, d_local_mask, d_local_prefix, d_global_mask, d_global_prefix, gpos_base, n_commit,
                                                       (hipStream_t)stream));
}

int acx_shard_find(acx_shard* h, int64_t gpos, int64_t* id, void* stream) {
    if (!h || !id) return fail(ACX_E_INVAL, "acx_shard_find: bad argument");
    ACX_SHARD_DISPATCH(&h->any, return shard_find<W>(E, gpos, id, (hipStream_t)stream));
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
