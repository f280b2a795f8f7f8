// acx_bfs_multi.h -- MANY independent breadth-first searches in one launch: one persistent workgroup per search.
//
// BASELINE config 4 is bfs over the 1190 Miller-Schupp presentations with a budget of 1e6 nodes each
// (trivialize_miller_schupp_through_search, miller_schupp.py:95-177, runs them one after another).  Driven as 1190 separate
// acx_search calls (16 host threads, a stream each) the GPU sees a few launches of a few microseconds per batch and search:
// 1.3 s for the sweep, 7e8 nodes/s, whatever the number of threads or hardware queues.  Here a search never leaves its
// compute unit: the workgroup expands a CHUNK of 85 consecutive parents (1020 children, one per lane), folds the chunk's
// duplicates in LDS (all candidates of a chunk live in this workgroup, so the minimum tag per key is known before the visited
// table is touched), probes its own stamp table (acx_bfs.h: 8-byte slots fingerprint | parent | action, four slots per
// probe; a new key is one CAS), numbers the new states by a workgroup scan in tag order, applies the reference's budget /
// success / error rules (breadth_first.py:84-95) and appends the nodes -- then takes the next chunk.  Five dependent memory
// round trips per chunk and compute unit, and every compute unit carries two searches.
//
// The visiting order, and with it (solved, path) and the node counts, are those of the reference and of acx_search's bfs;
// tests compare the three.
#pragma once
#include "acx_bfs.h"

namespace acx {

#ifndef ACX_BFS_MULTI_THREADS
#define ACX_BFS_MULTI_THREADS 512
#endif
constexpr int kBmT = ACX_BFS_MULTI_THREADS;  // lanes of the workgroup
// Workgroup shape: 512 lanes x 3 children per lane (a chunk is 128 parents).  A search is a chain of dependent memory round
// trips per chunk, so what counts for a sweep is how many searches a compute unit carries: at <= 128 registers two
// 8-wave workgroups share a CU (the 1190 Miller-Schupp searches: 0.26 s; 1024 x 1, one workgroup per CU: 0.33 s; 512 x 2:
// 0.28 s; 512 x 4 needs more registers than two workgroups get: 0.33 s).  More children per lane lengthen the chunk (the
// probes of a lane run one after the other), fewer leave lanes idle.
// Round 3 tried the lane layout of k_bfs_expand_insert here (lane = parent, one action per wave-instruction, children in
// registers, the two-multiply hash): 0.24-0.30 s against 0.26-0.28 s -- a sweep is bound by the chain of memory round trips
// per chunk, not by vector issue, and the 128-bit searches then need more than the 128 registers that let two workgroups
// share a compute unit.  Not adopted.
// The two 128-bit groups of the Miller-Schupp sweep are its critical path (twice the vector instructions of a 64-bit group, one
// workgroup per compute unit at 133 registers, 340 workgroups for 256 compute units).  Bounding those kernels to 128 registers
// (__launch_bounds__(512, 4): no spills in the normal-form modes) gave 0.305 -> 0.28 s -- and a v_lshrrev_b64 with its amount in
// v127 of 128 (tools/check_shift64.py), the gfx950 fault of DESIGN.md section 7, at the architectural limit where no pad can
// move it.  Not adopted either.
#ifndef ACX_BFS_MULTI_ITEMS
#define ACX_BFS_MULTI_ITEMS 3
#endif
// 128-bit keys (max_relator_length 30 .. 61): TWO children per lane.  With three the kernel needs 133 registers, i.e. one workgroup
// per compute unit, and the two 128-bit groups of the Miller-Schupp sweep (n = 6, 7) were its critical path: 130 ms per group
// against 113 ms for a 64-bit one.  With two it fits the 128 registers at which two workgroups share a compute unit (no spills; the
// 64-bit-shift check of the Makefile passes): the sweep 0.30-0.32 -> 0.26-0.27 s.  (Three 64-bit workgroups per compute unit at 80
// registers -- ACX_BFS_MULTI_WAVES=6 -- spill 68 bytes per lane and lose: 0.32-0.33 s.)
#ifndef ACX_BFS_MULTI_ITEMS_128
#define ACX_BFS_MULTI_ITEMS_128 2
#endif
template <typename W> struct bm_cfg {
    static constexpr int kItems = sizeof(W) > 8 ? ACX_BFS_MULTI_ITEMS_128 : ACX_BFS_MULTI_ITEMS;
    static constexpr int kCand = kBmT * kItems;
    static constexpr int kParents = kCand / 12;
    static constexpr int kLds = kCand <= 1024 ? 2048 : (kCand <= 2048 ? 4096 : 8192);  // LDS fold table: a POWER OF TWO >= 2 x the chunk (2 * kCand = 3072
                                                                                        // made `& (kLds - 1)` two separate 1024-slot tables)
};
constexpr int kBmMaxParents = 170;  // upper bound of the chunk size over the build variants (sizes the node arenas)

enum : uint32_t { BFS_RUNNING = 0, BFS_SOLVED = 1, BFS_BUDGET = 2, BFS_EXHAUSTED = 3, BFS_MOVE_ERROR = 5, BFS_TABLE_FULL = 6 };

template <typename W> struct BfsJob {
    W* k0;  // node arena [cap_nodes]
    W* k1;
    uint32_t* parent;
    uint32_t* depth;
    uint8_t* act;
    uint8_t* tlen;
    unsigned long long* stab;  // stamp table, all ones on entry
    uint32_t stmask;
    uint32_t cap_nodes;
    W root_k0, root_k1;
    long long max_nodes;
    int32_t L, cyclical;
};

struct BfsOut {
    uint32_t status, nodes, min_len, err;
    uint32_t path_n, pad_;
    unsigned long long expanded, batches;
    unsigned long long t_phase[6];  // -DACX_BFS_MULTI_PROFILE=1: shader-clock cycles of lane 0 in expand, fold, table, number + decide, commit, tail
};

// node-arena reads bypass the vector L1: the nodes were written by other waves of this workgroup
template <typename T> __device__ __forceinline__ T ld_l2(const T* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u128 ld_l2(const u128* p) {
    const unsigned long long* q = (const unsigned long long*)p;
    return ((u128)ld_l2(q + 1) << 64) | ld_l2(q);
}

#ifndef ACX_BFS_MULTI_PROFILE
#define ACX_BFS_MULTI_PROFILE 0
#endif
#ifndef ACX_BFS_MULTI_WAVES
#define ACX_BFS_MULTI_WAVES 1
#endif
template <typename W, int MODE>
__global__ void __launch_bounds__(kBmT, ACX_BFS_MULTI_WAVES) k_bfs_multi(const BfsJob<W>* __restrict__ jobs, BfsOut* __restrict__ outs, int32_t* __restrict__ path_act,
                                                    int32_t* __restrict__ path_len, long long path_cap) {
    constexpr int kBmItems = bm_cfg<W>::kItems, kBmCand = bm_cfg<W>::kCand, kBmParents = bm_cfg<W>::kParents, kBmLds = bm_cfg<W>::kLds;
    __shared__ W s_k0[kBmCand];
    __shared__ W s_k1[kBmCand];
    __shared__ uint32_t s_slot[kBmLds];
    __shared__ uint32_t s_wsum[kBmItems * kBmT / 64];
    __shared__ unsigned long long s_err_tag;
    __shared__ uint32_t s_solved_tag, s_min_len, s_pb, s_committed, s_head, s_nodes, s_status, s_full;
#if ACX_BFS_MULTI_WAVES >= 6  // three 64-bit workgroups per compute unit: 80 registers (experiment)
    if (MODE == kMoveGeneral) {
        ACX_VGPR_PAD_W(W, "v143", "v151");
    } else {
        ACX_VGPR_PAD_W(W, "v79", "v127");
    }
#else
    if (MODE == kMoveGeneral) {
        ACX_VGPR_PAD_W(W, "v143", "v151");
    } else {
        ACX_VGPR_PAD_W(W, "v119", "v127");
    }
#endif
    const BfsJob<W> g = jobs[blockIdx.x];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    unsigned long long expanded = 0, batches = 0;
#if ACX_BFS_MULTI_PROFILE
    unsigned long long tph[6] = {}, tc = clock64();
#define ACX_BM_TICK(k) do { const unsigned long long now__ = clock64(); tph[k] += now__ - tc; tc = now__; } while (0)
#else
#define ACX_BM_TICK(k) do { } while (0)
#endif
    if (tid == 0) {
        g.k0[0] = g.root_k0;
        g.k1[0] = g.root_k1;
        g.parent[0] = kEmpty;
        g.act[0] = 0xff;
        g.tlen[0] = (uint8_t)(keyops<W>::len(g.root_k0) + keyops<W>::len(g.root_k1));
        g.depth[0] = 0;
        const uint64_t hk = stamp_hash(g.root_k0, g.root_k1);
        g.stab[(uint32_t)hk & g.stmask & ~3u] = slot_make(hk, 0u, kSelfAction);
        s_head = 0;
        s_nodes = 1;
        s_status = BFS_RUNNING;
        s_min_len = (uint32_t)g.tlen[0];
        s_full = 0;
        s_solved_tag = 0xFFFFFFFFu;
        s_err_tag = ~0ull;
        s_pb = 0xFFFFFFFFu;
        s_committed = 0;
    }
    uint32_t solved_parent = 0, solved_action = 0, err_code = 0;
    __syncthreads();
    for (;;) {
        const uint32_t head = s_head, nodes = s_nodes;
        if (head >= nodes) {  // queue exhausted (breadth_first.py:61)
            if (tid == 0) s_status = BFS_EXHAUSTED;
            break;
        }
        const uint32_t np = min(nodes - head, (uint32_t)kBmParents), m = 12u * np;
        // (s_solved_tag / s_err_tag / s_pb / s_committed are reset by lane 0 BEHIND the chunk's last reads of them, in front of the
        // barrier that ends the iteration -- a reset up here would race with the other waves' atomicMin of the expand phase)
        for (uint32_t i = tid; i < (uint32_t)kBmLds; i += kBmT) s_slot[i] = kEmpty;
        // ---- expand: candidate c = it * 1024 + tid is child (parent c / 12, action c % 12): c IS the reference's generation order
        bool probe[kBmItems];
        uint32_t pdepth[kBmItems];
        uint32_t tl_min = 0xFFFFFFFFu;
#pragma unroll
        for (int it = 0; it < kBmItems; it++) {
            const uint32_t c = (uint32_t)it * kBmT + tid;
            probe[it] = false;
            pdepth[it] = 0;
            W c0 = 0, c1 = 0;
            if (c < m) {
                const uint32_t p = c / 12u, a = c - 12u * p, pid = head + p;
                const W pk0 = ld_l2(g.k0 + pid), pk1 = ld_l2(g.k1 + pid);
                pdepth[it] = ld_l2(g.depth + pid);  // with the key: the commit below then needs no load of its own
                Pres<W> s;
                key_to_pres<W>(pk0, pk1, s);
                const int e = search_move<W, MODE>(s, (int)a, g.L, g.cyclical != 0);
                if (e) atomicMin(&s_err_tag, ((unsigned long long)c << 8) | (unsigned long long)e);
                c0 = keyops<W>::make(s.w0, s.n0);
                c1 = keyops<W>::make(s.w1, s.n1);
                const uint32_t tl = (uint32_t)(s.n0 + s.n1);
                tl_min = min(tl_min, tl);
                if (tl == 2) atomicMin(&s_solved_tag, c);  // breadth_first.py:84: tested before the dedup
                probe[it] = !(c0 == pk0 && c1 == pk1);      // an unchanged state is its (visited) parent
            }
            s_k0[c] = c0;
            s_k1[c] = c1;
        }
        for (int o = 32; o > 0; o >>= 1) tl_min = min(tl_min, (uint32_t)__shfl_xor((int)tl_min, o));
        if (lane == 0 && tl_min < s_min_len) atomicMin(&s_min_len, tl_min);
        __syncthreads();
        ACX_BM_TICK(0);
        // ---- duplicates inside the chunk: LDS table of candidate numbers, minimum (= first discoverer) per key
        uint32_t ls[kBmItems];
#pragma unroll
        for (int it = 0; it < kBmItems; it++) {
            const uint32_t c = (uint32_t)it * kBmT + tid;
            ls[it] = 0;
            if (!probe[it]) continue;
            const W c0 = s_k0[c], c1 = s_k1[c];
            uint32_t q = (uint32_t)(stamp_hash(c0, c1) >> 40) & (kBmLds - 1);
            for (;;) {
                uint32_t v = s_slot[q];
                if (v == kEmpty) {
                    v = atomicCAS(&s_slot[q], kEmpty, c);
                    if (v == kEmpty) break;
                }
                if (s_k0[v] == c0 && s_k1[v] == c1) {
                    if (v > c) atomicMin(&s_slot[q], c);
                    break;
                }
                q = (q + 1) & (kBmLds - 1);
            }
            ls[it] = q;
        }
        __syncthreads();
        ACX_BM_TICK(1);
        // ---- the chunk's distinct keys against the visited table: seen before, or one CAS ----------------------------------------
        // (54 % of a chunk's 22 us, -DACX_BFS_MULTI_PROFILE=1.  Taking the first turn of a lane's three candidates together --
        // bucket loads, then claims, issued before the first result is used -- changed nothing: the phase lasts as long as the
        // slowest of the workgroup's ~1000 probes, and at a load factor of 0.5 some lane always needs a second bucket or the
        // key of an occupant.)
        uint32_t win[kBmItems];
#pragma unroll
        for (int it = 0; it < kBmItems; it++) {
            const uint32_t c = (uint32_t)it * kBmT + tid;
            win[it] = 0;
            if (!probe[it] || s_slot[ls[it]] != c) continue;
            const W c0 = s_k0[c], c1 = s_k1[c];
            const uint32_t p = c / 12u, a = c - 12u * p;
            const uint64_t hk = stamp_hash(c0, c1);
            const unsigned long long me = slot_make(hk, head + p, a);
            uint32_t base = (uint32_t)hk & g.stmask & ~3u, probes = 0;
            bool open = true;
            while (open) {
                // the bucket as TWO 16-byte loads (rounds 2-3: four 8-byte agent-scope loads, i.e. four L2 requests per probing lane --
                // three quarters of the sweep's 8.2e9 L2 requests).  Plain loads may be served by a stale line of the vector L1; that is
                // harmless here as in the fused kernel: a slot only ever moves free -> stamp of key K, so a stale "free" costs one
                // failed CAS whose return value is then looked at, and a stale stamp is the stamp
                const ulonglong2 lo = *(const ulonglong2*)(g.stab + base), hi = *(const ulonglong2*)(g.stab + base + 2);
                const unsigned long long v[4] = {lo.x, lo.y, hi.x, hi.y};
                uint32_t cand = 0;
#pragma unroll
                for (int j = 0; j < 4; j++) cand |= (v[j] == kSlotFree || (v[j] >> 36) == (me >> 36)) ? 1u << j : 0u;
                while (cand) {
                    const uint32_t j = (uint32_t)__builtin_ctz(cand);
                    cand &= cand - 1;
                    unsigned long long st = j == 0 ? v[0] : (j == 1 ? v[1] : (j == 2 ? v[2] : v[3]));
                    if (st == kSlotFree) {
                        st = atomicCAS(g.stab + base + j, kSlotFree, me);
                        if (st == kSlotFree) {
                            win[it] = 1;
                            open = false;
                            break;
                        }
                    }
                    if ((st >> 36) == (me >> 36)) {  // fingerprint match: rebuild the occupant's key from its parent
                        const uint32_t hp = slot_parent(st), ha = slot_action(st);
                        W q0 = ld_l2(g.k0 + hp), q1 = ld_l2(g.k1 + hp);
                        if (ha != kSelfAction) {
                            Pres<W> s;
                            key_to_pres<W>(q0, q1, s);
                            (void)search_move<W, MODE>(s, (int)ha, g.L, g.cyclical != 0);
                            q0 = keyops<W>::make(s.w0, s.n0);
                            q1 = keyops<W>::make(s.w1, s.n1);
                        }
                        if (q0 == c0 && q1 == c1) {  // a state of an earlier chunk (this chunk's duplicates were folded in LDS)
                            open = false;
                            break;
                        }
                    }
                }
                base = (base + 4) & g.stmask;
                if (open && ++probes > g.stmask / 4) {
                    s_full = 1;
                    open = false;
                }
            }
        }
        // ---- number the new states in tag order: candidates 0..1023 (it = 0) come before 1024..2047 (it = 1) --------------------
        uint32_t rank[kBmItems];
#pragma unroll
        for (int it = 0; it < kBmItems; it++) {
            const unsigned long long b = __ballot(win[it] != 0);
            rank[it] = (uint32_t)__popcll(b & ((1ull << lane) - 1ull));
            if (lane == 0) s_wsum[it * (kBmT / 64) + wave] = (uint32_t)__popcll(b);
        }
        __syncthreads();
        ACX_BM_TICK(2);
        uint32_t total = 0;
        {
            uint32_t before[kBmItems] = {};
            for (uint32_t w = 0; w < (uint32_t)kBmItems * (kBmT / 64); w++) {
                const uint32_t v = s_wsum[w];
#pragma unroll
                for (int it = 0; it < kBmItems; it++)
                    if (w < (uint32_t)it * (kBmT / 64) + wave) before[it] += v;
                total += v;
            }
#pragma unroll
            for (int it = 0; it < kBmItems; it++) rank[it] += before[it];
        }
        // ---- the reference's decisions (breadth_first.py:84-95) ---------------------------------------------------------------
        const long long need = g.max_nodes - (long long)nodes;
        if (need >= 1 && (long long)total >= need) {
#pragma unroll
            for (int it = 0; it < kBmItems; it++)
                if (win[it] && (long long)rank[it] == need - 1) s_pb = ((uint32_t)it * kBmT + tid) / 12u;  // parent of the state that reaches the budget
        }
        __syncthreads();
        uint32_t p_end = np - 1;
        bool budget_hit = false;
        if (need < 1) {  // only the very first parent can see this (budget <= 1)
            p_end = 0;
            budget_hit = true;
        } else if (s_pb != 0xFFFFFFFFu) {
            p_end = s_pb;
            budget_hit = true;
        }
        const uint32_t stag = s_solved_tag;
        bool is_solved = stag != 0xFFFFFFFFu && stag / 12u <= p_end;
        const unsigned long long et = s_err_tag;
        const bool err_hit = et != ~0ull && (uint32_t)((et >> 8) / 12u) <= p_end && !(is_solved && (unsigned long long)stag < (et >> 8));
        if (err_hit) is_solved = false;
        const uint32_t cutoff = is_solved ? stag : 12u * (p_end + 1);
        ACX_BM_TICK(3);
        // ---- commit the new states below the cutoff --------------------------------------------------------------------------
#pragma unroll
        for (int it = 0; it < kBmItems; it++) {
            const uint32_t c = (uint32_t)it * kBmT + tid;
            if (!win[it] || c >= cutoff) continue;
            const uint32_t id = nodes + rank[it];
            atomicMax(&s_committed, rank[it] + 1);
            if (id >= g.cap_nodes) continue;  // cannot happen: cap_nodes covers the budget plus a chunk
            const uint32_t p = c / 12u, pid = head + p;
            const W c0 = s_k0[c], c1 = s_k1[c];
            g.k0[id] = c0;
            g.k1[id] = c1;
            g.parent[id] = pid;
            g.act[id] = (uint8_t)(c - 12u * p);
            g.tlen[id] = (uint8_t)(keyops<W>::len(c0) + keyops<W>::len(c1));
            g.depth[id] = pdepth[it] + 1;
        }
        batches++;
        expanded += is_solved ? stag / 12u + 1 : p_end + 1;
        __syncthreads();  // (also: the nodes written above are in L2 before anybody reads them in the next chunk)
        ACX_BM_TICK(4);
        if (tid == 0) {
            s_nodes = nodes + s_committed;
            s_head = head + p_end + 1;
            s_solved_tag = 0xFFFFFFFFu;  // for the next chunk (every read of this chunk's values is behind the barrier above)
            s_err_tag = ~0ull;
            s_pb = 0xFFFFFFFFu;
            s_committed = 0;
            if (s_full) s_status = BFS_TABLE_FULL;
            else if (err_hit) s_status = BFS_MOVE_ERROR;
            else if (is_solved) s_status = BFS_SOLVED;
            else if (budget_hit) s_status = BFS_BUDGET;
        }
        if (is_solved) {
            solved_parent = head + stag / 12u;
            solved_action = stag % 12u;
        }
        if (err_hit) err_code = (uint32_t)(et & 0xff);
        __syncthreads();
        ACX_BM_TICK(5);
        if (s_status != BFS_RUNNING) break;
    }
#undef ACX_BM_TICK
    __syncthreads();
    if (tid == 0) {
        BfsOut* out = outs + blockIdx.x;
        out->status = s_status;
        out->nodes = s_nodes;
        out->min_len = s_status == BFS_SOLVED ? 2u : s_min_len;
        out->err = err_code;
        out->expanded = expanded;
        out->batches = batches;
#if ACX_BFS_MULTI_PROFILE
        for (int k = 0; k < 6; k++) out->t_phase[k] = tph[k];
#endif
        out->path_n = 0;
        if (s_status == BFS_SOLVED) {  // path of the parent + (action, 2)
            uint32_t v = solved_parent;
            const uint32_t dep = ld_l2(g.depth + v);
            out->path_n = dep + 2;
            int32_t* pa = path_act + (size_t)blockIdx.x * path_cap;
            int32_t* pl = path_len + (size_t)blockIdx.x * path_cap;
            if ((long long)dep + 2 <= path_cap) {
                pa[dep + 1] = (int32_t)solved_action;
                pl[dep + 1] = 2;
                for (uint32_t k = dep;; k--) {
                    const uint8_t a = ld_l2(g.act + v);
                    pa[k] = a == 0xff ? -1 : (int32_t)a;
                    pl[k] = ld_l2(g.tlen + v);
                    if (k == 0) break;
                    v = ld_l2(g.parent + v);
                }
            }
        }
    }
}

}  // namespace acx
