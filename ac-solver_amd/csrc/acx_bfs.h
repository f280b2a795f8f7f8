// acx_bfs.h -- batch kernels of the fused single-GPU BFS (bfs, ac_solver/search/breadth_first.py:55-97) on the STAMP TABLE.
//
// What round 1's inline-key table cost (profiles/r1_bfs_1e8_pmc_summary.txt) and what a random access costs on MI355X
// (tools/microbench_table.hip: reads 50-55 G/s, 64-bit CAS 18 G/s, stores 22-24 G/s, and everything worse once the table
// outgrows ~4 GB) decided this layout.  Of the children a BFS of the AC graph generates (tools/dup_structure.cpp, AK(3)):
// 33 % leave the state unchanged, 8 % undo the move that made their parent, 37 % are new, 18 % duplicate a state first
// seen in the SAME batch (more than half of those inside the same 85 parents) and only 4 % meet a state of an older
// batch.  So the visited table is mostly written, and the three memory-side operations a new key used to cost (CAS of the
// stamp + two key words stored into a 32-byte entry of an 8.6 GB table) are the thing to cut:
//
//   * one 8-byte slot per state:  fingerprint(28) | parent node id(32) | action(4), all ones = free.  A slot names the
//     state as "child `action` of node `parent`" (action 15: the node itself, used for the root), so the full key of an
//     occupant is RECOMPUTED from the parent's key in the node arena (two 8-byte loads that hit in L2 / Infinity Cache
//     far more often than a table entry does, plus one apply_move) -- and only when the fingerprint matches.  Exact set
//     semantics are kept (full-key compare before "seen"); a new state costs ONE CAS and no store.
//   * the batch is told apart by the parent id: parents of the running batch are the nodes >= pbegin, everything below
//     is a committed state.  (parent, action) is also the reference's generation order, so the minimum-tag fold among
//     equal keys is a 64-bit atomicMin on the slot, as before.  No epochs, no commit pass.
//   * expand and insert are ONE kernel: the child never travels through a candidate arena (16 B written + 16 B read
//     per child before); k_bfs_compact recomputes the winners' keys the same way.
//   * children equal to their parent are dropped before any probe (visited by construction), and a workgroup first
//     folds the duplicates among its own 256 candidates in an LDS table: only the smallest tag of each key inside the
//     tile goes to the global table.
//
// Visibility across the eight non-coherent L2s: as in round 1, only the returned values of the device-scope CAS /
// atomicMin decide; a plain (possibly stale) load of a slot can only show an older state of the same slot, which the
// atomic then corrects.  Keys of occupants come from the node arena written by earlier kernels.
#pragma once
#include "acx_frontier.h"

namespace acx {

constexpr unsigned long long kSlotFree = ~0ull;
constexpr uint32_t kSelfAction = 15u;  // slot names the node itself (the root)

// hash of the stamp table: one multiply-xorshift round per key word (bucket = low bits, fingerprint = bits 36..63, LDS fold
// slot = bits 40..); acx_frontier.h's hash_key costs five 64-bit multiplies = fifteen quarter-rate instructions per child
ACX_HD uint64_t stamp_mix(uint64_t h, uint64_t w) {
    h = (h ^ w) * 0x9E3779B97F4A7C15ull;
    return h ^ (h >> 29);
}
ACX_HD uint64_t stamp_hash(uint64_t k0, uint64_t k1) { return stamp_mix(stamp_mix(0, k0), k1); }
ACX_HD uint64_t stamp_hash(u128 k0, u128 k1) {
    uint64_t h = stamp_mix(stamp_mix(0, (uint64_t)k0), (uint64_t)(k0 >> 64));
    return stamp_mix(stamp_mix(h, (uint64_t)k1), (uint64_t)(k1 >> 64));
}

ACX_HD unsigned long long slot_make(uint64_t hk, uint32_t pid, uint32_t act) { return (hk & ~((1ull << 36) - 1)) | ((unsigned long long)pid << 4) | act; }
ACX_HD uint32_t slot_parent(unsigned long long s) { return (uint32_t)(s >> 4); }
ACX_HD uint32_t slot_action(unsigned long long s) { return (uint32_t)s & 15u; }

// The move code of the BFS kernels.  MODE 0: the general ACMove (apply_move) -- needed when the ROOT is not in normal form
// (unreduced input: its children can even have an empty relator, utils.py:261-278).  With a root in normal form every node
// of the search is, and the kernels run the shorter apply_move_nf: MODE 1 for cyclical = False, MODE 2 for cyclical = True.
enum : int { kMoveGeneral = 0, kMoveNf = 1, kMoveNfCyclical = 2 };
template <typename W, int MODE> __device__ __forceinline__ int search_move(Pres<W>& s, int a, int L, bool cyclical) {
    if (MODE == kMoveNf) return apply_move_nf<W, kSearchSafe>(s, a, L, false);
    if (MODE == kMoveNfCyclical) return apply_move_nf<W, kSearchSafe>(s, a, L, true);
    return apply_move<W, kSearchSafe>(s, a, L, cyclical);
}

// key of the state a slot names: child `act` of node `pid` (or the node itself)
template <typename W, int MODE> __device__ __forceinline__ void slot_key(const SearchDev<W>& d, uint32_t pid, uint32_t act, W& q0, W& q1) {
    q0 = d.k0[pid];
    q1 = d.k1[pid];
    if (act != kSelfAction) {
        Pres<W> s;
        key_to_pres<W>(q0, q1, s);
        (void)search_move<W, MODE>(s, (int)act, d.L, d.cyclical != 0);
        q0 = keyops<W>::make(s.w0, s.n0);
        q1 = keyops<W>::make(s.w1, s.n1);
    }
}

template <typename W> __global__ void k_bfs_root(SearchDev<W> d, W k0, W k1, uint32_t tl) {
    ACX_VGPR_PAD("v15");
    d.k0[0] = k0;
    d.k1[0] = k1;
    d.parent[0] = kEmpty;
    d.act[0] = 0xff;
    d.tlen[0] = (uint8_t)tl;
    d.depth[0] = 0;
    const uint64_t hk = stamp_hash(k0, k1);
    d.stab[(uint32_t)hk & d.stmask & ~3u] = slot_make(hk, 0u, kSelfAction);  // first slot of its bucket
}

// ---- expand + dedup of a batch in ONE launch ------------------------------------------------------------------------------------
// tag t = 12 * p + a (parent p of the batch, action a).  btook[t] = 1 when t took its slot (claimed it free, or replaced a
// larger tag of the same key); brepl[tag] = 1 is set for a holder that was replaced (brepl is zero on entry: cleared once per
// search, k_bfs_compact zeroes what a batch set).
// Tile shape (round 3, the sharded engine's expansion): 128 parents x 12 actions per 512-lane workgroup; wave w of a group of
// four computes actions 3w .. 3w + 2 of the group's 64 parents, ONE ACTION PER WAVE-INSTRUCTION.  Rounds 1-2 had a lane per
// (parent, action) in tag order, 256 lanes per tile (1024 / 512 lanes measured slower then: 14.0 / 13.2 vs 12.8 ms): the twelve
// actions of a parent sat in adjacent lanes, so every wave issued the concatenation AND the conjugation path of every child --
// the kernel is atomics-bound, but its vector units were ~65 % busy as well (2.3e9 wave-instructions per 1e8-node search).
// The LDS fold now sees 1536 candidates instead of 256, so fewer duplicates of a batch meet in the global table, and the
// hash costs two 64-bit multiplies instead of five.  1e8-node AK(3) search: 10.2 -> 9.7 ms.
constexpr int kBfsParents = 128, kBfsItems = 3, kBfsThreads = kBfsParents * 4, kBfsTile = kBfsParents * 12, kBfsFold = 4096, kBfsTileBits = 11;


template <typename W, int MODE>
__global__ void __launch_bounds__(kBfsThreads) k_bfs_expand_insert(SearchDev<W> d, uint32_t pbegin, uint32_t np, const BfsCursor* __restrict__ cur = nullptr) {
    __shared__ W s_k0[kBfsTile];
    __shared__ W s_k1[kBfsTile];
    __shared__ uint32_t s_slot[kBfsFold];
    __shared__ uint32_t s_took[kBfsTile / 4];  // one byte per tag of the tile, written out as dwords
    ACX_VGPR_PAD_W(W, "v79", "v111");  // (with "v103" the 128-bit kernel had exactly 104 registers and a v_lshrrev_b64 with its amount in v103: tools/check_shift64.py flagged it, and the budget-sweep test failed on the GPU -- DESIGN.md section 7)
    if (cur) {  // run-ahead mode: the batch is whatever the cursor says (np arrives as the batch capacity)
        if (cur->status) return;
        pbegin = cur->head;
        const uint32_t avail = cur->nodes - pbegin;
        np = avail < np ? avail : np;
    }
    const uint32_t tid = threadIdx.x, l = (tid & 63u) + 64u * (tid >> 8), w = (tid >> 6) & 3u;
    if (blockIdx.x * kBfsParents >= np) return;  // (a full-size grid over a short batch)
    for (uint32_t i = tid; i < (uint32_t)kBfsFold; i += kBfsThreads) s_slot[i] = kEmpty;
    if (tid < kBfsTile / 4) s_took[tid] = 0;
    const uint32_t p = blockIdx.x * kBfsParents + l, pid = pbegin + p;
    const bool live = p < np;
    W pk0 = 0, pk1 = 0;
    uint32_t pa = 0xffu;
    if (live) {
        pk0 = d.k0[pid];
        pk1 = d.k1[pid];
        if (MODE == kMoveNf) pa = d.act[pid];  // 0xff for the root
    }
    W c0[kBfsItems], c1[kBfsItems];
    bool probe[kBfsItems];
    uint32_t tl_min = 0xFFFFFFFFu;
#pragma unroll
    for (int it = 0; it < kBfsItems; it++) {
        const uint32_t a = (uint32_t)__builtin_amdgcn_readfirstlane((int)(w * kBfsItems + it));  // uniform across the wave
        const uint32_t j = a * (uint32_t)kBfsParents + l;  // the child's slot in the tile (action major: conflict-free LDS rows)
        probe[it] = false;
        c0[it] = c1[it] = 0;
        if (live) {
            const uint32_t t = 12u * p + a;  // tag inside the batch: the reference's generation order
            Pres<W> s;
            key_to_pres<W>(pk0, pk1, s);
            const int e = search_move<W, MODE>(s, (int)a, d.L, d.cyclical != 0);
            // the reference's ACMove raises here -- but only if it gets this far (k_decide_tab): the FIRST such move of the batch counts
            if (e) atomicMin(d.err_tag, ((unsigned long long)t << 8) | (unsigned long long)e);
            c0[it] = keyops<W>::make(s.w0, s.n0);
            c1[it] = keyops<W>::make(s.w1, s.n1);
            const uint32_t tl = (uint32_t)(s.n0 + s.n1);
            tl_min = min(tl_min, tl);
            if (d.first_len && tl < d.min_len_start) atomicMin(&d.first_len[tl], (unsigned long long)t);  // a candidate for "New minimal length found"
            if (tl == 2) atomicMin(d.solved_tag, (unsigned long long)t);  // breadth_first.py:84: tested before the dedup
            probe[it] = !(c0[it] == pk0 && c1[it] == pk1);                // unchanged state = its (visited) parent
#ifndef ACX_BFS_NO_UNDO_DROP
            // Normal-form search with cyclical = False: the child of action inverse(act[parent]) IS the parent's own tree parent
            // (g^-1 (g r g^-1) g = r and (r_i r_j) r_j^-1 = r_i as reduced words, and the result fits because it did before) -- a
            // visited state, 8 % of all children: no probe.  One byte per parent, no dependent load (round 2's grandparent test
            // compared keys: two dependent loads in front of the barrier, slower than the probes it saved).  Checked on the
            // oracle for every node of the CPU suite's sharded searches (tests/test_sharded_cpu.py) and by the searches' own
            // node-for-node comparisons with the oracle.
            if (MODE == kMoveNf && pa < 12u && a == (pa < 4u ? (pa ^ 2u) : (pa < 8u ? pa + 4u : pa - 4u))) probe[it] = false;
#endif
        }
        s_k0[j] = c0[it];
        s_k1[j] = c1[it];
    }
    {  // smallest total length of the batch: wave minimum, then one atomic per wave that lowers it
        for (int o = 32; o > 0; o >>= 1) tl_min = min(tl_min, (uint32_t)__shfl_xor((int)tl_min, o));
        if ((tid & 63u) == 0 && tl_min < *(volatile uint32_t*)d.min_len) atomicMin(d.min_len, tl_min);
    }
    __syncthreads();
    // ---- duplicates inside the tile: LDS table of (tag inside the tile) << 11 | slot j, minimum = first discoverer ----------------
    uint32_t ls[kBfsItems], me[kBfsItems];
    uint64_t hk[kBfsItems];
#pragma unroll
    for (int it = 0; it < kBfsItems; it++) {
        const uint32_t a = w * kBfsItems + it, j = a * (uint32_t)kBfsParents + l;
        me[it] = ((12u * l + a) << kBfsTileBits) | j;
        ls[it] = 0;
        hk[it] = 0;
        if (!probe[it]) continue;
        hk[it] = stamp_hash(c0[it], c1[it]);
        uint32_t q = (uint32_t)(hk[it] >> 40) & (kBfsFold - 1);
        for (;;) {
            uint32_t v = s_slot[q];
            if (v == kEmpty) {
                v = atomicCAS(&s_slot[q], kEmpty, me[it]);
                if (v == kEmpty) break;
            }
            if (s_k0[v & ((1u << kBfsTileBits) - 1u)] == c0[it] && s_k1[v & ((1u << kBfsTileBits) - 1u)] == c1[it]) {  // any holder of this slot has my key
                if (v > me[it]) atomicMin(&s_slot[q], me[it]);
                break;
            }
            q = (q + 1) & (kBfsFold - 1);
        }
        ls[it] = q;
    }
    __syncthreads();
    // ---- the tile's winners probe the global stamp table.  Slots are probed a BUCKET at a time: four slots = one aligned
    // 32-byte sector = one memory access.  A key lives in the first slot that was free in scan order (bucket of its hash from
    // slot 0, then the following buckets), so a later probe meets it before it meets a free slot.  Free slot -> one CAS; matching
    // fingerprint -> the occupant's key is rebuilt from its parent and compared in full; equal keys of the running batch fold to
    // the smaller (parent, action) with a 64-bit atomicMin (the replaced candidate is flagged by the one that replaced it).
#pragma unroll
    for (int it = 0; it < kBfsItems; it++) {
        if (!(probe[it] && s_slot[ls[it]] == me[it])) continue;
        const uint32_t a = w * kBfsItems + it;
        const unsigned long long mine = slot_make(hk[it], pid, a);
        uint32_t base = (uint32_t)hk[it] & d.stmask & ~3u, probes = 0, took = 0;
        bool open = true;
        while (open) {
            const ulonglong2 lo = *(const ulonglong2*)(d.stab + base), hi = *(const ulonglong2*)(d.stab + base + 2);
            const unsigned long long v0 = lo.x, v1 = lo.y, v2 = hi.x, v3 = hi.y;
            auto hot = [&](unsigned long long v) { return v == kSlotFree || (v >> 36) == (mine >> 36); };  // free, or my fingerprint
            uint32_t cand = (hot(v0) ? 1u : 0u) | (hot(v1) ? 2u : 0u) | (hot(v2) ? 4u : 0u) | (hot(v3) ? 8u : 0u);
            while (cand) {
                const uint32_t jj = (uint32_t)__builtin_ctz(cand);
                cand &= cand - 1;
                unsigned long long st = jj == 0 ? v0 : (jj == 1 ? v1 : (jj == 2 ? v2 : v3));
                unsigned long long* slot = d.stab + base + jj;
                if (st == kSlotFree) {
                    st = atomicCAS(slot, kSlotFree, mine);
                    if (st == kSlotFree) {
                        took = 1;
                        open = false;
                        break;
                    }
                }
                if ((st >> 36) == (mine >> 36)) {  // fingerprint match: rebuild the occupant's key
                    const uint32_t hp = slot_parent(st), ha = slot_action(st);
                    W q0, q1;
                    slot_key<W, MODE>(d, hp, ha, q0, q1);
                    if (q0 == c0[it] && q1 == c1[it]) {
                        if (hp >= pbegin && ha != kSelfAction && st > mine) {  // a candidate of this batch with a larger tag
                            const unsigned long long prev = atomicMin(slot, mine);
                            if (prev > mine) {
                                took = 1;
                                d.brepl[12u * (slot_parent(prev) - pbegin) + slot_action(prev)] = 1;  // no longer the first discoverer
                            }
                        }
                        open = false;
                        break;
                    }
                }
            }
            base = (base + 4) & d.stmask;
            if (open && ++probes > d.stmask / 4) {
                atomicOr(d.err, kErrTableFull);
                open = false;
            }
        }
        if (took) ((uint8_t*)s_took)[12u * l + a] = 1;
    }
    __syncthreads();
    {  // btook of the tile's tags, coalesced (every tag of the batch is written: zero = did not take a slot)
        const uint32_t m = 12u * np, t0 = blockIdx.x * (uint32_t)kBfsTile;
        if (tid < kBfsTile / 4 && t0 + 4u * tid < m) ((uint32_t*)(d.btook + t0))[tid] = s_took[tid];  // (m is a multiple of 4; t0 of 1536)
    }
}

// Winners -> nodes in one pass: k_compact_tab (acx_frontier.h) with the winners' keys recomputed from their parents.
template <typename W, int MODE>
__global__ void __launch_bounds__(256) k_bfs_compact(SearchDev<W> d, uint32_t pbegin, uint32_t m, uint32_t base, uint32_t cap_nodes, uint32_t epoch,
                                                     unsigned long long* __restrict__ status, uint32_t* __restrict__ ticket, uint32_t* __restrict__ total_out,
                                                     const BfsCursor* __restrict__ cur = nullptr) {
    __shared__ uint32_t s_tile, s_prefix, s_wsum[4];
    __shared__ uint16_t s_list[kCompactTile];
#ifdef ACX_HAZARD_REPRO  // tools/hazard24/repro_compact.sh: the kernel as it was when it corrupted (exactly the 32 registers it uses)
    asm volatile("" ::: "v31");
#else
    ACX_VGPR_PAD_W(W, "v47", "v63");
#endif
    if (cur) {  // run-ahead mode (m arrives as the capacity 12 * bmax)
        if (cur->status) return;
        pbegin = cur->head;
        const uint32_t avail = 12u * (cur->nodes - pbegin);
        m = avail < m ? avail : m;
        base = cur->nodes;
        epoch = cur->batches + 1;
    }
    if (blockIdx.x >= (m + kCompactTile - 1) / kCompactTile) return;  // a full-size grid over a short batch: only the batch's tiles take tickets
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if (tid == 0) s_tile = atomicAdd(ticket, 1u);
    __syncthreads();
    const uint32_t tile = s_tile, ntiles = (m + kCompactTile - 1) / kCompactTile;
    const uint32_t t0 = tile * kCompactTile + tid * kCompactItems;
    uint32_t fl = 0;  // bit i: candidate t0 + i is a winner (took its slot and was not replaced)
    if (t0 + kCompactItems <= m) {
#pragma unroll
        for (uint32_t q = 0; q < kCompactItems / 8; q++) {
            const unsigned long long tb = *(const unsigned long long*)(d.btook + t0 + 8 * q), rb = *(const unsigned long long*)(d.brepl + t0 + 8 * q);
            if (rb) *(unsigned long long*)(d.brepl + t0 + 8 * q) = 0;  // zero again for the next batch (no memset launch per batch)
            const unsigned long long w = tb & ~rb;  // bytes are 0 / 1
#pragma unroll
            for (uint32_t i = 0; i < 8; i++) fl |= (uint32_t)((w >> (8u * i)) & 1ull) << (8 * q + i);
        }
    } else {
        for (uint32_t i = 0; i < kCompactItems; i++)
            if (t0 + i < m) {
                const uint8_t rb = d.brepl[t0 + i];
                if (rb) d.brepl[t0 + i] = 0;
                if (d.btook[t0 + i] && !rb) fl |= 1u << i;
            }
    }
    const uint32_t cnt = (uint32_t)__popc(fl);
    uint32_t incl = cnt;
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = (uint32_t)__shfl_up((int)incl, o);
        if (lane >= (uint32_t)o) incl += v;
    }
    if (lane == 63) s_wsum[wave] = incl;
    __syncthreads();
    uint32_t wbase = 0;
    for (uint32_t w = 0; w < wave; w++) wbase += s_wsum[w];
    const uint32_t block_total = s_wsum[0] + s_wsum[1] + s_wsum[2] + s_wsum[3];
    {
        uint32_t pos = wbase + incl - cnt, f = fl;
        while (f) {
            const uint32_t i = (uint32_t)__builtin_ctz(f);
            f &= f - 1;
            s_list[pos++] = (uint16_t)(tid * kCompactItems + i);
        }
    }
    if (wave == 0) {  // decoupled look-back over the tiles' status words (see k_compact_tab)
        const unsigned long long tagged = (unsigned long long)epoch << 34;
        if (lane == 0)
            __hip_atomic_store(&status[tile], tagged | ((tile == 0 ? kTileIncl : kTileAgg) << 32) | block_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t excl = 0;
        long long j0 = (long long)tile - 1;
        while (j0 >= 0) {
            const long long j = j0 - (long long)lane;
            unsigned long long w = tagged | (kTileIncl << 32);
            for (;;) {
                if (j >= 0) w = __hip_atomic_load(&status[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const bool ready = (w >> 34) == (unsigned long long)epoch && ((w >> 32) & 3ull) != 0;
                if (__all(ready)) break;
                __builtin_amdgcn_s_sleep(1);
            }
            const unsigned long long inc = __ballot(((w >> 32) & 3ull) == kTileIncl);
            const uint32_t first = inc ? (uint32_t)__builtin_ctzll(inc) : 63u;
            uint32_t v = lane <= first ? (uint32_t)w : 0u;
            for (int o = 32; o > 0; o >>= 1) v += (uint32_t)__shfl_xor((int)v, o);
            excl += v;
            if (inc) break;
            j0 -= 64;
        }
        if (lane == 0) {
            if (tile != 0) __hip_atomic_store(&status[tile], tagged | (kTileIncl << 32) | (excl + block_total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_prefix = excl;
            if (tile == ntiles - 1) {
                *total_out = excl + block_total;
                *ticket = 0;
            }
        }
    }
    __syncthreads();
    const uint32_t first_id = base + s_prefix, tbase = tile * kCompactTile;
#ifdef ACX_HAZARD_REPRO
    // the write loop of commit 57c6f83 (one winner per lane and round), kept verbatim for the reproducer of DESIGN.md section 7
    for (uint32_t j = tid; j < block_total; j += 256) {
        const uint32_t id = first_id + j;
        if (id >= cap_nodes) break;  // beyond the budget: never read
        const uint32_t t = tbase + s_list[j];
        const uint32_t p = t / 12u, pid = pbegin + p, a = t - 12u * p;
        Pres<W> s;
        key_to_pres<W>(d.k0[pid], d.k1[pid], s);
        (void)search_move<W, MODE>(s, (int)a, d.L, d.cyclical != 0);
        d.k0[id] = keyops<W>::make(s.w0, s.n0);
        d.k1[id] = keyops<W>::make(s.w1, s.n1);
        d.parent[id] = pid;
        d.act[id] = (uint8_t)a;
        d.tlen[id] = (uint8_t)(s.n0 + s.n1);
        d.depth[id] = d.depth[pid] + 1;
    }
#else
    // kU winners per lane and round: their parent keys and depths are loaded together, then the moves, then the stores
    // (kU = 4 -- one memory round trip for four nodes -- measured no faster than 1 on the 1e8-node search: the pass is not
    // bound by this loop's latency; 1 keeps the kernel at 36 registers)
#ifndef ACX_COMPACT_UNROLL
#define ACX_COMPACT_UNROLL 1
#endif
    constexpr uint32_t kU = ACX_COMPACT_UNROLL;
    for (uint32_t j0 = tid; j0 < block_total; j0 += 256 * kU) {
        W pk0[kU], pk1[kU];
        uint32_t pid[kU], act[kU], dep[kU];
        bool on[kU];
#pragma unroll
        for (uint32_t u = 0; u < kU; u++) {
            const uint32_t j = j0 + 256 * u;
            on[u] = j < block_total && first_id + j < cap_nodes;  // ids beyond the budget are never read
            const uint32_t t = tbase + s_list[on[u] ? j : 0];
            const uint32_t p = t / 12u;
            pid[u] = pbegin + p;
            act[u] = t - 12u * p;
            pk0[u] = on[u] ? d.k0[pid[u]] : (W)0;
            pk1[u] = on[u] ? d.k1[pid[u]] : (W)0;
            dep[u] = on[u] ? d.depth[pid[u]] : 0u;
        }
#pragma unroll
        for (uint32_t u = 0; u < kU; u++) {
            if (!on[u]) continue;
            const uint32_t id = first_id + j0 + 256 * u;
            Pres<W> s;
            key_to_pres<W>(pk0[u], pk1[u], s);
            (void)search_move<W, MODE>(s, (int)act[u], d.L, d.cyclical != 0);
#ifdef ACX_BFS_CHECK_NF
            {
                Pres<W> g;
                key_to_pres<W>(pk0[u], pk1[u], g);
                (void)apply_move<W, kSearchSafe>(g, (int)act[u], d.L, d.cyclical != 0);
                if (g.w0 != s.w0 || g.w1 != s.w1 || g.n0 != s.n0 || g.n1 != s.n1 || !is_normal_form<W>(s, d.cyclical != 0))
                    printf("COMPACT: node %u = move(%u, %u): parent %llx %llx nf %llx %llx general %llx %llx\n", id, pid[u], act[u], (unsigned long long)pk0[u],
                           (unsigned long long)pk1[u], (unsigned long long)keyops<W>::make(s.w0, s.n0), (unsigned long long)keyops<W>::make(s.w1, s.n1),
                           (unsigned long long)keyops<W>::make(g.w0, g.n0), (unsigned long long)keyops<W>::make(g.w1, g.n1));
            }
#endif
            d.k0[id] = keyops<W>::make(s.w0, s.n0);
            d.k1[id] = keyops<W>::make(s.w1, s.n1);
            d.parent[id] = pid[u];
            d.act[id] = (uint8_t)act[u];
            d.tlen[id] = (uint8_t)(s.n0 + s.n1);
            d.depth[id] = dep[u] + 1;
        }
    }
#endif
}

}  // namespace acx
