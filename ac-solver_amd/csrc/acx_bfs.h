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
//   * children equal to their parent or to their GRANDPARENT are dropped before any probe (both are visited by
//     construction), and a 1024-candidate workgroup first folds its own duplicates in an LDS table: only the smallest
//     tag of each key inside the tile goes to the global table.
//
// Visibility across the eight non-coherent L2s: as in round 1, only the returned values of the device-scope CAS /
// atomicMin decide; a plain (possibly stale) load of a slot can only show an older state of the same slot, which the
// atomic then corrects.  Keys of occupants come from the node arena written by earlier kernels.
#pragma once
#include "acx_frontier.h"

namespace acx {

constexpr unsigned long long kSlotFree = ~0ull;
constexpr uint32_t kSelfAction = 15u;  // slot names the node itself (the root)
constexpr int kBfsBlock = 1024;        // lanes = candidates per workgroup tile (85 parents)
constexpr int kBfsLdsSlots = 2048;

// Every kernel that inlines apply_move declares at least 32 VGPRs (see DESIGN.md "24-VGPR hazard"): an empty asm that
// names v31 as clobbered raises the kernel descriptor's count without costing an instruction or occupancy.
#define ACX_MIN_VGPRS_32() asm volatile("" ::: "v31")

ACX_HD unsigned long long slot_make(uint64_t hk, uint32_t pid, uint32_t act) { return (hk & ~((1ull << 36) - 1)) | ((unsigned long long)pid << 4) | act; }
ACX_HD uint32_t slot_parent(unsigned long long s) { return (uint32_t)(s >> 4); }
ACX_HD uint32_t slot_action(unsigned long long s) { return (uint32_t)s & 15u; }

// key of the state a slot names: child `act` of node `pid` (or the node itself)
template <typename W> __device__ __forceinline__ void slot_key(const SearchDev<W>& d, uint32_t pid, uint32_t act, W& q0, W& q1) {
    q0 = d.k0[pid];
    q1 = d.k1[pid];
    if (act != kSelfAction) {
        Pres<W> s;
        key_to_pres<W>(q0, q1, s);
        (void)apply_move<W, kSearchSafe>(s, (int)act, d.L, d.cyclical != 0);
        q0 = keyops<W>::make(s.w0, s.n0);
        q1 = keyops<W>::make(s.w1, s.n1);
    }
}

template <typename W> __global__ void k_bfs_root(SearchDev<W> d, W k0, W k1, uint32_t tl) {
    d.k0[0] = k0;
    d.k1[0] = k1;
    d.parent[0] = kEmpty;
    d.act[0] = 0xff;
    d.tlen[0] = (uint8_t)tl;
    d.depth[0] = 0;
    const uint64_t hk = hash_key<W>(k0, k1);
    d.stab[(uint32_t)hk & d.stmask] = slot_make(hk, 0u, kSelfAction);
}

// one lane per (parent, action): tag t = 12 * p + a.  btook[t] = 1 when t took its slot (claimed it free, or replaced a
// larger tag of the same key); brepl[tag] = 1 is set for a holder that was replaced (brepl is zero on entry).
template <typename W>
__global__ void __launch_bounds__(kBfsBlock, 8) k_bfs_expand_insert(SearchDev<W> d, uint32_t pbegin, uint32_t np) {
    __shared__ W s_k0[kBfsBlock];
    __shared__ W s_k1[kBfsBlock];
    __shared__ uint32_t s_slot[kBfsLdsSlots];
    ACX_MIN_VGPRS_32();
    const uint32_t tid = threadIdx.x;
    const uint32_t t = blockIdx.x * kBfsBlock + tid;
    const uint32_t m = 12u * np;
    s_slot[tid] = kEmpty;
    s_slot[tid + kBfsBlock] = kEmpty;
    W c0 = 0, c1 = 0;
    uint32_t tl = 0xFFFFFFFFu, pid = 0, a = 0;
    bool probe = false;
    if (t < m) {
        const uint32_t p = t / 12u;
        a = t - 12u * p;
        pid = pbegin + p;
        const W pk0 = d.k0[pid], pk1 = d.k1[pid];
        const uint32_t gp = d.parent[pid];
        Pres<W> s;
        key_to_pres<W>(pk0, pk1, s);
        const int e = apply_move<W, kSearchSafe>(s, (int)a, d.L, d.cyclical != 0);
        // the reference's ACMove raises here -- but only if it gets this far (k_decide_tab): the FIRST such move of the batch counts
        if (e) atomicMin(d.err_tag, ((unsigned long long)t << 8) | (unsigned long long)e);
        c0 = keyops<W>::make(s.w0, s.n0);
        c1 = keyops<W>::make(s.w1, s.n1);
        tl = (uint32_t)(s.n0 + s.n1);
        if (tl == 2) atomicMin(d.solved_tag, (unsigned long long)t);  // breadth_first.py:84: tested before the dedup
        probe = !(c0 == pk0 && c1 == pk1);                            // unchanged state = its (visited) parent
        if (probe && gp != kEmpty) probe = !(d.k0[gp] == c0 && d.k1[gp] == c1);  // back to the (visited) grandparent
    }
    {  // smallest total length of the batch: wave minimum, then one atomic per wave that lowers it
        uint32_t mn = tl;
        for (int o = 32; o > 0; o >>= 1) mn = min(mn, (uint32_t)__shfl_xor((int)mn, o));
        if ((tid & 63u) == 0 && mn < *(volatile uint32_t*)d.min_len) atomicMin(d.min_len, mn);
    }
    s_k0[tid] = c0;
    s_k1[tid] = c1;
    __syncthreads();
    // ---- duplicates inside the tile: LDS table of lane ids, minimum lane (= minimum tag) per key ------------------------
    const uint64_t hk = hash_key<W>(c0, c1);
    uint32_t ls = 0;
    if (probe) {
        ls = (uint32_t)(hk >> 40) & (kBfsLdsSlots - 1);
        for (;;) {
            uint32_t v = s_slot[ls];
            if (v == kEmpty) {
                v = atomicCAS(&s_slot[ls], kEmpty, tid);
                if (v == kEmpty) break;
            }
            if (s_k0[v] == c0 && s_k1[v] == c1) {  // any holder of this slot has my key
                if (v > tid) atomicMin(&s_slot[ls], tid);
                break;
            }
            ls = (ls + 1) & (kBfsLdsSlots - 1);  // at most 1024 of the 2048 slots are ever taken
        }
    }
    __syncthreads();
    uint32_t took = 0;
    if (probe && s_slot[ls] == tid) {
        // ---- the global stamp table ------------------------------------------------------------------------------------
        const unsigned long long me = slot_make(hk, pid, a);
        uint32_t h = (uint32_t)hk & d.stmask, probes = 0;
        for (;;) {
            unsigned long long st = d.stab[h];
            if (st == kSlotFree) {
                st = atomicCAS(&d.stab[h], kSlotFree, me);
                if (st == kSlotFree) {
                    took = 1;
                    break;
                }
            }
            if ((st >> 36) == (me >> 36)) {  // fingerprint match: rebuild the occupant's key
                const uint32_t hp = slot_parent(st), ha = slot_action(st);
                W q0, q1;
                slot_key<W>(d, hp, ha, q0, q1);
                if (q0 == c0 && q1 == c1) {
                    if (hp >= pbegin && ha != kSelfAction && st > me) {  // a candidate of this batch with a larger tag
                        const unsigned long long prev = atomicMin(&d.stab[h], me);
                        if (prev > me) {
                            took = 1;
                            d.brepl[12u * (slot_parent(prev) - pbegin) + slot_action(prev)] = 1;  // no longer the first discoverer
                        }
                    }
                    break;
                }
            }
            h = (h + 1) & d.stmask;
            if (++probes > d.stmask) {
                atomicOr(d.err, kErrTableFull);
                break;
            }
        }
    }
    if (t < m) d.btook[t] = (uint8_t)took;
}

// Winners -> nodes in one pass: k_compact_tab (acx_frontier.h) with the winners' keys recomputed from their parents.
template <typename W>
__global__ void __launch_bounds__(256) k_bfs_compact(SearchDev<W> d, uint32_t pbegin, uint32_t m, uint32_t base, uint32_t cap_nodes, uint32_t epoch,
                                                     unsigned long long* __restrict__ status, uint32_t* __restrict__ ticket, uint32_t* __restrict__ total_out) {
    __shared__ uint32_t s_tile, s_prefix, s_wsum[4];
    __shared__ uint16_t s_list[kCompactTile];
    ACX_MIN_VGPRS_32();
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if (tid == 0) s_tile = atomicAdd(ticket, 1u);
    __syncthreads();
    const uint32_t tile = s_tile, ntiles = (m + kCompactTile - 1) / kCompactTile;
    const uint32_t t0 = tile * kCompactTile + tid * kCompactItems;
    uint32_t fl = 0;  // bit i: candidate t0 + i is a winner (took its slot and was not replaced)
    if (t0 + kCompactItems <= m) {
        const unsigned long long tb = *(const unsigned long long*)(d.btook + t0), rb = *(const unsigned long long*)(d.brepl + t0);
        const unsigned long long w = tb & ~rb;  // bytes are 0 / 1
#pragma unroll
        for (uint32_t i = 0; i < kCompactItems; i++) fl |= (uint32_t)((w >> (8u * i)) & 1ull) << i;
    } else {
        for (uint32_t i = 0; i < kCompactItems; i++)
            if (t0 + i < m && d.btook[t0 + i] && !d.brepl[t0 + i]) fl |= 1u << i;
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
    for (uint32_t j = tid; j < block_total; j += 256) {
        const uint32_t id = first_id + j;
        if (id >= cap_nodes) break;  // beyond the budget: never read
        const uint32_t t = tbase + s_list[j];
        const uint32_t p = t / 12u, pid = pbegin + p, a = t - 12u * p;
        Pres<W> s;
        key_to_pres<W>(d.k0[pid], d.k1[pid], s);
        (void)apply_move<W, kSearchSafe>(s, (int)a, d.L, d.cyclical != 0);
        d.k0[id] = keyops<W>::make(s.w0, s.n0);
        d.k1[id] = keyops<W>::make(s.w1, s.n1);
        d.parent[id] = pid;
        d.act[id] = (uint8_t)a;
        d.tlen[id] = (uint8_t)(s.n0 + s.n1);
        d.depth[id] = d.depth[pid] + 1;
    }
}

}  // namespace acx
