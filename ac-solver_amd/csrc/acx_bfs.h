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
//   * one 8-byte slot per state:  epoch(8) | fingerprint(20) | parent node id(32) | action(4) (slot_make below; rounds 2-4: a
//     28-bit fingerprint, all ones = free).  A slot names the
//     state as "child `action` of node `parent`" (action 15: the node itself, used for the root), so the full key of an
//     occupant is RECOMPUTED from the parent's key in the node arena (two 8-byte loads that hit in L2 / Infinity Cache
//     far more often than a table entry does, plus one apply_move) -- and only when the fingerprint matches.  Exact set
//     semantics are kept (full-key compare before "seen"); a new state costs ONE CAS and no store.
//   * the batch is told apart by the parent id: parents of the running batch are the nodes >= pbegin, everything below
//     is a committed state.  (parent, action) is also the reference's generation order, so the minimum-tag fold among
//     equal keys is a 64-bit atomicMin on the slot, as before.  No per-batch epochs, no commit pass.  (The epoch of round 5 is per
//     SEARCH: a slot that carries another search's epoch is free, so a table is handed from search to search without a refill.)
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

// A stamp: epoch(8) | fingerprint(20, the top bits of the key's hash) | parent node id(32) | action(4).  The EPOCH is the search's
// (SearchDev::epoch, 1 .. 254): a slot whose epoch field differs is FREE -- what an earlier search left in the table's memory
// (acx_frontier.h: StampBuf hands a table of a finished search to the next one with the next epoch, no refill), or the 0xFF.. of a
// fresh fill.  A free slot is claimed by a CAS on the value that was seen there; stamps of one search share their top eight bits, so
// equal keys still fold to the smaller (parent, action) with a 64-bit atomicMin.
ACX_HD unsigned long long slot_make(uint64_t hk, uint32_t pid, uint32_t act, uint32_t epoch) {
    return ((unsigned long long)epoch << 56) | ((hk >> 44) << 36) | ((unsigned long long)pid << 4) | act;
}
ACX_HD bool slot_free(unsigned long long s, uint32_t epoch) { return (uint32_t)(s >> 56) != epoch; }
ACX_HD uint32_t slot_parent(unsigned long long s) { return (uint32_t)(s >> 4); }
ACX_HD uint32_t slot_action(unsigned long long s) { return (uint32_t)s & 15u; }

// The move code of the BFS kernels.  MODE 0: the general ACMove (apply_move) -- needed when the ROOT is not in normal form
// (unreduced input: its children can even have an empty relator, utils.py:261-278).  With a root in normal form every node
// of the search is, and the kernels run the shorter apply_move_nf: MODE 1 for cyclical = False, MODE 2 for cyclical = True.
enum : int { kMoveGeneral = 0, kMoveNf = 1, kMoveNfCyclical = 2 };
template <typename W, int MODE> __device__ __forceinline__ int search_move(Pres<W>& s, int a, int L, bool cyclical) {
    if (MODE == kMoveNf) return apply_move_nf<W, kSearchSafeOf<W>>(s, a, L, false);
    if (MODE == kMoveNfCyclical) return apply_move_nf<W, kSearchSafeOf<W>>(s, a, L, true);
    return apply_move<W, kSearchSafeOf<W>>(s, a, L, cyclical);
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
    d.stab[(uint32_t)hk & d.stmask & ~3u] = slot_make(hk, 0u, kSelfAction, d.epoch);  // first slot of its bucket
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


// (the kernels' bodies are functions of the tile index `bx`: k_bfs_*_many run them for MANY searches in one launch, a search per blockIdx.y)
template <typename W, int MODE>
__device__ __forceinline__ void bfs_expand_insert_body(const SearchDev<W>& d, uint32_t pbegin, uint32_t np, const BfsCursor* __restrict__ cur, const uint32_t bx) {
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
    if (bx * kBfsParents >= np) return;  // (a full-size grid over a short batch)
    for (uint32_t i = tid; i < (uint32_t)kBfsFold; i += kBfsThreads) s_slot[i] = kEmpty;
    if (tid < kBfsTile / 4) s_took[tid] = 0;
    const uint32_t p = bx * kBfsParents + l, pid = pbegin + p;
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
    // (Tried: the first turn of a lane's three winners together -- their buckets loaded before the first is looked at, their
    // claims issued before the first result is used, three times the accesses in flight per lane.  10.0 ms instead of 9.3 at
    // 1e8 nodes: the memory-side atomic units are saturated as it is, a deeper queue is only more latency.  Fewer resident
    // workgroups are no better: two per compute unit instead of three 10.5 ms, one 15.4 ms.)
#pragma unroll
    for (int it = 0; it < kBfsItems; it++) {
        if (!(probe[it] && s_slot[ls[it]] == me[it])) continue;
        const uint32_t a = w * kBfsItems + it;
        const unsigned long long mine = slot_make(hk[it], pid, a, d.epoch);
        uint32_t base = (uint32_t)hk[it] & d.stmask & ~3u, probes = 0, took = 0;
        bool open = true;
        while (open) {
            const ulonglong2 lo = *(const ulonglong2*)(d.stab + base), hi = *(const ulonglong2*)(d.stab + base + 2);
            const unsigned long long v0 = lo.x, v1 = lo.y, v2 = hi.x, v3 = hi.y;
            auto hot = [&](unsigned long long v) { return slot_free(v, d.epoch) || (v >> 36) == (mine >> 36); };  // free, or my epoch + fingerprint
            uint32_t cand = (hot(v0) ? 1u : 0u) | (hot(v1) ? 2u : 0u) | (hot(v2) ? 4u : 0u) | (hot(v3) ? 8u : 0u);
            while (cand) {
                const uint32_t jj = (uint32_t)__builtin_ctz(cand);
                cand &= cand - 1;
                unsigned long long st = jj == 0 ? v0 : (jj == 1 ? v1 : (jj == 2 ? v2 : v3));
                unsigned long long* slot = d.stab + base + jj;
                if (slot_free(st, d.epoch)) {  // claim it: whoever changes the slot first writes a stamp of THIS epoch, so a failed CAS returns one
                    const unsigned long long seen = st;
                    st = atomicCAS(slot, seen, mine);
                    if (st == seen) {
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
        const uint32_t m = 12u * np, t0 = bx * (uint32_t)kBfsTile;
        if (tid < kBfsTile / 4 && t0 + 4u * tid < m) ((uint32_t*)(d.btook + t0))[tid] = s_took[tid];  // (m is a multiple of 4; t0 of 1536)
    }
}

template <typename W, int MODE>
__global__ void __launch_bounds__(kBfsThreads) k_bfs_expand_insert(SearchDev<W> d, uint32_t pbegin, uint32_t np, const BfsCursor* __restrict__ cur = nullptr) {
    bfs_expand_insert_body<W, MODE>(d, pbegin, np, cur, blockIdx.x);
}

// Winners -> nodes, in TWO launches since round 3:
//   k_bfs_count    a tile of 8192 candidates per workgroup: the took / replaced flags -> one winner bit per candidate (32 per lane,
//                  kept in `masks`) and the tile's winner count; zeroes the replaced-flags it read
//   k_bfs_compact  position of a tile's first winner = sum of the counts of ALL earlier tiles (256 at a time, independent loads),
//                  then the nodes in tag order, the winners' keys recomputed from their parents.
// Rounds 1-2 did this in one launch (a ticket per tile, decoupled look-back over status words).  Its
// own clock (-DACX_COMPACT_PROFILE) showed where the 58 us per 2^20-parent batch went: the ~1500 same-address returning
// atomics of the tickets take 18 us, and every memory access of the kernel crawls while they last (a tile that had its ticket
// after 1 us saw its 32 KB of flags after 20 us); then 8 us of look-back, then 22 us of node writes at ~3 TB/s.
#ifdef ACX_COMPACT_PROFILE
#define ACX_CP_DECL unsigned long long tp[4] = {}, tc0 = clock64()
#define ACX_CP_TICK(k) do { tp[k] = clock64() - tc0; } while (0)
#else
#define ACX_CP_DECL do { } while (0)
#define ACX_CP_TICK(k) do { } while (0)
#endif

template <typename W>
__device__ __forceinline__ void bfs_count_body(const SearchDev<W>& d, uint32_t m, uint32_t* __restrict__ counts, uint32_t* __restrict__ masks, const BfsCursor* __restrict__ cur, const uint32_t bx) {
    __shared__ uint32_t s_wsum[4];
    ACX_VGPR_PAD("v31");
    static_assert(kCompactItems == 32, "two 16-candidate halves per lane");
    if (cur) {  // run-ahead mode (m arrives as the capacity 12 * bmax)
        if (cur->status) return;
        const uint32_t np = cur->nodes - cur->head;  // clamp in PARENTS (as cursor_begin and k_bfs_expand_insert do): 12 x a frontier above 3.6e8 nodes wraps 32 bits
        if (np < m / 12u) m = 12u * np;
    }
    const uint32_t tile = bx, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if (tile >= (m + kCompactTile - 1) / kCompactTile) return;  // a full-size grid over a short batch
    // A lane takes 16 consecutive candidates of each half of the tile (bits 0..15 / 16..31 of fl): its flags are two 16-byte loads
    // per array and the wave reads 1 KB per instruction.
    uint32_t fl = 0;  // bit i < 16: candidate 16 tid + i of the tile; bit 16 + i: candidate kCompactTile / 2 + 16 tid + i -- a winner (took its slot and was not replaced)
#pragma unroll
    for (uint32_t h = 0; h < 2; h++) {
        const uint32_t c0 = tile * kCompactTile + h * (kCompactTile / 2) + 16u * tid;
        if (c0 + 16u <= m) {
            const uint4 tb = *(const uint4*)(d.btook + c0), rb = *(const uint4*)(d.brepl + c0);
            if (rb.x | rb.y | rb.z | rb.w) *(uint4*)(d.brepl + c0) = make_uint4(0, 0, 0, 0);  // zero again for the next batch (no memset launch per batch)
            auto squeeze = [](uint32_t v) { return (v & 1u) | ((v >> 7) & 2u) | ((v >> 14) & 4u) | ((v >> 21) & 8u); };  // bytes are 0 / 1
            fl |= (squeeze(tb.x & ~rb.x) | (squeeze(tb.y & ~rb.y) << 4) | (squeeze(tb.z & ~rb.z) << 8) | (squeeze(tb.w & ~rb.w) << 12)) << (16u * h);
        } else {
            for (uint32_t i = 0; i < 16u; i++)
                if (c0 + i < m) {
                    const uint8_t rb = d.brepl[c0 + i];
                    if (rb) d.brepl[c0 + i] = 0;
                    if (d.btook[c0 + i] && !rb) fl |= 1u << (16u * h + i);
                }
        }
    }
    masks[tile * 256u + tid] = fl;
    uint32_t cnt = (uint32_t)__popc(fl & 0xFFFFu) | ((uint32_t)__popc(fl >> 16) << 16);  // both halves in one word (<= 4096 each)
    for (int o = 32; o > 0; o >>= 1) cnt += (uint32_t)__shfl_xor((int)cnt, o);
    if (lane == 0) s_wsum[wave] = cnt;
    __syncthreads();
    if (tid == 0) counts[tile] = s_wsum[0] + s_wsum[1] + s_wsum[2] + s_wsum[3];
}

template <typename W>
__global__ void __launch_bounds__(256) k_bfs_count(SearchDev<W> d, uint32_t m, uint32_t* __restrict__ counts, uint32_t* __restrict__ masks, const BfsCursor* __restrict__ cur = nullptr) {
    bfs_count_body<W>(d, m, counts, masks, cur, blockIdx.x);
}

template <typename W, int MODE>
__device__ __forceinline__ void bfs_compact_body(const SearchDev<W>& d, uint32_t pbegin, uint32_t m, uint32_t base, uint32_t cap_nodes, const uint32_t* __restrict__ counts,
                                                 const uint32_t* __restrict__ masks, uint32_t* __restrict__ total_out, const BfsCursor* __restrict__ cur, const uint32_t bx) {
    __shared__ uint32_t s_wsum[4], s_psum[4];
    __shared__ uint16_t s_list[kCompactTile];
    // the tile's parents (its 8192 candidates belong to at most 684 consecutive parents), loaded once, coalesced: every parent has
    // ~3.6 winners, and a gather of its key per winner (rounds 2-3) read it that many times
    constexpr uint32_t kTileParents = kCompactTile / 12 + 2;
    __shared__ W s_pk0[kTileParents];
    __shared__ W s_pk1[kTileParents];
    __shared__ uint32_t s_pdep[kTileParents];
    ACX_VGPR_PAD_W(W, "v47", "v63");
    if (cur) {  // run-ahead mode (m arrives as the capacity 12 * bmax)
        if (cur->status) return;
        pbegin = cur->head;
        const uint32_t np = cur->nodes - pbegin;  // clamp in PARENTS: 12 x a frontier above 3.6e8 nodes wraps 32 bits
        if (np < m / 12u) m = 12u * np;
        base = cur->nodes;
    }
    const uint32_t ntiles = (m + kCompactTile - 1) / kCompactTile, tile = bx;
    if (tile >= ntiles) return;  // a full-size grid over a short batch
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    ACX_CP_DECL;
    const uint32_t fl = masks[tile * 256u + tid];
    const uint32_t p_first = (tile * kCompactTile) / 12u;
    {
        const uint32_t m_end = min(m, (tile + 1) * kCompactTile), p_end = (m_end + 11u) / 12u;  // parents p_first .. p_end - 1 touch this tile
        for (uint32_t i = tid; p_first + i < p_end; i += 256) {
            const uint32_t pid = pbegin + p_first + i;
            s_pk0[i] = d.k0[pid];
            s_pk1[i] = d.k1[pid];
            s_pdep[i] = d.depth[pid];
        }
    }
    uint32_t part = 0;  // winners of all earlier tiles: their counts, 256 at a time
    for (uint32_t j = tid; j < tile; j += 256) {
        const uint32_t c = counts[j];
        part += (c & 0xFFFFu) + (c >> 16);
    }
    for (int o = 32; o > 0; o >>= 1) part += (uint32_t)__shfl_xor((int)part, o);
    // positions in tag order: the first halves of all lanes, then the second halves; both scans in one (counts <= 4096 fit 16 bits)
    const uint32_t cnt = (uint32_t)__popc(fl & 0xFFFFu) | ((uint32_t)__popc(fl >> 16) << 16);
    uint32_t incl = cnt;
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = (uint32_t)__shfl_up((int)incl, o);
        if (lane >= (uint32_t)o) incl += v;
    }
    if (lane == 63) s_wsum[wave] = incl;
    if (lane == 0) s_psum[wave] = part;
    __syncthreads();
    ACX_CP_TICK(0);
    uint32_t wbase = 0;
    for (uint32_t w = 0; w < wave; w++) wbase += s_wsum[w];
    const uint32_t both_total = s_wsum[0] + s_wsum[1] + s_wsum[2] + s_wsum[3];
    const uint32_t first_total = both_total & 0xFFFFu, block_total = first_total + (both_total >> 16);
    const uint32_t excl = s_psum[0] + s_psum[1] + s_psum[2] + s_psum[3];
    if (tid == 0 && tile == ntiles - 1) *total_out = excl + block_total;
    {
        const uint32_t before = wbase + incl - cnt;
        uint32_t pos = before & 0xFFFFu, f = fl & 0xFFFFu;
        while (f) {
            const uint32_t i = (uint32_t)__builtin_ctz(f);
            f &= f - 1;
            s_list[pos++] = (uint16_t)(16u * tid + i);
        }
        pos = first_total + (before >> 16);
        f = fl >> 16;
        while (f) {
            const uint32_t i = (uint32_t)__builtin_ctz(f);
            f &= f - 1;
            s_list[pos++] = (uint16_t)(kCompactTile / 2 + 16u * tid + i);
        }
    }
    __syncthreads();
    ACX_CP_TICK(1);
    const uint32_t first_id = base + excl, tbase = tile * kCompactTile;
    // kU winners per lane and round: their parent keys and depths are loaded together, then the moves, then the stores
    // (kU = 4 -- one memory round trip for four nodes -- measured no faster than 1: the writes run at ~3 TB/s either way)
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
            pk0[u] = on[u] ? s_pk0[p - p_first] : (W)0;
            pk1[u] = on[u] ? s_pk1[p - p_first] : (W)0;
            dep[u] = on[u] ? s_pdep[p - p_first] : 0u;
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
                (void)apply_move<W, kSearchSafeOf<W>>(g, (int)act[u], d.L, d.cyclical != 0);
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
#ifdef ACX_COMPACT_PROFILE
    ACX_CP_TICK(2);
    if (tid == 0 && ntiles > 1000 && (tile % 300u) == 7u)
        printf("[compact] tile %u of %u: winners %u; cycles to: prefix + scan %llu, list %llu, nodes written %llu\n", tile, ntiles, block_total, tp[0], tp[1], tp[2]);
#endif
}
template <typename W, int MODE>
__global__ void __launch_bounds__(256) k_bfs_compact(SearchDev<W> d, uint32_t pbegin, uint32_t m, uint32_t base, uint32_t cap_nodes, const uint32_t* __restrict__ counts,
                                                     const uint32_t* __restrict__ masks, uint32_t* __restrict__ total_out, const BfsCursor* __restrict__ cur = nullptr) {
    bfs_compact_body<W, MODE>(d, pbegin, m, base, cap_nodes, counts, masks, total_out, cur, blockIdx.x);
}

#undef ACX_CP_DECL
#undef ACX_CP_TICK

}  // namespace acx
