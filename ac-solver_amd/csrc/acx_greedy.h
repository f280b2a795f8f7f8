// acx_greedy.h -- device-resident priority frontier for greedy_search (ac_solver/search/greedy.py:15-121).
//
// ONE persistent workgroup (1024 lanes, 16 waves) runs the whole best-first loop of one search on the
// GPU: no host round trip per heap bucket.  The heap of the reference, keyed (total length, depth, state
// tuple) (greedy.py:104-113), is held as BUCKETS keyed (total length, depth):
//
//   * a bucket is a contiguous region of node ids in an arena: [head, sorted_end) is sorted by the signed
//     state tuple (the third heap field), [sorted_end, cnt) are later arrivals; a bucket that is selected
//     with an unsorted tail is sorted in LDS (bitonic network on compare_pres); a bucket that does not fit
//     the LDS is sorted as LDS-sized runs that are then merged by rank (binary searches in the other runs);
//   * the minimum bucket is found from per-length live counts in LDS and a per-length depth bitmap in HBM;
//   * a BATCH is a prefix of the minimum bucket (up to R*1024/12 parents, R children per lane), treated
//     exactly as the batch-per-launch path treats it (acx_search.hip: k_expand / k_lookup / k_insert /
//     k_mark / k_decide / k_commit): expand, success test before dedup, read-only probe of the visited table,
//     in-batch dedup to the minimum tag in an LDS table, scan in tag order, budget / cut decision, commit;
//   * a batch is cut after the first parent that inserts a NEW child shorter than the bucket (that child is
//     the heap's next minimum); the speculative children of later parents are dropped.  The batch
//     size adapts: it doubles while batches run uncut and returns to one pass after a cut.
//
// A single search (acx_search) hands buckets of >= GreedyDev::hand_min queued parents to the whole-GPU kernels of
// acx_greedy_mega.h: the kernel parks what it keeps in LDS (GreedyState), returns GREEDY_HANDOFF and is relaunched afterwards.
// Several searches (acx_search_many) run as concurrent one-workgroup kernels on their own streams.
// When a capacity of this scheme is exceeded (depth >= kDepthCap, arena exhausted) the kernel reports
// GREEDY_FALLBACK and the host reruns the search on the batch-per-launch path.
#pragma once
#include "acx_word.h"

namespace acx {

#ifndef ACX_GREEDY_PROFILE
#define ACX_GREEDY_PROFILE 0  // 1: thread 0 accumulates shader-clock cycles per phase (costs ~10 %); see GreedyOut::t_phase
#endif

#ifndef ACX_GREEDY_THREADS
#define ACX_GREEDY_THREADS 1024
#endif
constexpr int kGT = ACX_GREEDY_THREADS;  // lanes of the persistent workgroup
constexpr uint32_t kDepthCap = 16384;  // bucket table rows per total length

enum : uint32_t { GREEDY_RUNNING = 0, GREEDY_SOLVED = 1, GREEDY_BUDGET = 2, GREEDY_EXHAUSTED = 3, GREEDY_FALLBACK = 4, GREEDY_MOVE_ERROR = 5, GREEDY_HANDOFF = 6,
                  GREEDY_MEGA_MORE = 7 };  // a relaunch that found the whole-GPU kernels still at work on the handed-off bucket: nothing done

// What the persistent workgroup keeps in LDS between batches, parked in HBM when a single search hands a big bucket to the
// whole-GPU kernels of acx_greedy_mega.h (GREEDY_HANDOFF) and read back when it is relaunched (`resume`).
struct GreedyState {
    uint32_t len_count[132], hint[132], hist[32];
    uint32_t arena_top, nodes, status, reason, err, seen_min, np_cap, last_parent, solved_pid, last_child_len, solved_action, max_bucket;
    uint32_t reported;  // the frontier kernel has written the search's final GreedyOut (chained launches behind it return at once)
    uint32_t cur_len, cur_depth, resume, depth_hi;  // depth_hi: the largest depth any node has (buckets of deeper depths still hold the zeroes of the set-up)
    unsigned long long expanded, batches, sorts, big_sorts, mega_batches, mega_parents;
    unsigned long long t_phase[24];  // ACX_GREEDY_PROFILE
};

struct BucketRec {
    uint32_t off;         // first arena entry of the region
    uint32_t cap;         // region size (0: never allocated)
    uint32_t head;        // [head, cnt) are queued
    uint32_t sorted_end;  // [head, sorted_end) is in signed state order
    uint32_t cnt;
    uint32_t pad[3];
};

template <typename W> struct alignas(2 * sizeof(W)) NodeKey {
    W k0, k1;
};
constexpr unsigned long long kTabEmpty = ~0ull;

template <typename W> struct GreedyDev {
    SearchDev<W> d;
    BucketRec* bk;      // [nlen * kDepthCap], zero-initialised
    uint32_t* bitmap;   // [nlen * kDepthCap / 32]: bit set <=> bucket (len, depth) is non-empty
    uint32_t* arena;    // bucket regions
    W* gk0;             // scratch for sorting buckets larger than the LDS: keys and ids in run order
    W* gk1;
    uint32_t* gid;
    NodeKey<W>* nkeys;  // [cap_nodes] packed key of every node, both relators in one 16 / 32-byte record (one request per compare)
    unsigned long long* tab;  // visited table: node id | 32-bit key fingerprint << 32, kTabEmpty when free; tmask = entries - 1.
    uint32_t tmask;           // Probe sequences advance in 16-byte pairs of slots (one memory request per step).
    W root_k0, root_k1; // the initial presentation (node 0)
    uint32_t arena_cap; // entries
    uint32_t nlen;      // 2L + 1 total lengths
    long long max_nodes;
    uint32_t root_len;
    uint32_t nf;         // 1: the root is in normal form (freely -- with `cyclical`, cyclically -- reduced, both relators non-empty), so every node is: the
                         // kernels instantiated for that run the shorter move code apply_move_nf (acx_word.h; what the BFS kernels call kMoveNf)
    uint32_t hand_min;   // 0: never; else a selected bucket with at least this many queued parents ends the kernel with GREEDY_HANDOFF
    GreedyState* state;  // nullable (k_greedy_sched): where the frontier is parked / resumed from
    const uint32_t* mega_status;  // nullable: {status, cut, remaining} of the whole-GPU kernels (acx_greedy_mega.h: MegaScalars).  The host
                                  // relaunches this kernel behind every mega-batch WITHOUT waiting for the batch's outcome (one
                                  // synchronisation per hand-off instead of two); when the bucket is not finished the launch is a no-op
    uint32_t rank_max;   // chained mode: a handed-off bucket of more entries is ordered by the frontier kernel itself first (kMegaRankMax)
    uint32_t* hand_ctl;  // nullable: {pending, queued parents, 1 = unsorted tail} of MegaScalars -- chained mode (acx_greedy_mega.h): the hand-off
                         // is read by the whole-GPU kernels the host has ALREADY enqueued behind this one, not by the host
    // k_greedy_sched: gk0 / gk1 / gid of a slot hold `scratch_cap` entries, enough for the usual bucket that outgrows the LDS; a sort that
    // needs more borrows one of `big_n` full-size regions ([gk0 | gk1 | gid], `big_key_bytes` per key array, `big_stride` bytes apart)
    // shared by all slots of the call: big_lock[r] = 1 while a workgroup sorts in region r.  big_lock == nullptr: gk0 / gk1 / gid are full size.
    uint32_t scratch_cap;
    uint32_t big_n;
    uint32_t* big_lock;
    uint8_t* big_base;
    unsigned long long big_key_bytes, big_stride;
};

struct GreedyOut {
    uint32_t status, nodes, min_len, err;
    uint32_t solved_parent, solved_action, last_parent, last_child_len;
    unsigned long long expanded, batches;
    uint32_t fallback_reason, max_bucket;
    uint32_t path_n, arena_top;  // arena_top: entries of the bucket arena the search used
    uint32_t hand_len, hand_depth, hand_live, hand_sort;  // GREEDY_HANDOFF: the bucket, its queued parents, 1 = it has an unsorted tail
    unsigned long long sorts, big_sorts;
    uint32_t hist_sort[16];  // sorts by log2(bucket size)
    uint32_t hist_np[16];    // batches by log2(parents)
    unsigned long long t_phase[24];  // [12..15] inside commit: stores + ballots, per-length positions, seen + CAS issue, barrier; [16..23] the phases 0..7 over the batches of <= 21 parents only; [8] global probe rounds (cycles), [9] probe rounds of wave 0 (count)  // shader-clock cycles per phase (thread 0): select, sort, expand, probe, scan+decide, commit, file, tail
};

// Handed-off buckets of at most this many entries are ordered by the whole-GPU counting sort of acx_greedy_mega.h (k_gm_rank); in
// chained mode a larger one is ordered by the frontier kernel itself (its bitonic network through HBM) BEFORE it is handed off.
constexpr uint32_t kMegaRankMax = 16384;  // GreedyDev::rank_max unless the option ACX_OPT_MEGA_RANK_MAX says otherwise (tests: a small value sends AK(3) through the own-sort path)

template <typename W> struct greedy_cfg;
// k_greedy_sched (many searches per launch, one per workgroup at a time): lanes and candidates per batch.  Measured on the 1190
// Miller-Schupp searches of 1e6 nodes (tools/ms_sweep_warm.py greedy).  Rounds 2-3, one workgroup per search in static launches:
// 1024 lanes x 4 children 0.40 s, 512-lane workgroups two per compute unit 0.49-0.55 s (a launch waited for its slowest search: latency
// counted).  Round 4, searches as jobs on persistent workgroups: 1024 x 4 0.172-0.176 s.  Round 5, the same with 512 lanes, TWO
// workgroups per compute unit (128 registers each, waves-per-SIMD bound 4) and 512 slots: 0.152-0.158 s with 1, 2 or 4 children per
// lane -- 2 has the fewest spills (35 registers against 73-88) and the smallest LDS (40 / 72 KB); 512 lanes alone on a compute unit, no
// spills: 0.177 s; 256 lanes: 0.178 s.  A search does not need the lanes, the chip needs the searches in flight.
#ifndef ACX_GREEDY_MULTI_THREADS
#define ACX_GREEDY_MULTI_THREADS 512
#endif
constexpr uint32_t kGreedyMultiThreads = ACX_GREEDY_MULTI_THREADS;
#ifndef ACX_GREEDY_MULTI_R
#define ACX_GREEDY_MULTI_R 2
#endif
constexpr uint32_t kGreedyPerCu = 2;  // workgroups of k_greedy_sched a compute unit holds (512 lanes, 128 registers each)
template <> struct greedy_cfg<uint64_t> { static constexpr uint32_t kSortCap = ACX_GREEDY_MULTI_R * kGreedyMultiThreads; };
template <> struct greedy_cfg<u128> { static constexpr uint32_t kSortCap = 2 * kGreedyMultiThreads; };
template <> struct greedy_cfg<u128x> : greedy_cfg<u128> {};  // (max_relator_length 62 .. 64, acx_keys.h: the same 16-byte keys)

// The hash of the greedy frontier's tables (visited table: slot = low bits, 32-bit fingerprint = high half; in-batch tables: low bits /
// bits 40..): one multiply-xorshift round per key word, as the BFS stamp tables use (acx_bfs.h: stamp_hash).  Rounds 1-3 used
// acx_frontier.h's hash_key -- five 64-bit multiplies, fifteen quarter-rate instructions on the critical path of every batch.
ACX_HD uint64_t greedy_mix(uint64_t h, uint64_t w) {
    h = (h ^ w) * 0x9E3779B97F4A7C15ull;
    return h ^ (h >> 29);
}
ACX_HD uint64_t greedy_hash(uint64_t k0, uint64_t k1) { return greedy_mix(greedy_mix(0, k0), k1); }
ACX_HD uint64_t greedy_hash(u128 k0, u128 k1) {
    const uint64_t h = greedy_mix(greedy_mix(0, (uint64_t)k0), (uint64_t)(k0 >> 64));
    return greedy_mix(greedy_mix(h, (uint64_t)k1), (uint64_t)(k1 >> 64));
}

template <typename W> __device__ __forceinline__ bool key_less(W a0, W a1, W b0, W b1) {
    Pres<W> a, b;
    key_to_pres<W>(a0, a1, a);
    key_to_pres<W>(b0, b1, b);
    return compare_pres<W>(a, b) < 0;
}

// compare-exchange of LDS entries i < x: afterwards entry i precedes entry x in signed state order when `asc`
// (padding ids 0xFFFFFFFF are the largest elements)
template <typename W> __device__ __forceinline__ void lds_cmpx(W* sk0, W* sk1, uint32_t* sid, uint32_t i, uint32_t x, bool asc) {
    const uint32_t ia = sid[i], ib = sid[x];
    bool a_after_b;
    if (ia == 0xFFFFFFFFu) a_after_b = ib != 0xFFFFFFFFu;
    else if (ib == 0xFFFFFFFFu) a_after_b = false;
    else a_after_b = key_less<W>(sk0[x], sk1[x], sk0[i], sk1[i]);
    if (a_after_b == asc) {
        const W t0 = sk0[i], t1 = sk1[i];
        sk0[i] = sk0[x];
        sk1[i] = sk1[x];
        sk0[x] = t0;
        sk1[x] = t1;
        sid[i] = ib;
        sid[x] = ia;
    }
}

// Bitonic sort of the n entries (sid, sk0, sk1)[0..n) in LDS by signed state order; whole workgroup.
// `descending`: direction of the final merge (a chunk of a larger bitonic network is sorted against its neighbour).
template <typename W, uint32_t kGT> __device__ __forceinline__ void lds_sort(W* sk0, W* sk1, uint32_t* sid, uint32_t n, uint32_t tid, bool descending = false) {
    uint32_t P = 2;
    while (P < n) P <<= 1;
    for (uint32_t i = n + tid; i < P; i += kGT) sid[i] = 0xFFFFFFFFu;  // padding sorts last
    __syncthreads();
    for (uint32_t k = 2; k <= P; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t q = tid; q < P / 2; q += kGT) {
                const uint32_t i = ((q & ~(j - 1)) << 1) | (q & (j - 1));  // the q-th index with bit j clear
                const bool asc = k == P ? !descending : (i & k) == 0;
                lds_cmpx<W>(sk0, sk1, sid, i, i | j, asc);
            }
            __syncthreads();
        }
    }
}

// The merge stages j = P/2 .. 1 of a bitonic network on the P (power of two) LDS entries, all in one direction.
template <typename W, uint32_t kGT> __device__ __forceinline__ void lds_merge(W* sk0, W* sk1, uint32_t* sid, uint32_t P, uint32_t tid, bool asc) {
    for (uint32_t j = P >> 1; j > 0; j >>= 1) {
        for (uint32_t q = tid; q < P / 2; q += kGT) {
            const uint32_t i = ((q & ~(j - 1)) << 1) | (q & (j - 1));
            lds_cmpx<W>(sk0, sk1, sid, i, i | j, asc);
        }
        __syncthreads();
    }
}

// Workgroup barrier that orders LDS traffic only: outstanding global stores / atomics stay in flight.  Global data written
// before it is NOT guaranteed visible to the other waves afterwards.  (With ROCm 7.2's compiler __syncthreads() on gfx950 is
// the same two instructions -- no barrier of this file's kernels is preceded by a vmcnt wait: the waves of a workgroup share
// a compute unit and its L1 -- so this spelling only pins down what the commit phase relies on; round 5 checked the assembly.)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// exclusive prefix of up to 64 wave counts by one wave; returns the grand total (valid in every lane)
__device__ __forceinline__ uint32_t wave_excl_scan(uint32_t v, uint32_t lane, uint32_t& total) {
    uint32_t x = v;
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t y = (uint32_t)__shfl_up((int)x, o);
        if ((int)lane >= o) x += y;
    }
    total = (uint32_t)__shfl((int)x, 63);
    return x - v;
}

// The whole search, executed by one 1024-lane workgroup.  path_act / path_len (nullable, `path_cap` entries): the
// reference's return path, (-1, len0), (action, length) ... , written by lane 0 at the end; out->path_n is its length.
template <typename W, uint32_t SC, uint32_t kGT, bool NF>  // SC candidates per batch, kGT lanes (shadows the namespace constant: the single search's workgroup), NF: normal-form search
__device__ __forceinline__ void greedy_run(const GreedyDev<W>& g, GreedyOut* __restrict__ out, int32_t* __restrict__ path_act,
                                           int32_t* __restrict__ path_len, long long path_cap, uint32_t* depth_hi_out = nullptr) {
    constexpr int R = (int)(SC / kGT);                     // children per lane in a full batch
    constexpr uint32_t kPmax = (uint32_t)(R * kGT) / 12u;  // parents in a full batch
    constexpr uint32_t kBT = 2 * SC;                       // in-batch dedup table (LDS)
    constexpr uint32_t kNW = (uint32_t)R * (kGT / 64);     // wave-level winner counts per batch (<= 64)
    constexpr uint32_t kDirectPos = 192;                   // batches with at most this many new states take their bucket positions with one LDS atomic per lane
    __shared__ W sk0[SC];  // sort keys; double as the candidate keys of a batch (indexed by tag)
    __shared__ W sk1[SC];
    __shared__ uint32_t sid[SC];
    __shared__ uint32_t s_btab[kBT];
    __shared__ uint8_t s_clen[SC];
    __shared__ W sp_k0[kPmax + 3];  // parents of the running batch
    __shared__ W sp_k1[kPmax + 3];
    __shared__ uint32_t sp_id[kPmax + 3];
    __shared__ uint32_t s_lcnt[132], s_len_count[132], s_hint[132], s_pushbase[132];
    __shared__ uint32_t s_job[132 * 3];
    __shared__ uint32_t s_wcnt[64], s_woff[64];
    __shared__ BucketRec s_rec;
    __shared__ BucketRec s_frec[132];  // records of the buckets (length, s_fdepth): the depth the running batches file into
    // Fresh buckets (round 4, batches of one candidate per lane): a bucket that was EMPTY before the last batch filed into it consists
    // of that batch's children, and those are still in LDS (candidate keys sk0 / sk1 by tag, their ids and bucket positions in
    // s_keep).  When such a bucket is selected next -- the usual case behind a cut, and behind a bucket that one batch used up -- it
    // is ordered and staged from there: no arena load, no key load (two dependent trips of the sort, two of the staging).
#ifndef ACX_GREEDY_FRESH_MULTI
#define ACX_GREEDY_FRESH_MULTI 0  // 1: the many-search kernel (four candidates per lane) orders fresh buckets from LDS too -- measured on the 1190-search sweep: 0.42-0.44 s with it, 0.38-0.43 s without (16 KB more LDS, more registers in the selection), so only the single search does
#endif
    constexpr bool kFresh = R == 1 || ACX_GREEDY_FRESH_MULTI != 0;
    __shared__ uint32_t s_keep[kFresh ? SC : 1];  // per candidate of the last batch: bit 31 committed, bits 12..23 position in its bucket, bits 0..11 rank among the batch's new nodes
    static_assert(SC <= 4096, "positions and ranks of a batch's candidates take 12 bits each in s_keep");
    __shared__ uint32_t s_keep_n;                  // candidates of the last batch that s_keep / sk0 / sk1 / s_clen describe (0: none)
    __shared__ uint8_t s_fresh[132];               // bucket (length, filing depth) held nothing before the last batch
    __shared__ uint32_t s_keep_nodes;              // id of the last batch's first new node
    __shared__ uint32_t s_fdepth;
    __shared__ uint32_t s_minlen, s_mindepth, s_solved, s_shorter, s_lo, s_err, s_seen_min, s_njobs, s_arena_top, s_committed, s_flag, s_total;
    __shared__ uint32_t s_p_end, s_cutoff, s_budget_hit, s_is_solved, s_last_parent, s_solved_pid, s_cur_len, s_cur_depth, s_nodes, s_status,
        s_reason, s_max_bucket, s_np_cap, s_last_child_len, s_solved_action, s_sorted_in_lds, s_depth_hi;
    __shared__ unsigned long long s_expanded, s_batches, s_sorts, s_big_sorts, s_err_tag;
    __shared__ uint32_t s_hist[32];
    __shared__ unsigned long long s_tph[24], s_tc;  // phase clock, kept by thread 0
    __shared__ uint32_t s_small;  // (profile) the running bucket holds <= 21 parents
    __shared__ uint32_t s_big;    // the region of the shared sort scratch this workgroup holds
    (void)s_small;
#if ACX_GREEDY_PROFILE
#define ACX_TICK(k) do { if (threadIdx.x == 0) { const unsigned long long now__ = clock64(); s_tph[k] += now__ - s_tc; if ((k) < 8 && s_small) s_tph[16 + (k)] += now__ - s_tc; s_tc = now__; } } while (0)
#define ACX_SUBTICK(k) do { if (threadIdx.x == 0) { const unsigned long long now__ = clock64(); s_tph[k] += now__ - s_tc2; s_tc2 = now__; } } while (0)
#else
#define ACX_TICK(k) do { } while (0)
#define ACX_SUBTICK(k) do { } while (0)
#endif

    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63, wv = tid >> 6;
    if (g.hand_ctl && g.state->resume && g.state->reported) return;  // chained launches behind the end of the search
    if (g.mega_status && g.mega_status[0] == GREEDY_RUNNING && g.mega_status[1] == 0 && g.mega_status[2] != 0) {  // more mega-batches to come
        if (tid == 0) out->status = GREEDY_MEGA_MORE;
        return;
    }

    const SearchDev<W>& d = g.d;
    const uint32_t nlen = g.nlen;
    const uint32_t gmask = g.tmask & ~1u;  // probe sequences start on a 2-slot (16-byte) pair

    if (tid < 132) {
        s_len_count[tid] = 0;
        s_hint[tid] = kDepthCap;
    }
    if (tid < 32) s_hist[tid] = 0;
    if (tid < 132) s_fresh[tid] = 0;
    if (tid == 0) s_keep_n = 0;
    if (tid < 24) s_tph[tid] = 0;
    if (tid == 0) s_small = 0;
#if ACX_GREEDY_PROFILE
    if (tid == 0) s_tc = clock64();
#else
    (void)s_tc;
#endif
    __syncthreads();
    const bool resume = g.state != nullptr && g.state->resume != 0;
    if (resume) {  // relaunched after a hand-off: the frontier as it was parked (and as the whole-GPU kernels left it)
        const GreedyState* ps = g.state;
        if (tid < 132) {
            s_len_count[tid] = ps->len_count[tid];
            s_hint[tid] = ps->hint[tid];
        }
        if (tid < 32) s_hist[tid] = ps->hist[tid];
        if (tid < 24) s_tph[tid] = ps->t_phase[tid];
        if (tid == 0) {
            s_arena_top = ps->arena_top;
            s_nodes = ps->nodes;
            s_status = ps->status;
            s_reason = ps->reason;
            s_err = ps->err;
            s_seen_min = ps->seen_min;
            s_expanded = ps->expanded;
            s_batches = ps->batches;
            s_sorts = ps->sorts;
            s_big_sorts = ps->big_sorts;
            s_max_bucket = ps->max_bucket;
            s_last_parent = ps->last_parent;
            s_solved_pid = ps->solved_pid;
            s_last_child_len = ps->last_child_len;
            s_solved_action = ps->solved_action;
            s_np_cap = ps->np_cap;
            s_depth_hi = ps->depth_hi;
            s_flag = 0;
            s_sorted_in_lds = 0;
            s_fdepth = 0xFFFFFFFFu;
        }
    } else if (tid == 0) {
        // root: node 0, first entry of the visited table and of the heap (bucket (root_len, 0))
        g.nkeys[0].k0 = g.root_k0;
        g.nkeys[0].k1 = g.root_k1;
        d.parent[0] = kEmpty;
        d.act[0] = 0xff;
        d.tlen[0] = (uint8_t)g.root_len;
        d.depth[0] = 0;
        const uint64_t h = greedy_hash(g.root_k0, g.root_k1);
        g.tab[(uint32_t)h & gmask] = (h >> 32) << 32;  // node 0 with its fingerprint
        BucketRec r = {0u, 16u, 0u, 1u, 1u, {0u, 0u, 0u}};
        g.arena[0] = 0;
        g.bk[(size_t)g.root_len * kDepthCap] = r;
        g.bitmap[(size_t)g.root_len * (kDepthCap / 32)] = 1u;
        s_len_count[g.root_len] = 1;
        s_hint[g.root_len] = 0;
        s_arena_top = 16;
        s_nodes = 1;
        s_status = GREEDY_RUNNING;
        s_reason = 0;
        s_err = 0;
        s_seen_min = g.root_len;
        s_expanded = 0;
        s_batches = 0;
        s_sorts = 0;
        s_big_sorts = 0;
        s_max_bucket = 1;
        s_last_parent = 0;
        s_solved_pid = 0;
        s_last_child_len = 0;
        s_solved_action = 0;
        s_np_cap = kGT / 12;
        s_depth_hi = 0;
        s_flag = 0;  // 1: the current bucket record in s_rec is valid
        s_sorted_in_lds = 0;
        s_fdepth = 0xFFFFFFFFu;
    }
    __syncthreads();

    while (s_status == GREEDY_RUNNING) {  // (a resumed search may have ended in the whole-GPU kernels)
        // ================================================================= select the minimum bucket ====
        if (!s_flag) {
            if (tid == 0) {
                s_minlen = 0xFFFFFFFFu;
                s_mindepth = 0xFFFFFFFFu;
            }
            __syncthreads();
            if (tid < nlen && s_len_count[tid] > 0) atomicMin(&s_minlen, tid);
            __syncthreads();
            const uint32_t l = s_minlen;
            if (l == 0xFFFFFFFFu) {  // heap exhausted (greedy.py:71)
                if (tid == 0) s_status = GREEDY_EXHAUSTED;
                __syncthreads();
                break;
            }
            // Shortcut (round 4): when the bucket of length l at the CACHED depth holds every queued entry of that length, it is the
            // minimum-depth bucket of l -- no bitmap load, no record load: the two dependent trips of this phase (2.1 us of a
            // 10 us small batch).  That is the common case: a batch that was cut by a new shorter child is followed by the bucket
            // it has just created, and a bucket that was used up by the one its children went to.
            const uint32_t fd0 = s_fdepth;
            const bool whole = fd0 != 0xFFFFFFFFu && s_frec[l].cnt - s_frec[l].head == s_len_count[l];  // (uniform: nobody writes these here)
#if ACX_GREEDY_PROFILE
            if (tid == 0) {
                s_hist[28]++;
                if (whole) s_hist[29]++;
                if (kFresh && whole && s_fresh[l] != 0) s_hist[30]++;
            }
#endif
            if (!whole) {
                const uint32_t w = s_hint[l] / 32 + tid;
                if (w < kDepthCap / 32) {
                    const uint32_t bits = g.bitmap[(size_t)l * (kDepthCap / 32) + w];
                    if (bits) atomicMin(&s_mindepth, w * 32 + (uint32_t)__builtin_ctz(bits));
                }
                __syncthreads();
            }
            {
                // the selected bucket's record comes from the LDS cache when it is a bucket of the cached depth; the cache
                // then moves to depth + 1 (the children's depth): old records out, new records in, one trip together -- and no
                // trip at all for a depth that no node has reached yet (its records are still the zeroes of the set-up)
                const uint32_t D = whole ? fd0 : s_mindepth, fd = fd0;
                BucketRec cur, nxt;
                if (tid == 0) cur = D == fd ? s_frec[l] : g.bk[(size_t)l * kDepthCap + D];
                const bool move = fd != D + 1 && D + 1 < kDepthCap && tid < nlen;
                if (move) {
                    if (D + 1 > s_depth_hi) nxt = BucketRec{0u, 0u, 0u, 0u, 0u, {0u, 0u, 0u}};
                    else nxt = g.bk[(size_t)tid * kDepthCap + D + 1];
                    if (fd != 0xFFFFFFFFu) g.bk[(size_t)tid * kDepthCap + fd] = s_frec[tid];
                }
                __syncthreads();  // (tid 0 read s_frec[l] before it is replaced)
                if (move) s_frec[tid] = nxt;
                if (tid == 0) {
                    s_cur_len = l;
                    s_cur_depth = D;
                    s_hint[l] = D;
                    s_rec = cur;
                    s_flag = 1;
                    s_sorted_in_lds = 0;
                    if (D + 1 < kDepthCap) s_fdepth = D + 1;
                }
            }
            __syncthreads();
            const uint32_t n = s_rec.cnt - s_rec.head;
#if ACX_GREEDY_PROFILE
            if (tid == 0) s_small = n <= 21u ? 1u : 0u;
            __syncthreads();
#endif
            ACX_TICK(0);
            const bool hand = g.hand_min && n >= g.hand_min;  // (uniform)
            const bool own_sort = hand && g.hand_ctl && n > g.rank_max && s_rec.sorted_end < s_rec.cnt;  // chained: too large for the counting sort
            if (hand && !own_sort) {
                // a big bucket: park the frontier; the host runs this bucket on the whole GPU and relaunches this kernel
                // (every bucket record goes back to HBM: the selected one is there already, the cached depth follows)
                if (tid < nlen && s_fdepth != 0xFFFFFFFFu) g.bk[(size_t)tid * kDepthCap + s_fdepth] = s_frec[tid];
                if (tid == 0) s_status = GREEDY_HANDOFF;
                __syncthreads();
                break;
            }
            const bool fresh_sel = whole && s_keep_n != 0 && s_fresh[l] != 0;  // (uniform) the bucket is made of the last batch's children
            // ---- a fresh bucket: its entries are the last batch's children, still in LDS ---------------------------------
            const bool fast = kFresh && fresh_sel && n <= SC;  // (uniform)
            if (fast) {
                uint32_t kp[R];
                W key0[R], key1[R];
#pragma unroll
                for (int r = 0; r < R; r++) {  // candidate t = r * lanes + tid of the last batch
                    const uint32_t t = (uint32_t)r * kGT + tid;
                    kp[r] = t < s_keep_n ? s_keep[kFresh ? t : 0] : 0u;
                    if ((uint32_t)s_clen[t] != l) kp[r] = 0;
                    key0[r] = key1[r] = 0;
                    if (kp[r] >> 31) {
                        key0[r] = sk0[t];
                        key1[r] = sk1[t];
                    }
                }
                __syncthreads();  // every candidate key is read before the sort arrays overwrite them
#pragma unroll
                for (int r = 0; r < R; r++)
                    if (kp[r] >> 31) {
                        const uint32_t at = (kp[r] >> 12) & 0xFFFu;
                        sk0[at] = key0[r];
                        sk1[at] = key1[r];
                        sid[at] = s_keep_nodes + (kp[r] & 0xFFFu);
                    }
                if (tid == 0 && n == 1) s_sorted_in_lds = 1;  // nothing to order: the staging below takes it from here
                __syncthreads();
            }
            // ---- order the bucket by the signed state tuple if it has an unsorted tail ----------------------
            if (s_rec.sorted_end < s_rec.cnt && n > 1) {
                const uint32_t base = s_rec.off + s_rec.head;
                if (tid == 0) {
                    s_hist[min(15, 31 - __builtin_clz(n))]++;
                    s_sorts++;
                    if (n > SC) s_big_sorts++;
                }
                if (n <= 256) {
                    // all-pairs rank sort: 1024 / n2 lanes share the comparisons of one element (n2 = n rounded up to a power of two)
                    uint32_t myid = 0;
                    W m0 = 0, m1 = 0;
                    if (tid < n) {
                        if (fast) {
                            myid = sid[tid];
                            m0 = sk0[tid];
                            m1 = sk1[tid];
                        } else {
                            myid = g.arena[base + tid];
                            const NodeKey<W> nk = g.nkeys[myid];
                            m0 = nk.k0;
                            m1 = nk.k1;
                            sk0[tid] = m0;
                            sk1[tid] = m1;
                        }
                        s_btab[tid] = 0;  // rank
                    }
                    __syncthreads();
                    uint32_t n2 = 2;
                    while (n2 < n) n2 <<= 1;
                    const uint32_t parts = kGT / n2, i = tid / parts, part = tid - i * parts;
                    if (i < n) {
                        const W e0 = sk0[i], e1 = sk1[i];
                        uint32_t c = 0;
                        for (uint32_t j = part; j < n; j += parts) c += key_less<W>(sk0[j], sk1[j], e0, e1) ? 1u : 0u;
                        if (c) atomicAdd(&s_btab[i], c);
                    }
                    __syncthreads();
                    uint32_t rank = 0;
                    if (tid < n) rank = s_btab[tid];
                    __syncthreads();
                    if (tid < n) {
                        sid[rank] = myid;
                        sk0[rank] = m0;
                        sk1[rank] = m1;
                        g.arena[base + rank] = myid;
                    }
                    if (tid == 0) s_sorted_in_lds = 1;
                } else if (n <= SC) {
                    if (!fast)
                        for (uint32_t i = tid; i < n; i += kGT) {
                            const uint32_t id = g.arena[base + i];
                            sid[i] = id;
                            const NodeKey<W> nk = g.nkeys[id];
                            sk0[i] = nk.k0;
                            sk1[i] = nk.k1;
                        }
                    lds_sort<W, kGT>(sk0, sk1, sid, n, tid);
                    for (uint32_t i = tid; i < n; i += kGT) g.arena[base + i] = sid[i];
                    if (tid == 0) s_sorted_in_lds = 1;
                } else {
                    // bucket larger than the LDS: one bitonic network over P = 2^k >= n entries (padding sorts last).  Chunks of
                    // SC entries are sorted / merged in LDS; only the compare-exchange stages whose stride reaches SC stream
                    // through the scratch arrays in HBM (coalesced, a few hundred KB per stage).
                    uint32_t P = SC;
                    while (P < n) P <<= 1;
                    W* q0 = g.gk0;
                    W* q1 = g.gk1;
                    uint32_t* qi = g.gid;
                    const bool borrow = g.big_lock != nullptr && P > g.scratch_cap;  // (uniform)
                    if (borrow) {  // a full-size region of the call's shared pool: held for this one sort
                        if (tid == 0) {
                            uint32_t r = blockIdx.x % g.big_n;
                            while (atomicCAS(&g.big_lock[r], 0u, 1u) != 0u) {
                                r = r + 1 == g.big_n ? 0u : r + 1;
                                __builtin_amdgcn_s_sleep(32);
                            }
                            s_big = r;
                        }
                        __syncthreads();
                        __threadfence();  // (acquire: nothing of the region's last holder in this compute unit's L1)
                        uint8_t* b = g.big_base + (size_t)s_big * g.big_stride;
                        q0 = (W*)b;
                        q1 = (W*)(b + g.big_key_bytes);
                        qi = (uint32_t*)(b + 2 * g.big_key_bytes);
                    }
                    for (uint32_t c0 = 0; c0 < P; c0 += SC) {
                        for (uint32_t i = tid; i < SC; i += kGT) {
                            const uint32_t gi = c0 + i;
                            if (gi < n) {
                                const uint32_t id = g.arena[base + gi];
                                const NodeKey<W> nk = g.nkeys[id];
                                sid[i] = id;
                                sk0[i] = nk.k0;
                                sk1[i] = nk.k1;
                            } else {
                                sid[i] = 0xFFFFFFFFu;
                            }
                        }
                        lds_sort<W, kGT>(sk0, sk1, sid, SC, tid, ((c0 / SC) & 1u) != 0);
                        for (uint32_t i = tid; i < SC; i += kGT) {
                            qi[c0 + i] = sid[i];
                            q0[c0 + i] = sk0[i];
                            q1[c0 + i] = sk1[i];
                        }
                        __syncthreads();
                    }
                    for (uint32_t k = 2 * SC; k <= P; k <<= 1) {
                        for (uint32_t j = k >> 1; j >= SC; j >>= 1) {  // strides that span chunks: through HBM
                            for (uint32_t q = tid; q < P / 2; q += kGT) {
                                const uint32_t i = ((q & ~(j - 1)) << 1) | (q & (j - 1)), x = i | j;
                                const bool asc = (i & k) == 0;
                                const uint32_t ia = qi[i], ib = qi[x];
                                const W a0 = q0[i], a1 = q1[i], b0 = q0[x], b1 = q1[x];
                                bool a_after_b;
                                if (ia == 0xFFFFFFFFu) a_after_b = ib != 0xFFFFFFFFu;
                                else if (ib == 0xFFFFFFFFu) a_after_b = false;
                                else a_after_b = key_less<W>(b0, b1, a0, a1);
                                if (a_after_b == asc) {
                                    qi[i] = ib;
                                    q0[i] = b0;
                                    q1[i] = b1;
                                    qi[x] = ia;
                                    q0[x] = a0;
                                    q1[x] = a1;
                                }
                            }
                            __syncthreads();
                        }
                        for (uint32_t c0 = 0; c0 < P; c0 += SC) {  // the remaining strides stay inside a chunk: in LDS
                            for (uint32_t i = tid; i < SC; i += kGT) {
                                sid[i] = qi[c0 + i];
                                sk0[i] = q0[c0 + i];
                                sk1[i] = q1[c0 + i];
                            }
                            __syncthreads();
                            lds_merge<W, kGT>(sk0, sk1, sid, SC, tid, (c0 & k) == 0);
                            for (uint32_t i = tid; i < SC; i += kGT) {
                                qi[c0 + i] = sid[i];
                                q0[c0 + i] = sk0[i];
                                q1[c0 + i] = sk1[i];
                            }
                            __syncthreads();
                        }
                    }
                    for (uint32_t i = tid; i < n; i += kGT) g.arena[base + i] = qi[i];
                    if (borrow) {  // every access to the region is through before it is another workgroup's
                        __threadfence();
                        __syncthreads();
                        if (tid == 0) atomicExch(&g.big_lock[s_big], 0u);
                    }
                }
                __syncthreads();
            }
            if (tid == 0) {
                s_rec.sorted_end = s_rec.cnt;
                if (n > s_max_bucket) s_max_bucket = n;
            }
            __syncthreads();
#if ACX_GREEDY_PROFILE
            if (tid == 0) {  // sort cycles by bucket-size class: [10] n > SC, [11] 256 < n <= SC (the rest is n <= 256)
                const unsigned long long now = clock64();
                if (n > SC) s_tph[10] += now - s_tc;
                else if (n > 256) s_tph[11] += now - s_tc;
            }
#endif
            ACX_TICK(1);
            if (own_sort) {  // ordered here: now it goes to the whole-GPU kernels (its record, with the new sorted_end, back to HBM first)
                if (tid < nlen && s_fdepth != 0xFFFFFFFFu) g.bk[(size_t)tid * kDepthCap + s_fdepth] = s_frec[tid];
                if (tid == 0) {
                    g.bk[(size_t)s_cur_len * kDepthCap + s_cur_depth] = s_rec;
                    s_status = GREEDY_HANDOFF;
                }
                __syncthreads();
                break;
            }
        }

        // ================================================================= one batch =====================
        const uint32_t cur_len = s_cur_len, cur_depth = s_cur_depth;
        const uint32_t D1 = cur_depth + 1;
        const uint32_t live = s_rec.cnt - s_rec.head;
        const uint32_t np = live < s_np_cap ? live : s_np_cap;
        const uint32_t m = 12u * np;
        const uint32_t nodes = s_nodes;
        uint32_t bt = 256;  // in-batch table size for this batch
        while (bt < 2 * m) bt <<= 1;
        // stage the parents (ids and keys) in LDS: straight from the sorted arrays when the bucket was just sorted there
        {
            uint32_t pi = 0;
            W p0 = 0, p1 = 0;
            if (tid < np) {
                if (s_sorted_in_lds) {
                    pi = sid[tid];
                    p0 = sk0[tid];
                    p1 = sk1[tid];
                } else {
                    pi = g.arena[s_rec.off + s_rec.head + tid];
                    const NodeKey<W> nk = g.nkeys[pi];
                    p0 = nk.k0;
                    p1 = nk.k1;
                }
            }
            __syncthreads();  // the sort arrays are free from here on
            if (tid < np) {
                sp_id[tid] = pi;
                sp_k0[tid] = p0;
                sp_k1[tid] = p1;
            }
        }
        if (tid == 0) {
            s_hist[16 + 31 - __builtin_clz(np)]++;
            s_solved = 0xFFFFFFFFu;
            s_shorter = 0xFFFFFFFFu;
            s_lo = 0xFFFFFFFFu;
            s_njobs = 0;
            s_committed = 0;
            s_sorted_in_lds = 0;
            s_err_tag = ~0ull;
        }
        if (tid < 132) s_lcnt[tid] = 0;
        for (uint32_t i = tid; i < bt; i += kGT) s_btab[i] = kEmpty;
        __syncthreads();
        // per-lane candidates: tag t = r * 1024 + tid  (parent t / 12, action t % 12); keys live in LDS (sk0/sk1[t])
        uint32_t hv[R], hb[R], fl[R], fpv[R];  // fl: bit0 active, bit1 known, bit2 winner, bit3 commit; fpv: 32-bit key fingerprint
#pragma unroll
        for (int r = 0; r < R; r++) {
            const uint32_t t = (uint32_t)r * kGT + tid;
            fl[r] = 0;
            hv[r] = hb[r] = fpv[r] = 0;
            if (t < m) {
                const uint32_t p = t / 12u;
                Pres<W> s;
                key_to_pres<W>(sp_k0[p], sp_k1[p], s);
                const int e = NF ? apply_move_nf<W, kSearchSafeOf<W>>(s, (int)(t - 12u * p), d.L, d.cyclical != 0) : apply_move<W, kSearchSafeOf<W>>(s, (int)(t - 12u * p), d.L, d.cyclical != 0);
                if (e) atomicMin(&s_err_tag, ((unsigned long long)t << 8) | (unsigned long long)e);  // counts only if the reference gets this far
                const W c0 = keyops<W>::make(s.w0, s.n0), c1 = keyops<W>::make(s.w1, s.n1);
                const uint32_t tl = (uint32_t)(s.n0 + s.n1);
                sk0[t] = c0;
                sk1[t] = c1;
                s_clen[t] = (uint8_t)tl;
                const uint64_t h = greedy_hash(c0, c1);
                hv[r] = (uint32_t)h & gmask;
                hb[r] = (uint32_t)h & (bt - 1);
                fl[r] = 1u;
                fpv[r] = (uint32_t)(h >> 32);
                if (tl == 2) atomicMin(&s_solved, t);  // greedy.py:91, before the membership test
            }
        }
        __syncthreads();  // candidate keys visible, tables cleared
        ACX_TICK(2);
        // read-only probe of the visited table: a 16-byte pair of slots (id + 32-bit fingerprint each) per memory request,
        // the lane's R candidates in flight together; a full key is fetched only behind a matching fingerprint
        {
            uint32_t pend = 0;
#pragma unroll
            for (int r = 0; r < R; r++) pend |= (fl[r] & 1u) << r;
            while (pend) {
#if ACX_GREEDY_PROFILE
                if (tid == 0) s_tph[9]++;
#endif
                ulonglong2 sl[R];
#pragma unroll
                for (int r = 0; r < R; r++)
                    if ((pend >> r) & 1u) sl[r] = *(const ulonglong2*)(g.tab + hv[r]);
                uint32_t cand[R];
                NodeKey<W> nk[R];
#pragma unroll
                for (int r = 0; r < R; r++) {
                    cand[r] = kEmpty;
                    if (!((pend >> r) & 1u)) continue;
                    const unsigned long long s2[2] = {sl[r].x, sl[r].y};
                    const uint32_t j0 = hb[r] >> 28;  // sub-slot to resume at (after a fingerprint that belonged to another key)
                    bool stop = false;
#pragma unroll
                    for (int j = 0; j < 2; j++) {
                        if (stop || (uint32_t)j < j0) continue;
                        if (s2[j] == kTabEmpty) {  // first empty slot of the probe sequence: not in the table
                            hv[r] += (uint32_t)j;
                            pend &= ~(1u << r);
                            stop = true;
                        } else if ((uint32_t)(s2[j] >> 32) == fpv[r]) {
                            cand[r] = (uint32_t)s2[j];
                            hb[r] = (hb[r] & 0x0FFFFFFFu) | ((uint32_t)(j + 1) << 28);
                            stop = true;
                        }
                    }
                    if (!stop) {  // pair exhausted
                        hv[r] = (hv[r] + 2) & g.tmask;
                        hb[r] &= 0x0FFFFFFFu;
                    }
                    if (cand[r] != kEmpty) nk[r] = g.nkeys[cand[r]];
                }
#pragma unroll
                for (int r = 0; r < R; r++) {
                    if (cand[r] == kEmpty) continue;
                    const uint32_t t = (uint32_t)r * kGT + tid;
                    if (nk[r].k0 == sk0[t] && nk[r].k1 == sk1[t]) {
                        fl[r] |= 2u;
                        pend &= ~(1u << r);
                    } else if ((hb[r] >> 28) >= 2) {  // false fingerprint match in the pair's last slot
                        hv[r] = (hv[r] + 2) & g.tmask;
                        hb[r] &= 0x0FFFFFFFu;
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < R; r++) hb[r] &= 0x0FFFFFFFu;
        }
        ACX_TICK(8);
        // in-batch dedup: the minimum tag among equal keys wins
#pragma unroll
        for (int r = 0; r < R; r++) {
            if ((fl[r] & 3u) != 1u) continue;
            const uint32_t t = (uint32_t)r * kGT + tid;
            const W c0 = sk0[t], c1 = sk1[t];
            uint32_t h = hb[r];
            for (;;) {
                uint32_t v = s_btab[h];
                if (v == kEmpty) {
                    v = atomicCAS(&s_btab[h], kEmpty, t);
                    if (v == kEmpty) break;
                }
                if (sk0[v] == c0 && sk1[v] == c1) {
                    if (v > t) atomicMin(&s_btab[h], t);
                    break;
                }
                h = (h + 1) & (bt - 1);
            }
            hb[r] = h;
        }
        __syncthreads();
        ACX_TICK(3);
        uint32_t cpos[R];
#pragma unroll
        for (int r = 0; r < R; r++) {
            const uint32_t t = (uint32_t)r * kGT + tid;
            const bool win = (fl[r] & 3u) == 1u && s_btab[hb[r]] == t;
            if (win) fl[r] |= 4u;
            {  // one LDS atomic per wave (tags grow with the lane, so the lowest flagged lane holds the wave's minimum)
                const unsigned long long sb = __ballot(win && (uint32_t)s_clen[t] < cur_len);
                if (sb && lane == (uint32_t)__builtin_ctzll(sb)) atomicMin(&s_shorter, t);
            }
            const unsigned long long bal = __ballot(win);
            cpos[r] = (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
            if (lane == 0) s_wcnt[r * (kGT / 64) + wv] = (uint32_t)__popcll(bal);
        }
        __syncthreads();
        if (wv == 0) {
            uint32_t tot;
            const uint32_t ex = wave_excl_scan(lane < kNW ? s_wcnt[lane] : 0u, lane, tot);
            s_woff[lane] = ex;
            if (lane == 0) s_total = tot;
        }
        __syncthreads();
        const uint32_t total = s_total;
        const bool over = (long long)nodes < g.max_nodes && (long long)nodes + total >= g.max_nodes;
#pragma unroll
        for (int r = 0; r < R; r++) {
            cpos[r] += s_woff[r * (kGT / 64) + wv];
            if (over && (fl[r] & 4u) && (long long)cpos[r] + 1 == g.max_nodes - (long long)nodes) s_lo = (uint32_t)r * kGT + tid;  // the child that reaches the budget
        }
        __syncthreads();
        if (tid == 0) {  // k_decide of the batch-per-launch path
            uint32_t p_end = np - 1, budget_hit = 0;
            if (s_shorter != 0xFFFFFFFFu) p_end = min(p_end, s_shorter / 12u);
            if ((long long)nodes >= g.max_nodes) {
                p_end = 0;
                budget_hit = 1;
            } else if (over) {
                const uint32_t pb = s_lo / 12u;
                if (pb <= p_end) {
                    p_end = pb;
                    budget_hit = 1;
                }
            }
            uint32_t is_solved = s_solved != 0xFFFFFFFFu && s_solved / 12u <= p_end;
            // a move on which the reference's ACMove raises: it does so when the move is executed before the search ends
            if (s_err_tag != ~0ull && (uint32_t)((s_err_tag >> 8) / 12u) <= p_end && !(is_solved && s_solved < (uint32_t)(s_err_tag >> 8))) {
                s_err = (uint32_t)(s_err_tag & 0xffu);
                is_solved = 0;
            }
            s_cutoff = is_solved ? s_solved : 12u * (p_end + 1);
            s_p_end = is_solved ? s_solved / 12u : p_end;
            s_budget_hit = budget_hit;
            s_is_solved = is_solved;
            s_last_parent = sp_id[s_p_end];
            s_last_child_len = s_clen[12u * s_p_end + 11u];  // greedy.py:121
            if (is_solved) {
                s_solved_pid = sp_id[s_solved / 12u];
                s_solved_action = s_solved % 12u;
            }
        }
        __syncthreads();
        ACX_TICK(4);
        const uint32_t cutoff = s_cutoff, p_end = s_p_end;
        const bool is_solved = s_is_solved != 0;
#if ACX_GREEDY_PROFILE
        unsigned long long s_tc2 = clock64();
#endif
        uint32_t pos[R];
        uint32_t seen = 0xFFFFFFFFu;
#pragma unroll
        for (int r = 0; r < R; r++) {
            const uint32_t t = (uint32_t)r * kGT + tid;
            pos[r] = 0;
            if ((fl[r] & 1u) && t < 12u * (p_end + 1)) seen = min(seen, (uint32_t)s_clen[t]);
            const bool cm = (fl[r] & 4u) && t < cutoff;
            uint32_t tl = 0;
            if (cm) {  // k_commit
                fl[r] |= 8u;
                const uint32_t id = nodes + cpos[r], p = t / 12u;
                tl = s_clen[t];
                NodeKey<W> nk;
                nk.k0 = sk0[t];
                nk.k1 = sk1[t];
                g.nkeys[id] = nk;
                d.parent[id] = sp_id[p];
                d.act[id] = (uint8_t)(t - 12u * p);
                d.tlen[id] = (uint8_t)tl;
                d.depth[id] = D1;
            }
            // LDS atomics are aggregated per wave (thousands of lanes on one address would serialise):
            unsigned long long cb = __ballot(cm);
            ACX_SUBTICK(12);
            if (cb && lane == 63u - (uint32_t)__builtin_clzll(cb)) atomicMax(&s_committed, cpos[r] + 1);  // cpos grows with the lane
            if (total <= kDirectPos) {
                // A small batch (most are: 8-15 parents): one returning LDS atomic per committed lane.  The aggregation below is a chain
                // of four dependent LDS operations per distinct length and wave (2300 cycles of a 21 000-cycle batch in the kernel's
                // own clock); a few dozen lanes on a handful of counters serialise for far less.  (The order of a bucket's entries
                // does not matter: a bucket is sorted by state before it is popped.)
                if (cm) pos[r] = atomicAdd(&s_lcnt[tl], 1u);
                cb = 0;
            }
            while (cb) {  // position inside the target bucket: one atomicAdd per (wave, total length)
                const uint32_t lead = (uint32_t)__builtin_ctzll(cb);
                const uint32_t v = (uint32_t)__shfl((int)tl, (int)lead);
                const unsigned long long same = __ballot(cm && tl == v) & cb;
                uint32_t base = 0;
                if (lane == lead) base = atomicAdd(&s_lcnt[v], (uint32_t)__popcll(same));
                base = (uint32_t)__shfl((int)base, (int)lead);
                if (cm && tl == v) pos[r] = base + (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
                cb &= ~same;
            }
            if (kFresh) s_keep[kFresh ? (uint32_t)r * kGT + tid : 0] = cm ? (0x80000000u | (pos[r] << 12) | cpos[r]) : 0u;
        }
        if (tid == 0) {
            s_keep_nodes = nodes;
            s_keep_n = kFresh ? SC : 0u;
        }
        ACX_SUBTICK(13);
        for (int o = 32; o > 0; o >>= 1) seen = min(seen, (uint32_t)__shfl_xor((int)seen, o));
        if (lane == 0 && seen != 0xFFFFFFFFu) atomicMin(&s_seen_min, seen);
        // visited-table insertion: the keys are pairwise distinct and absent; the CAS goes to the empty slot the probe
        // ended on and is only looked at after the filing phase (its round trip overlaps with it)
        unsigned long long cas_old[R];
#pragma unroll
        for (int r = 0; r < R; r++)
            cas_old[r] = (fl[r] & 8u) ? atomicCAS(&g.tab[hv[r]], kTabEmpty, (unsigned long long)(nodes + cpos[r]) | ((unsigned long long)fpv[r] << 32)) : kTabEmpty;
        ACX_SUBTICK(14);
        lds_barrier();
        ACX_SUBTICK(15);
        ACX_TICK(5);
        // ---- file the new nodes into their buckets (total length, depth + 1): one owner lane per length ----
        if (tid < 132) s_fresh[tid] = 0;
        if (tid < nlen && s_lcnt[tid] > 0 && !is_solved) {
            const uint32_t k = s_lcnt[tid];
            if (D1 >= kDepthCap) {
                s_status = GREEDY_FALLBACK;
                s_reason = 1;
            } else {
                BucketRec r = s_frec[tid];
                const uint32_t lv = r.cnt - r.head;
                if (r.cap == 0 || r.cnt + k > r.cap) {
                    uint32_t nc = 16;
                    while (nc < 2 * (lv + k)) nc <<= 1;
                    const uint32_t no = atomicAdd(&s_arena_top, nc);
                    if ((unsigned long long)no + nc > g.arena_cap) {
                        s_status = GREEDY_FALLBACK;
                        s_reason = 2;
                        nc = 0;
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
                s_pushbase[tid] = r.off + r.cnt;
                r.cnt += k;
                if (s_status == GREEDY_RUNNING) {
                    s_frec[tid] = r;
                    if (lv == 0) s_fresh[tid] = 1;
                    if (lv == 0) atomicOr(&g.bitmap[(size_t)tid * (kDepthCap / 32) + D1 / 32], 1u << (D1 & 31));
                    s_len_count[tid] += k;
                    if (D1 < s_hint[tid]) s_hint[tid] = D1;
                }
            }
        }
        lds_barrier();
        ACX_TICK(6);
        if (s_status != GREEDY_RUNNING) break;
        // The value every CAS is compared with, made opaque HERE: without this the compiler compares (and so waits for) the result
        // right behind the atomic -- `global_atomic_cmpswap_x2; s_waitcnt vmcnt(0)` in front of the filing phase, the whole round
        // trip exposed (7 % of the kernel's cycles by its own clock) -- instead of here, behind it.
        unsigned long long tab_empty = kTabEmpty;
        asm volatile("" : "+v"(tab_empty));
#pragma unroll
        for (int r = 0; r < R; r++) {  // settle the table insertions (a failed CAS: another new key took the slot in this batch, rare)
            if (!(fl[r] & 8u) || cas_old[r] == tab_empty) continue;
            const unsigned long long mine = (unsigned long long)(nodes + cpos[r]) | ((unsigned long long)fpv[r] << 32);
            do hv[r] = (hv[r] + 1) & g.tmask;
            while (atomicCAS(&g.tab[hv[r]], kTabEmpty, mine) != kTabEmpty);
        }
        for (uint32_t j = 0; j < s_njobs; j++) {  // grown buckets move to their new region
            const uint32_t src = s_job[3 * j], dst = s_job[3 * j + 1], cnt = s_job[3 * j + 2];
            for (uint32_t i = tid; i < cnt; i += kGT) g.arena[dst + i] = g.arena[src + i];
        }
#pragma unroll
        for (int r = 0; r < R; r++)
            if ((fl[r] & 8u) && !is_solved) g.arena[s_pushbase[s_clen[(uint32_t)r * kGT + tid]] + pos[r]] = nodes + cpos[r];
        if (tid == 0) {
            const uint32_t popped = p_end + 1;
            s_batches++;
            s_expanded += popped;
            s_nodes = nodes + s_committed;
            if (s_committed && !is_solved && D1 > s_depth_hi) s_depth_hi = D1;
            if (s_err) {
                s_status = GREEDY_MOVE_ERROR;
            } else if (is_solved) {
                s_status = GREEDY_SOLVED;
            } else {
                s_rec.head += popped;
                s_len_count[cur_len] -= popped;
                const bool empty = s_rec.head == s_rec.cnt;
                const bool cut = s_shorter != 0xFFFFFFFFu;
                if (empty) {
                    s_rec.head = s_rec.cnt = s_rec.sorted_end = 0;
                    atomicAnd(&g.bitmap[(size_t)cur_len * (kDepthCap / 32) + cur_depth / 32], ~(1u << (cur_depth & 31)));
                }
                if (s_budget_hit) s_status = GREEDY_BUDGET;
                if (empty || cut || s_budget_hit) {
                    if (cur_depth == s_fdepth) s_frec[cur_len] = s_rec;
                    else g.bk[(size_t)cur_len * kDepthCap + cur_depth] = s_rec;
                    s_flag = 0;
                }
                // speculation depth: back to one pass after a cut, doubled after an uncut batch
                s_np_cap = cut ? (uint32_t)(kGT / 12) : min(2 * s_np_cap, kPmax);
            }
        }
        __syncthreads();
        ACX_TICK(7);
        if (s_status != GREEDY_RUNNING) break;
    }
#undef ACX_TICK
    if (g.state) {
        GreedyState* ps = g.state;
        if (tid < 132) {
            ps->len_count[tid] = s_len_count[tid];
            ps->hint[tid] = s_hint[tid];
        }
        if (tid < 32) ps->hist[tid] = s_hist[tid];
        if (tid < 24) ps->t_phase[tid] = s_tph[tid];
        if (tid == 0) {
            ps->arena_top = s_arena_top;
            ps->nodes = s_nodes;
            ps->status = s_status == GREEDY_HANDOFF ? (uint32_t)GREEDY_RUNNING : s_status;
            ps->reason = s_reason;
            ps->err = s_err;
            ps->seen_min = s_seen_min;
            ps->expanded = s_expanded;
            ps->batches = s_batches;
            ps->sorts = s_sorts;
            ps->big_sorts = s_big_sorts;
            ps->max_bucket = s_max_bucket;
            ps->last_parent = s_last_parent;
            ps->solved_pid = s_solved_pid;
            ps->last_child_len = s_last_child_len;
            ps->solved_action = s_solved_action;
            ps->np_cap = s_np_cap;
            ps->depth_hi = s_depth_hi;
            ps->cur_len = s_cur_len;
            ps->cur_depth = s_cur_depth;
            ps->resume = 1;
            ps->reported = s_status != GREEDY_HANDOFF && s_status != GREEDY_RUNNING ? 1u : 0u;
        }
    }
    if (tid == 0 && g.hand_ctl) {
        g.hand_ctl[1] = s_status == GREEDY_HANDOFF ? s_rec.cnt - s_rec.head : 0u;
        g.hand_ctl[2] = s_status == GREEDY_HANDOFF && s_rec.sorted_end < s_rec.cnt ? 1u : 0u;
        g.hand_ctl[0] = s_status == GREEDY_HANDOFF ? 1u : 0u;
    }
    if (tid == 0) {
        out->hand_len = s_cur_len;
        out->hand_depth = s_cur_depth;
        out->hand_live = s_status == GREEDY_HANDOFF ? s_rec.cnt - s_rec.head : 0u;
        out->hand_sort = s_status == GREEDY_HANDOFF && s_rec.sorted_end < s_rec.cnt ? 1u : 0u;
        out->status = s_status;
        out->nodes = s_nodes;
        out->min_len = s_status == GREEDY_SOLVED ? 2u : s_seen_min;
        out->err = s_err;
        out->solved_parent = s_solved_pid;
        out->solved_action = s_solved_action;
        out->last_parent = s_last_parent;
        out->last_child_len = s_last_child_len;
        out->expanded = s_expanded;
        out->batches = s_batches;
        out->fallback_reason = s_reason;
        out->max_bucket = s_max_bucket;
        out->sorts = s_sorts;
        out->big_sorts = s_big_sorts;
        for (int k = 0; k < 24; k++) out->t_phase[k] = s_tph[k];
        for (int k = 0; k < 16; k++) out->hist_sort[k] = s_hist[k], out->hist_np[k] = s_hist[16 + k];
        out->path_n = 0;
        out->arena_top = s_arena_top;
        if (path_act && (s_status == GREEDY_SOLVED || s_status == GREEDY_BUDGET || s_status == GREEDY_EXHAUSTED)) {
            // greedy.py:93 (success) / :121 (failure): path of a popped node + one more (action, length) entry
            const bool ok = s_status == GREEDY_SOLVED;
            uint32_t v = ok ? s_solved_pid : s_last_parent;
            const uint32_t dep = d.depth[v];
            out->path_n = dep + 2;
            if ((long long)dep + 2 <= path_cap) {
                path_act[dep + 1] = ok ? (int32_t)s_solved_action : 11;
                path_len[dep + 1] = ok ? 2 : (int32_t)s_last_child_len;
                for (uint32_t k = dep;; k--) {
                    path_act[k] = d.act[v] == 0xff ? -1 : (int32_t)d.act[v];
                    path_len[k] = d.tlen[v];
                    if (k == 0) break;
                    v = d.parent[v];
                }
            }
        }
    }
    if (depth_hi_out) {  // (k_greedy_sched: the deepest bucket row this search can have written, for the slot's clean-up; every lane its own copy)
        __syncthreads();
        *depth_hi_out = s_depth_hi;
    }
}

// No register pad here (ACX_VGPR_PAD, acx_common.h): a 1024-lane workgroup may use at most 128 registers per lane and the
// frontier needs them all (it already spills); `amdgpu_num_vgpr(120)` does not lower the allocation under this launch bound.
// These two kernels are covered by the repeat-determinism tests instead (tests/test_gpu_determinism.py).
// The single search hands buckets of >= hand_min (1024) parents to acx_greedy_mega.h, so its own batches can be smaller:
// kSingleSortCap = 1024 candidates (one child per lane, 85 parents per batch) needs 36 bytes of scratch per lane where the
// 4096-candidate form of the many-search kernel needs 280, and the latency chain of a small batch is that much shorter
// (AK(3), 1e7 nodes: 234 -> 211 ms although the batches become more).
#ifndef ACX_GREEDY_SINGLE_SC
#define ACX_GREEDY_SINGLE_SC 1024  // 0: as the many-search kernel
#endif
template <typename W> constexpr uint32_t kSingleSortCap = ACX_GREEDY_SINGLE_SC ? (uint32_t)ACX_GREEDY_SINGLE_SC : greedy_cfg<W>::kSortCap;
template <typename W, bool NF>
__global__ void __launch_bounds__(kGT) k_greedy_persistent(GreedyDev<W> g, GreedyOut* __restrict__ out) {
    greedy_run<W, kSingleSortCap<W>, (uint32_t)kGT, NF>(g, out, nullptr, nullptr, 0);
}

// ---- many searches as jobs on persistent workgroups (round 4; the slots a pool since round 5) ------------------------------------------
// Rounds 2-3 gave every search of a launch its own workgroup and its own memory: a launch was as many searches as fit the memory
// budget (46 at 1e6 nodes), a batch of 170 four launches one after the other, and every launch waited for its slowest search while
// the compute units of the finished ones stood idle (5.5e10 workgroup cycles = 103 ms x 256 compute units in a 400 ms sweep).  Here
// the searches are JOBS that persistent workgroups take from a counter, one after the other, and a SLOT is the memory of one search:
// it is cleaned between two jobs by the workgroup itself (the visited table, the rows of the bucket table the last job can have
// touched).  Searches of different max_relator_length share a launch (a job carries its L), so a whole sweep is one launch per
// key width and move code.
template <typename W> struct GreedyJob {
    W root_k0, root_k1;
    uint32_t root_len, nlen;
    int32_t L, pad_;
};
#ifndef ACX_GREEDY_MULTI_WAVES_PER_EU
#define ACX_GREEDY_MULTI_WAVES_PER_EU 4  // the second launch bound: waves per SIMD the register allocation leaves room for -- two 512-lane workgroups per compute unit
#endif

// A slot as the host set it up: visited table free, bucket records and bitmap zero.  Only rows 0 .. hi + 2 of the first `nlen` lengths
// can be anything else after a search whose deepest node was at depth hi.
template <typename W> __device__ __forceinline__ void greedy_slot_clean(const GreedyDev<W>& g, uint32_t hi, uint32_t nlen, uint32_t tid) {
    ulonglong2* t2 = (ulonglong2*)g.tab;
    const uint32_t n2 = (g.tmask + 1u) / 2u;
    for (uint32_t i = tid; i < n2; i += kGreedyMultiThreads) t2[i] = make_ulonglong2(kTabEmpty, kTabEmpty);
    const uint32_t rows = min(hi + 3u, kDepthCap), words = (rows + 31u) / 32u;
    for (uint32_t l = 0; l < nlen; l++) {
        uint4* r4 = (uint4*)(g.bk + (size_t)l * kDepthCap);  // a record is two uint4
        for (uint32_t i = tid; i < 2u * rows; i += kGreedyMultiThreads) r4[i] = make_uint4(0, 0, 0, 0);
        for (uint32_t i = tid; i < words; i += kGreedyMultiThreads) g.bitmap[(size_t)l * (kDepthCap / 32) + i] = 0;
    }
}

// `slot_free`: one bit per slot of the call, set = free.  The slots are a POOL (round 5): a workgroup takes one when it gets its first
// job, keeps it for the jobs that follow (cleaning it in between) and hands it back CLEAN when the jobs are used up -- so the launches
// of the two key widths (their GreedyDev arrays describe the same memory) need no shares fixed beforehand: both are launched with as
// many workgroups as they have jobs, the chip holds 512 of them, and whichever launch runs out of jobs first leaves its compute units
// -- and its slots -- to the waiting workgroups of the other.
template <typename W, bool NF>
__global__ void __launch_bounds__(kGreedyMultiThreads, ACX_GREEDY_MULTI_WAVES_PER_EU) k_greedy_sched(const GreedyDev<W>* __restrict__ slots, uint32_t* __restrict__ slot_free, uint32_t slot_words,
                                                      const GreedyJob<W>* __restrict__ jobs, uint32_t n_jobs,
                                                      uint32_t* __restrict__ counter, GreedyOut* __restrict__ outs, int32_t* __restrict__ path_act,
                                                      int32_t* __restrict__ path_len, long long path_cap) {
    __shared__ uint32_t s_job, s_slot;
    const uint32_t tid = threadIdx.x;
    uint32_t slot = 0xFFFFFFFFu, used = 0, prev_hi = 0, prev_nlen = 0;
    for (;;) {
        // A workgroup takes its SLOT first and a job only once it holds one (round 6: it used to take the job first and then spin for a
        // slot -- a workgroup without a slot sat on a compute unit holding a search that could not start).  With fewer slots than resident
        // workgroups (ACX_OPT_GREEDY_SLOTS, little free memory) the ones that find no slot leave as soon as the job counter is exhausted.
        if (slot == 0xFFFFFFFFu) {
            __syncthreads();
            if (tid == 0) {
                uint32_t w = blockIdx.x % slot_words, found = 0xFFFFFFFFu;
                for (;;) {
                    const uint32_t bits = atomicOr(&slot_free[w], 0u);
                    if (bits) {
                        const uint32_t b = 1u << (uint32_t)__builtin_ctz(bits);
                        if (atomicAnd(&slot_free[w], ~b) & b) {
                            found = w * 32u + (uint32_t)__builtin_ctz(bits);
                            break;
                        }
                    } else {
                        w = w + 1 == slot_words ? 0u : w + 1;
                        if (w == blockIdx.x % slot_words) {  // once around without a free slot: more workgroups resident than slots
                            if (atomicOr(counter, 0u) >= n_jobs) break;  // nothing left to start: do not wait for a slot
                            __builtin_amdgcn_s_sleep(64);
                        }
                    }
                }
                s_slot = found;
            }
            __syncthreads();
            slot = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_slot);
            if (slot == 0xFFFFFFFFu) break;
        }
        __syncthreads();
        if (tid == 0) s_job = atomicAdd(counter, 1u);
        __syncthreads();
        const uint32_t j = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_job);  // (uniform: the slot and the job arrive through scalar loads)
        if (j >= n_jobs) break;
        GreedyDev<W> g = slots[slot];
        const uint32_t nlen_cap = g.nlen;  // the slot's bucket table has rows for this many total lengths
        const GreedyJob<W> jb = jobs[j];
        if (used) greedy_slot_clean<W>(g, prev_hi, prev_nlen, tid);
        // the vector L1 may hold lines of the slot as its LAST job left them (a kernel starts with an empty L1, a job does not -- and the slot's
        // last user may have been a workgroup on another compute unit): an agent-scope fence writes the clean-up through and invalidates them
        __threadfence();
        g.root_k0 = jb.root_k0;
        g.root_k1 = jb.root_k1;
        g.root_len = jb.root_len;
        g.nlen = min(jb.nlen, nlen_cap);
        g.d.L = jb.L;
        uint32_t hi = 0;
        const unsigned long long wall0 = __builtin_amdgcn_s_memrealtime();  // (the chip's 100 MHz counter: where the job sits in the launch, for ACX_DEBUG's timeline)
        greedy_run<W, greedy_cfg<W>::kSortCap, kGreedyMultiThreads, NF>(g, outs + j, path_act + (size_t)j * path_cap, path_len + (size_t)j * path_cap, path_cap, &hi);
        if (tid == 0) {
            outs[j].t_phase[21] = blockIdx.x;
            outs[j].t_phase[22] = wall0;
            outs[j].t_phase[23] = __builtin_amdgcn_s_memrealtime();
        }
        used = 1;
        prev_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)hi);
        prev_nlen = g.nlen;
    }
    if (slot != 0xFFFFFFFFu) {  // back into the pool, clean
        const GreedyDev<W> g = slots[slot];
        if (used) greedy_slot_clean<W>(g, prev_hi, prev_nlen, tid);
        __threadfence();
        __syncthreads();
        if (tid == 0) atomicOr(&slot_free[slot >> 5], 1u << (slot & 31u));
    }
}

}  // namespace acx
