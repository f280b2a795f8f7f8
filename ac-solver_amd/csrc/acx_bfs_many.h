// Many independent breadth-first searches, level-synchronous in ONE sequence of launches (round 4).
//
// acx_search_many(bfs) used to give every search one persistent workgroup (rounds 2-3, k_bfs_multi): 170 searches of a Miller-Schupp batch
// fill 170 of the 256 compute units with ONE workgroup each, and every search then runs at the latency of its own atomics.  Here the
// searches of a group share the kernels of the fused single search (acx_bfs.h: expand + dedup, count, compact, decide), launched over a
// 2-D grid: blockIdx.x = tile of the search's batch, blockIdx.y = search.  Every search has its own arenas and its own device-resident
// BfsCursor (acx_frontier.h), so a round of four launches advances EVERY running search by one batch; a search that has ended (its
// cursor's status is non-zero) costs its workgroups one load.  The host never reads a decision: it enqueues rounds and looks at the
// status words of the group two rounds late, then finishes each search from its cursor exactly as run_search finishes a single one.
// Same batches per search as the fused single search with the same batch size would form; the results (solved, path, nodes, expanded,
// min_len) do not depend on the batch size (tests/test_gpu_search.py), only `levels` (= batches) does.
#pragma once
#include "acx_bfs.h"

namespace acx {

template <typename W> struct BfsMany {
    SearchDev<W> d;
    BfsCursor* cur;
    uint32_t* counts;  // winners per compact tile ...
    uint32_t* masks;   // ... and one winner bit per candidate
    uint32_t* total;
    Decision* dec;
};

template <typename W> __global__ void k_bfs_root_many(const BfsMany<W>* __restrict__ q, const W* __restrict__ roots, uint32_t n) {
    ACX_VGPR_PAD("v15");
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const SearchDev<W>& d = q[s].d;
    const W k0 = roots[2 * s], k1 = roots[2 * s + 1];
    const uint32_t tl = (uint32_t)(keyops<W>::len(k0) + keyops<W>::len(k1));
    d.k0[0] = k0;
    d.k1[0] = k1;
    d.parent[0] = kEmpty;
    d.act[0] = 0xff;
    d.tlen[0] = (uint8_t)tl;
    d.depth[0] = 0;
    const uint64_t hk = stamp_hash(k0, k1);
    d.stab[(uint32_t)hk & d.stmask & ~3u] = slot_make(hk, 0u, kSelfAction, d.epoch);
    *d.solved_tag = kNoTag;
    *d.shorter_tag = kNoTag;
    *d.err_tag = kNoTag;
    *d.err = 0;
    *d.min_len = 0xffffffffu;
    *q[s].total = 0;
    BfsCursor* c = q[s].cur;
    c->head = 0;
    c->nodes = 1;
    c->status = 0;
    c->batches = 0;
    c->expanded = 0;
    c->min_len = tl;
}

template <typename W, int MODE> __global__ void __launch_bounds__(kBfsThreads) k_bfs_expand_insert_many(const BfsMany<W>* __restrict__ q, uint32_t bmax) {
    const BfsMany<W>& s = q[blockIdx.y];
    bfs_expand_insert_body<W, MODE>(s.d, 0u, bmax, s.cur, blockIdx.x);
}

template <typename W> __global__ void __launch_bounds__(256) k_bfs_count_many(const BfsMany<W>* __restrict__ q, uint32_t mcap) {
    const BfsMany<W>& s = q[blockIdx.y];
    bfs_count_body<W>(s.d, mcap, s.counts, s.masks, s.cur, blockIdx.x);
}

template <typename W, int MODE> __global__ void __launch_bounds__(256) k_bfs_compact_many(const BfsMany<W>* __restrict__ q, uint32_t mcap, uint32_t cap_nodes) {
    const BfsMany<W>& s = q[blockIdx.y];
    bfs_compact_body<W, MODE>(s.d, 0u, mcap, 0u, cap_nodes, s.counts, s.masks, s.total, s.cur, blockIdx.x);
}

// one lane per search; status_out[s] = the cursor's status as this round leaves it (this round's own slot: written here only, read by
// the host after the round's event -- never the live cursors, which the next round may be changing)
template <typename W>
__global__ void k_decide_tab_many(const BfsMany<W>* __restrict__ q, uint32_t n, uint32_t mcap, uint32_t bmax, uint32_t cap_nodes, long long max_nodes,
                                  uint32_t* __restrict__ status_out) {
    ACX_VGPR_PAD("v23");
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const BfsMany<W>& e = q[s];
    decide_tab_body<W>(e.d, mcap, bmax, 0u, 0u, cap_nodes, max_nodes, e.total, e.dec, 1, e.cur, nullptr);
    status_out[s] = e.cur->status;
}

// paths of the solved searches: node want[s] (kEmpty = none) from the root, root first; out_n[s] = depth + 1
template <typename W>
__global__ void k_paths_many(const BfsMany<W>* __restrict__ q, uint32_t n, const uint32_t* __restrict__ want, int32_t* __restrict__ out_act, int32_t* __restrict__ out_len,
                             uint32_t* __restrict__ out_n, long long cap) {
    ACX_VGPR_PAD("v15");
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const uint32_t id = want[s];
    if (id == kEmpty) {
        out_n[s] = 0;
        return;
    }
    const SearchDev<W>& d = q[s].d;
    const uint32_t dep = d.depth[id];
    out_n[s] = dep + 1;
    int32_t* oa = out_act + (long long)s * cap;
    int32_t* ol = out_len + (long long)s * cap;
    for (uint32_t v = id, k = dep;; k--) {
        if ((long long)k < cap) {
            oa[k] = d.act[v] == 0xff ? -1 : (int32_t)d.act[v];
            ol[k] = d.tlen[v];
        }
        if (k == 0) break;
        v = d.parent[v];
    }
}

}  // namespace acx
