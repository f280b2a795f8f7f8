// acx_search_greedy.hip -- greedy_search (search/greedy.py:15-121) of ONE presentation on the device-resident priority frontier:
// the persistent one-workgroup frontier kernel (acx_greedy.h) and, for buckets of many parents, the whole-GPU batch kernels
// (acx_greedy_mega.h), chained on one stream.  Called by acx_search (acx_search.hip); the batch-per-launch path there takes over when
// the persistent kernel reports that it outgrew one of its capacities.
#include "acx_searcher.h"
#include "acx_bfs.h"
#include "acx_greedy.h"
#include "acx_greedy_mega.h"

namespace acx {

template <typename W> struct AosKeys {  // node keys as the persistent frontier keeps them (node_digest)
    const NodeKey<W>* nk;
    __device__ void operator()(uint32_t i, W& a, W& b) const {
        a = nk[i].k0;
        b = nk[i].k1;
    }
};

// Device buffers of one greedy search on the persistent frontier
template <typename W> struct GreedySearch {
    Searcher<W> S;
    DevBuf bk, bitmap, arena, gk0, gk1, gid, nkeys, tab;
    GreedyDev<W> g;
    // `st`: stream for the bucket-table memsets (nullptr = the search's own stream, S.st)
    int setup(const Pres<W>& root, int L, int64_t max_nodes, int cyclical, hipStream_t st) {
        int rc = S.init(L, cyclical, max_nodes, 1024, false, true);
        if (rc) return rc;
        if (!st) st = S.st;
        g.d = S.d;
        g.nlen = (uint32_t)(2 * L + 1);
        g.max_nodes = (long long)max_nodes;
        g.root_len = (uint32_t)(root.n0 + root.n1);
        g.nf = is_normal_form<W>(root, cyclical != 0) ? 1u : 0u;
        g.hand_min = 0;
        g.state = nullptr;
        g.mega_status = nullptr;
        g.hand_ctl = nullptr;
        g.rank_max = 0;
        const uint64_t arena_entries = std::min<uint64_t>(8ull * (uint64_t)std::max<int64_t>(max_nodes, 1) + (1ull << 20), 1ull << 31);
        g.arena_cap = (uint32_t)arena_entries;
        if (nkeys.alloc(S.cap_nodes * sizeof(NodeKey<W>)) || tab.alloc(S.n_slots * 8)) return ACX_E_NOMEM;
        g.nkeys = (NodeKey<W>*)nkeys.p;
        g.tab = (unsigned long long*)tab.p;
        g.tmask = (uint32_t)(S.n_slots - 1);
        ACX_HIP_TRY(hipMemsetAsync(tab.p, 0xff, S.n_slots * 8, st));
        g.root_k0 = keyops<W>::make(root.w0, root.n0);
        g.root_k1 = keyops<W>::make(root.w1, root.n1);
        const size_t sort_cap = 2 * ((size_t)std::max<int64_t>(max_nodes, 1) + 64) + 4096;  // a bucket (<= all nodes) rounded up to a power of two
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

template <typename W> static void launch_greedy_persistent(const GreedyDev<W>& g, GreedyOut* out, hipStream_t st) {
    if (g.nf) hipLaunchKernelGGL((k_greedy_persistent<W, true>), dim3(1), dim3(kGT), 0, st, g, out);
    else hipLaunchKernelGGL((k_greedy_persistent<W, false>), dim3(1), dim3(kGT), 0, st, g, out);
}

// greedy_search on the device-resident priority frontier (acx_greedy.h).  *handled = false when the persistent
// kernel ran out of one of its capacities: the caller then reruns the search on the batch-per-launch path.
template <typename W>
int run_greedy_device(const Pres<W>& root, int L, int64_t max_nodes, int cyclical, int32_t* solved, int32_t* path_action, int32_t* path_len,
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
    // big buckets go to the whole-GPU kernels of acx_greedy_mega.h (0: the persistent workgroup does everything)
    // (a hand-off cycle costs ~90 us: buckets from 512 parents pay; measured 256 .. 1024: 179.9 / 177.8 / 177.3 / 177.5 / 179.8 ms.
    // ACX_OPT_GREEDY_HAND_MIN / ACX_OPT_MEGA_RANK_MAX: the tests lower both so that small fixtures take the whole-GPU route)
    const uint32_t hand_min = (uint32_t)option(ACX_OPT_GREEDY_HAND_MIN, 512);
    // handed-off buckets up to this size are ordered by counting (k_gm_rank); a larger one is ordered by the frontier kernel itself
    // (bitonic network through HBM) before it is handed off
    const uint32_t rank_max = std::max<uint32_t>(256, (uint32_t)option(ACX_OPT_MEGA_RANK_MAX, kMegaRankMax));
    DevBuf stateb, mck0, mck1, mclen, minfo, midv, mposv, mtab, mscal, mrank;
    MegaDev<W> md;
    GreedyState hstate;
    if (hand_min) {
        if (stateb.alloc(sizeof(GreedyState)) || mck0.alloc((size_t)kMegaTags * sizeof(W)) || mck1.alloc((size_t)kMegaTags * sizeof(W)) || mclen.alloc(kMegaTags) ||
            minfo.alloc((size_t)kMegaTags * 4) || midv.alloc((size_t)kMegaTags * 4) || mposv.alloc((size_t)kMegaTags * 4) || mtab.alloc((size_t)kMegaSlots * 4) ||
            mscal.alloc(sizeof(MegaScalars)) || mrank.alloc((size_t)rank_max * 4))
            return ACX_E_NOMEM;
        ACX_HIP_TRY(hipMemsetAsync(mrank.p, 0, (size_t)rank_max * 4, st));
        ACX_HIP_TRY(hipMemsetAsync(stateb.p, 0, sizeof(GreedyState), st));
        ACX_HIP_TRY(hipMemsetAsync(mscal.p, 0, sizeof(MegaScalars), st));  // (status RUNNING, cut 0, remaining 0: nothing handed off yet)
        g.hand_min = hand_min;
        g.hand_ctl = nullptr;
        g.rank_max = rank_max;
        g.state = (GreedyState*)stateb.p;
        g.mega_status = (const uint32_t*)((const uint8_t*)mscal.p + offsetof(MegaScalars, status));
        md.ck0 = (W*)mck0.p;
        md.ck1 = (W*)mck1.p;
        md.clen = (uint8_t*)mclen.p;
        md.info = (uint32_t*)minfo.p;
        md.idv = (uint32_t*)midv.p;
        md.posv = (uint32_t*)mposv.p;
        md.mtab = (uint32_t*)mtab.p;
        md.rank = (uint32_t*)mrank.p;
        md.sc = (MegaScalars*)mscal.p;
    } else {
        g.hand_min = 0;
        g.state = nullptr;
        g.mega_status = nullptr;
        g.hand_ctl = nullptr;
        g.rank_max = 0;
    }
    // chained (round 4): frontier kernel -> sort -> mega-batch -> frontier kernel ... enqueued back to back with fixed grids; every kernel
    // finds in MegaScalars whether and on what it has to work, the host reads the frontier kernel's status word two cycles late.
    const bool chain = hand_min != 0;
    if (chain) g.hand_ctl = (uint32_t*)((uint8_t*)mscal.p + offsetof(MegaScalars, h_pending));
    static_assert(offsetof(MegaScalars, h_live) == offsetof(MegaScalars, h_pending) + 4 && offsetof(MegaScalars, h_sort) == offsetof(MegaScalars, h_pending) + 8, "pending, live, sort are written as three consecutive words");
    static_assert(offsetof(MegaScalars, cut) == offsetof(MegaScalars, status) + 4 && offsetof(MegaScalars, remaining) == offsetof(MegaScalars, status) + 8, "status, cut, remaining are read as three consecutive words");
    md.g = g;
    EventPair evs;
    ACX_HIP_TRY(evs.create());
    hipEvent_t ev0 = evs.a, ev1 = evs.b;
    ACX_HIP_TRY(hipEventRecord(ev0, st));
    GreedyOut o;
    unsigned long long handoffs = 0;
    launch_greedy_persistent<W>(g, (GreedyOut*)outb.p, st);
    ACX_HIP_TRY(hipGetLastError());
    if (chain) {
        uint32_t* hst = (uint32_t*)S.h_pin;  // pinned: the frontier kernel's status word after every cycle, kRunAheadSlots entries
        for (uint64_t k = 0;; k++) {
            hipLaunchKernelGGL(k_gm_rank<W>, dim3(kRankGrid), dim3(256), 0, st, md, 0u, 1u);
            hipLaunchKernelGGL(k_gm_begin<W>, dim3(kMegaSlots / 1024), dim3(256), 0, st, md, 0u, 0u, 1u);
            hipLaunchKernelGGL(k_gm_expand<W>, dim3(kMegaTags / 256), dim3(256), 0, st, md, 0u, 0u, 1u);
            hipLaunchKernelGGL(k_gm_mark<W>, dim3(kMegaTiles), dim3(kMegaTile), 0, st, md, 0u, 1u);
            hipLaunchKernelGGL(k_gm_decide<W>, dim3(1), dim3(256), 0, st, md, 0u, 0u, 1u);
            hipLaunchKernelGGL(k_gm_commit<W>, dim3(kMegaTiles), dim3(kMegaTile), 0, st, md, 0u, 1u);
            hipLaunchKernelGGL(k_gm_file<W>, dim3(1), dim3(256), 0, st, md, 0u, 1u);
            hipLaunchKernelGGL(k_gm_push<W>, dim3(kMegaTags / 256), dim3(256), 0, st, md, 0u, 1u);
            launch_greedy_persistent<W>(g, (GreedyOut*)outb.p, st);
            ACX_HIP_TRY(hipGetLastError());
            handoffs++;
            const int slot = (int)(k % kRunAheadSlots);
            ACX_HIP_TRY(hipEventRecord(S.ev_batch[slot], st));
            ACX_HIP_TRY(hipStreamWaitEvent(S.st_copy, S.ev_batch[slot], 0));
            ACX_HIP_TRY(hipMemcpyAsync(&hst[slot], (const uint8_t*)outb.p + offsetof(GreedyOut, status), 4, hipMemcpyDeviceToHost, S.st_copy));
            ACX_HIP_TRY(hipEventRecord(S.ev_cursor[slot], S.st_copy));
            if (k >= kRunAheadLag) {
                const int old = (int)((k - kRunAheadLag) % kRunAheadSlots);
                ACX_HIP_TRY(hipEventSynchronize(S.ev_cursor[old]));
                if (hst[old] != GREEDY_HANDOFF && hst[old] != GREEDY_MEGA_MORE) break;  // (the cycles enqueued behind it found nothing to do)
            }
        }
    }
    ACX_HIP_TRY(hipMemcpyAsync(&o, outb.p, sizeof(o), hipMemcpyDeviceToHost, st));
    ACX_HIP_TRY(hipStreamSynchronize(st));
    if (o.status == GREEDY_HANDOFF || o.status == GREEDY_MEGA_MORE) return fail(ACX_E_NODEVICE, "greedy hand-off chain ended in state %u", o.status);
    ACX_HIP_TRY(hipEventRecord(ev1, st));
    ACX_HIP_TRY(hipEventSynchronize(ev1));
    float ms = 0;
    ACX_HIP_TRY(hipEventElapsedTime(&ms, ev0, ev1));
    if (hand_min && g_debug) {
        ACX_HIP_TRY(hipMemcpy(&hstate, stateb.p, sizeof(hstate), hipMemcpyDeviceToHost));
        fprintf(stderr, "[acx_greedy] hand-offs=%llu mega-batches=%llu with %llu parents\n", handoffs, hstate.mega_batches, hstate.mega_parents);
    }
    if (g_debug)
        fprintf(stderr, "[acx_greedy] status=%u nodes=%u batches=%llu expanded=%llu sorts=%llu big_sorts=%llu max_bucket=%u reason=%u %.3f ms\n", o.status, o.nodes,
                o.batches, o.expanded, o.sorts, o.big_sorts, o.max_bucket, o.fallback_reason, ms);
    if (g_debug) {
        fprintf(stderr, "[acx_greedy] sorts by log2(n):");
        for (int k = 0; k < 16; k++) fprintf(stderr, " %u", o.hist_sort[k]);
        fprintf(stderr, "\n[acx_greedy] batches by log2(parents):");
        for (int k = 0; k < 10; k++) fprintf(stderr, " %u", o.hist_np[k]);
        fprintf(stderr, "\n");
        if (o.hist_np[12]) fprintf(stderr, "[acx_greedy] selects %u, of them from the cached depth without a load %u, fresh buckets ordered from LDS %u\n", o.hist_np[12], o.hist_np[13], o.hist_np[14]);
        unsigned long long tot = 0;
        for (int k = 0; k < 8; k++) tot += o.t_phase[k];
        if (tot) fprintf(stderr, "[acx_greedy] sort cycles: %.1f%% of all in buckets > LDS, %.1f%% in 256 < n <= LDS\n", 100.0 * o.t_phase[10] / tot, 100.0 * o.t_phase[11] / tot);
        if (tot) fprintf(stderr, "[acx_greedy] probe: %.1f%% of the cycles in the table rounds, %.2f rounds per batch (wave 0)\n", 100.0 * o.t_phase[8] / (tot + o.t_phase[8]),
                         (double)o.t_phase[9] / (double)o.batches);
        if (tot) {
            unsigned long long ts = 0;
            for (int k = 16; k < 24; k++) ts += o.t_phase[k];
            fprintf(stderr, "[acx_greedy] buckets of <= 21 parents: %.1f%% of all cycles; their cycles%%: select %.1f sort %.1f expand %.1f probe %.1f scan %.1f commit %.1f file %.1f tail %.1f\n",
                    100.0 * ts / tot, 100.0 * o.t_phase[16] / ts, 100.0 * o.t_phase[17] / ts, 100.0 * o.t_phase[18] / ts, 100.0 * o.t_phase[19] / ts, 100.0 * o.t_phase[20] / ts,
                    100.0 * o.t_phase[21] / ts, 100.0 * o.t_phase[22] / ts, 100.0 * o.t_phase[23] / ts);
            fprintf(stderr, "[acx_greedy] inside commit (%% of all cycles): stores + ballots %.1f, per-length positions %.1f, seen + CAS issue %.1f, barrier %.1f\n",
                    100.0 * o.t_phase[12] / tot, 100.0 * o.t_phase[13] / tot, 100.0 * o.t_phase[14] / tot, 100.0 * o.t_phase[15] / tot);
        }
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
    rc = node_digest<W>(AosKeys<W>{g.nkeys}, S.d.parent, S.d.act, o.nodes, st);
    if (rc) return rc;
    if (*path_n > path_cap) return fail(ACX_E_CAPACITY, "path has %lld entries, buffer holds %lld", (long long)*path_n, (long long)path_cap);
    return ACX_OK;
}

template int run_greedy_device<uint64_t>(const Pres<uint64_t>&, int, int64_t, int, int32_t*, int32_t*, int32_t*, int64_t, int64_t*, acx_search_stats*, bool*);
template int run_greedy_device<u128>(const Pres<u128>&, int, int64_t, int, int32_t*, int32_t*, int32_t*, int64_t, int64_t*, acx_search_stats*, bool*);
template int run_greedy_device<u128x>(const Pres<u128x>&, int, int64_t, int, int32_t*, int32_t*, int32_t*, int64_t, int64_t*, acx_search_stats*, bool*);

}  // namespace acx
