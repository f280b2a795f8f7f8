// acx_search.hip -- device-resident BFS / greedy frontier over the AC graph (single GPU): ONE search per call (acx_search), the
// library's options and the digest / verbose hooks.  The device-resident greedy frontier of one search: acx_search_greedy.hip; many
// searches per call (acx_search_many, acx_search_groups): acx_search_many.hip.
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
//     breadth_first.py:87-89): a candidate claims an empty slot of the open-addressed table with a CAS and equal keys
//     fold to the smaller tag; the full key of the occupant is always compared (exact set).  BFS: the 8-byte stamp table
//     of acx_bfs.h; greedy: an id table
//   * winners are numbered by an exclusive scan in tag order, which reproduces the reference's insertion
//     order (BFS: k_bfs_count + k_bfs_compact, which also writes the nodes; the greedy batch-per-launch path: flag pass +
//     exclusive scan + k_commit); the per-parent budget test (breadth_first.py:91-95) becomes "first parent whose cumulative
//     winner count reaches the budget"; the solved test (:84-85) "minimum tag with total length 2"
//   * greedy additionally stops a batch right after the first parent that inserts a NEW child shorter than
//     the bucket (that child is the heap's next minimum); later parents stay queued
//
// Keys: a relator word and its length share one machine word (length in the top 6 bits):
// W = u64 for L <= 29, u128 for L <= 61, u128x (acx_keys.h: the same 128 bits, a key that needs no length field) for L <= 64.  Roofline: HBM (random table probes); see DESIGN.md.
#include <string.h>
#include <cstring>

#include "acx_searcher.h"
#include "acx_bfs.h"

namespace acx {

// ---- exclusive prefix sum of 32-bit flags (the batch-per-launch paths: greedy fallback, simplex graph) -------------------------------
// Round 6: three small kernels of the library's own instead of rocprim::exclusive_scan (the one rocprim call left; its include cost this
// unit several seconds of compile time).  Tiles of 4096 elements: scan inside the tile + the tile's total; one workgroup scans the
// totals; the tiles add their offset.  Not a throughput path.
constexpr int kScanU32Tile = 4096;
__global__ void __launch_bounds__(1024) k_scan_u32_tiles(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t* __restrict__ totals, size_t n) {
    __shared__ uint32_t s_w[16];
    ACX_VGPR_PAD("v31");
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const size_t p0 = (size_t)blockIdx.x * kScanU32Tile + (size_t)tid * 4;
    uint32_t v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] = p0 + k < n ? in[p0 + k] : 0u;
    const uint32_t sum = v[0] + v[1] + v[2] + v[3];
    uint32_t incl = sum;
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t a = (uint32_t)__shfl_up((int)incl, o);
        if (lane >= (uint32_t)o) incl += a;
    }
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    uint32_t before = incl - sum, total = 0;
#pragma unroll
    for (uint32_t w2 = 0; w2 < 16; w2++) {
        if (w2 < wave) before += s_w[w2];
        total += s_w[w2];
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (p0 + k < n) out[p0 + k] = before;
        before += v[k];
    }
    if (tid == 0) totals[blockIdx.x] = total;
}
__global__ void __launch_bounds__(1024) k_scan_u32_totals(uint32_t* __restrict__ totals, uint32_t tiles) {
    __shared__ uint32_t s_p[1024];
    ACX_VGPR_PAD("v31");
    const uint32_t tid = threadIdx.x, per = (tiles + 1023u) / 1024u;
    uint32_t sum = 0;
    for (uint32_t k = 0; k < per; k++) sum += tid * per + k < tiles ? totals[tid * per + k] : 0u;
    s_p[tid] = sum;
    __syncthreads();
    for (uint32_t o = 1; o < 1024; o <<= 1) {
        const uint32_t a = tid >= o ? s_p[tid - o] : 0u;
        __syncthreads();
        s_p[tid] += a;
        __syncthreads();
    }
    uint32_t ex = s_p[tid] - sum;
    for (uint32_t k = 0; k < per; k++) {
        const uint32_t t = tid * per + k;
        if (t < tiles) {
            const uint32_t c = totals[t];
            totals[t] = ex;
            ex += c;
        }
    }
}
__global__ void __launch_bounds__(1024) k_scan_u32_add(uint32_t* __restrict__ out, const uint32_t* __restrict__ totals, size_t n) {
    ACX_VGPR_PAD("v31");
    const uint32_t off = totals[blockIdx.x];
    const size_t p0 = (size_t)blockIdx.x * kScanU32Tile + (size_t)threadIdx.x * 4;
#pragma unroll
    for (int k = 0; k < 4; k++)
        if (p0 + k < n) out[p0 + k] += off;
}

int scan_u32_exclusive(void* tmp, size_t* tmp_bytes, const uint32_t* in, uint32_t* out, size_t n, hipStream_t st) {
    const size_t tiles = (n + kScanU32Tile - 1) / kScanU32Tile, need = (tiles + 1) * 4;
    if (!tmp) {
        *tmp_bytes = need;
        return ACX_OK;
    }
    if (*tmp_bytes < need) return fail(ACX_E_INVAL, "scan_u32_exclusive: %zu bytes of temporary storage, %zu needed", *tmp_bytes, need);
    if (n == 0) return ACX_OK;
    uint32_t* totals = (uint32_t*)tmp;
    hipLaunchKernelGGL(k_scan_u32_tiles, dim3((unsigned)tiles), dim3(1024), 0, st, in, out, totals, n);
    if (tiles > 1) {
        hipLaunchKernelGGL(k_scan_u32_totals, dim3(1), dim3(1024), 0, st, totals, (uint32_t)tiles);
        hipLaunchKernelGGL(k_scan_u32_add, dim3((unsigned)tiles), dim3(1024), 0, st, out, totals, n);
    }
    ACX_HIP_TRY(hipGetLastError());
    return ACX_OK;
}

std::atomic<int64_t> g_options[ACX_OPT_COUNT] = {{-1}, {-1}, {-1}, {-1}, {-1}, {-1}, {-1}, {-1}, {-1}};
const bool g_debug = getenv("ACX_DEBUG") != nullptr;  // the one look at the environment, when the library is loaded
std::atomic<int> g_digest_on{0};
thread_local uint64_t t_last_digest = 0;
thread_local int t_minima_on = 0;
thread_local std::vector<int32_t> t_last_minima;

static std::mutex g_handles_mu;
static std::vector<SearchHandles> g_handles_free;
int search_handles_take(SearchHandles& h) {
    int dev = 0;
    ACX_HIP_TRY(hipGetDevice(&dev));
    {
        std::lock_guard<std::mutex> lock(g_handles_mu);
        for (size_t i = 0; i < g_handles_free.size(); i++)
            if (g_handles_free[i].dev == dev) {
                h = g_handles_free[i];
                g_handles_free[i] = g_handles_free.back();
                g_handles_free.pop_back();
                return ACX_OK;
            }
    }
    h.dev = dev;
    // every search owns a stream, so that searches driven from different host threads overlap on the GPU; the cursor snapshots of
    // the fused BFS travel on a second one (a copy queued on `st` sits between two batches: 10 us)
    ACX_HIP_TRY(hipStreamCreateWithFlags(&h.st, hipStreamNonBlocking));
    ACX_HIP_TRY(hipStreamCreateWithFlags(&h.st_copy, hipStreamNonBlocking));
    for (auto& e : h.ev_cursor) ACX_HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto& e : h.ev_batch) ACX_HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    return ACX_OK;
}
void search_handles_give(SearchHandles& h) {
    if (!h.st) return;
    (void)hipStreamSynchronize(h.st_copy);  // (idle when a search ends normally; an error path may have left a copy in flight)
    (void)hipStreamSynchronize(h.st);
    std::lock_guard<std::mutex> lock(g_handles_mu);
    g_handles_free.push_back(h);
    h = SearchHandles();
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
    // The batch-per-launch greedy path below stays for two callers: verbose searches (the per-improvement lengths need each batch's
    // decision on the host) and searches that outgrow a capacity of the persistent kernel (depth >= 16 384, bucket arena); the tests
    // select it with ACX_OPT_GREEDY_HOST to hold it against the oracle.
    if constexpr (is_long_key<W>::value) {
        // max_relator_length 62 .. 64 (acx_keys.h): the key names a FREELY REDUCED word, which every state ACMove produces is -- the root has to be
        if (has_inverse_pair<W, true>(root.w0, root.n0) || has_inverse_pair<W, true>(root.w1, root.n1))
            return fail(ACX_E_INVAL, "acx_search: at max_relator_length %d (> 61) the presentation must be freely reduced", L);
    }
    if (greedy && !option(ACX_OPT_GREEDY_HOST, 0) && !t_minima_on) {  // device-resident priority frontier; falls through when it hits a capacity
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
    const bool stamp = !greedy;  // BFS: the stamp table of acx_bfs.h
    // a root in normal form keeps the whole search in normal form: the kernels then run the shorter move code (acx_bfs.h);
    // ACX_OPT_GENERAL_MOVE (tests): the general code for every root -- the two must build the same arena
    const int move_mode = !is_normal_form<W>(root, cyclical != 0) || option(ACX_OPT_GENERAL_MOVE, 0) ? kMoveGeneral : (cyclical ? kMoveNfCyclical : kMoveNf);
    Searcher<W> S;
    int rc = S.init(L, cyclical, max_nodes, bmax, greedy, false, stamp);
    if (rc) return rc;
    SearchDev<W>& d = S.d;
    hipStream_t st = S.st;
    EventPair evs;
    ACX_HIP_TRY(evs.create());
    hipEvent_t ev0 = evs.a, ev1 = evs.b;
    ACX_HIP_TRY(hipEventRecord(ev0, st));

    const uint32_t tl0 = (uint32_t)(root.n0 + root.n1);
    if (greedy) hipLaunchKernelGGL(k_root<W>, dim3(1), dim3(1), 0, st, d, keyops<W>::make(root.w0, root.n0), keyops<W>::make(root.w1, root.n1), tl0);
    else hipLaunchKernelGGL(k_bfs_root<W>, dim3(1), dim3(1), 0, st, d, keyops<W>::make(root.w0, root.n0), keyops<W>::make(root.w1, root.n1), tl0);
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
    const bool debug = g_debug;
    // ACX_OPT_BFS_NO_RUNAHEAD (tests): every batch's decision read back -- the path that small frontiers and verbose searches take anyway
    const bool no_runahead = option(ACX_OPT_BFS_NO_RUNAHEAD, 0) != 0;
    uint32_t adaptive = 64;  // greedy batch size: grows while buckets are consumed without a cut
    uint32_t printed_min = tl0;  // verbose mode: the minimum the reference has printed so far
    t_last_minima.clear();

    while (!done) {
        // ---- choose the batch of parents -------------------------------------------------------------
        uint32_t np = 0;
        const uint32_t* plist = nullptr;
        uint32_t pbegin = 0;
        int bucket_len = -1;
        uint32_t bucket_depth = 0;
        bool have_dec = false;  // the batch's decision is already in *dec (it came out of the run-ahead phase below)
        Decision* dec = (Decision*)S.h_pin;
        if (!greedy) {
            if (bfs_head >= nodes) break;  // queue exhausted (breadth_first.py:61)
            if (stamp && !d.first_len && !debug && !no_runahead && nodes - bfs_head >= bmax) {
                // ---- run-ahead (acx_frontier.h: BfsCursor): the frontier holds a full batch, so from here on the batches are enqueued
                // back to back with full-size grids and the host reads the cursor two batches late instead of every decision
                BfsCursor hc;
                memset(&hc, 0, sizeof(hc));
                hc.head = bfs_head;
                hc.nodes = (uint32_t)nodes;
                hc.batches = (uint32_t)batches;
                hc.expanded = expanded;
                hc.min_len = min_len;
                BfsCursor* dcur = (BfsCursor*)S.arena_cursor.p;
                BfsCursor* snap = (BfsCursor*)S.h_cursor;  // pinned, kRunAheadSlots entries
                ACX_HIP_TRY(hipMemcpyAsync(dcur, &hc, sizeof(hc), hipMemcpyHostToDevice, st));
                const uint32_t mcap = 12u * bmax;
                const dim3 egrid((bmax + kBfsParents - 1) / kBfsParents), eblock(kBfsThreads), cgrid((mcap + kCompactTile - 1) / kCompactTile), cblock(256);
                const BfsCursor* fin = nullptr;
                for (uint64_t k = 0; !fin; k++) {
                    if (batches + k + 1 >= (1ull << 30)) return fail(ACX_E_CAPACITY, "acx_search: more than 2^30 batches (the look-back words carry a 30-bit batch number)");
#define ACX_BFS_AHEAD(MODE)                                                                                                                              \
    hipLaunchKernelGGL((k_bfs_expand_insert<W, MODE>), egrid, eblock, 0, st, d, 0u, bmax, dcur);                                                       \
    hipLaunchKernelGGL(k_bfs_count<W>, cgrid, cblock, 0, st, d, mcap, S.d_counts, S.d_masks, dcur);                                                      \
    hipLaunchKernelGGL((k_bfs_compact<W, MODE>), cgrid, cblock, 0, st, d, 0u, mcap, 0u, (uint32_t)S.cap_nodes, S.d_counts, S.d_masks, S.d_total, dcur)
                    if (move_mode == kMoveNf) {
                        ACX_BFS_AHEAD(kMoveNf);
                    } else if (move_mode == kMoveNfCyclical) {
                        ACX_BFS_AHEAD(kMoveNfCyclical);
                    } else {
                        ACX_BFS_AHEAD(kMoveGeneral);
                    }
#undef ACX_BFS_AHEAD
                    const int slot = (int)(k % kRunAheadSlots);
                    hipLaunchKernelGGL(k_decide_tab<W>, dim3(1), dim3(1), 0, st, d, mcap, bmax, 0u, 0u, (uint32_t)S.cap_nodes, (long long)max_nodes, S.d_total, S.d_dec, 1, dcur,
                                       dcur + 1 + slot);
                    ACX_HIP_TRY(hipGetLastError());
                    // snapshot of the cursor as this batch leaves it: k_decide_tab wrote it into the batch's own device slot, which
                    // nothing touches again until the host has read it (the slot is reused by batch k + kRunAheadSlots, which is only
                    // enqueued after the host has waited for the copy of batch k + kRunAheadSlots - 1 - kRunAheadLag >= k, and the
                    // copies complete in order); copied on the side stream, behind an event of the main one -- never from the live
                    // cursor, which the next batch's kernels may be advancing while the copy runs
                    ACX_HIP_TRY(hipEventRecord(S.ev_batch[slot], st));
                    ACX_HIP_TRY(hipStreamWaitEvent(S.st_copy, S.ev_batch[slot], 0));
                    ACX_HIP_TRY(hipMemcpyAsync(&snap[slot], dcur + 1 + slot, sizeof(BfsCursor), hipMemcpyDeviceToHost, S.st_copy));
                    ACX_HIP_TRY(hipEventRecord(S.ev_cursor[slot], S.st_copy));
                    if (k >= kRunAheadLag) {
                        const int old = (int)((k - kRunAheadLag) % kRunAheadSlots);
                        ACX_HIP_TRY(hipEventSynchronize(S.ev_cursor[old]));
                        if (snap[old].status) fin = &snap[old];  // (the batches enqueued behind it returned at once and left the cursor alone)
                    }
                }
                bfs_head = fin->head;
                nodes = fin->nodes;
                batches = fin->batches;
                expanded = fin->expanded;
                min_len = std::min<uint32_t>(min_len, fin->min_len);
                if (fin->status == 3) continue;  // the queue ran empty: the check at the top of the loop ends the search
                // the batch that ends the search: finished below exactly like a batch whose decision was read back
                pbegin = fin->term_pbegin;
                np = fin->term_np;
                *dec = fin->term;
                have_dec = true;
            } else {
                np = (uint32_t)std::min<uint64_t>(nodes - bfs_head, bmax);
                pbegin = bfs_head;
            }
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
        if (!have_dec) {
        batches++;
        if (debug) fprintf(stderr, "[acx_search] batch %llu: np=%u nodes=%llu bucket=(%d,%u) buckets=%zu\n", (unsigned long long)batches, np,
                           (unsigned long long)nodes, bucket_len, bucket_depth, buckets.size());

        // ---- expand, dedup with min-tag resolution, number the winners, decide -- all on the stream --------
        if (!stamp || d.first_len) {  // (the stamp-table BFS resets its tags in k_decide_tab and its flags in k_bfs_compact)
            rc = S.reset_batch_scalars();
            if (rc) return rc;
        }
        d.min_len_start = printed_min;
        if (batches >= (1ull << 30)) return fail(ACX_E_CAPACITY, "acx_search: more than 2^30 batches (the look-back words carry a 30-bit batch number)");
        if (stamp) {
            // expand + dedup in one kernel, winners -> nodes in one pass, then the decision from the written nodes
            const dim3 egrid((np + kBfsParents - 1) / kBfsParents), eblock(kBfsThreads), cgrid((m + kCompactTile - 1) / kCompactTile);
#define ACX_BFS_LAUNCH(MODE)                                                                                                                          \
    hipLaunchKernelGGL((k_bfs_expand_insert<W, MODE>), egrid, eblock, 0, st, d, pbegin, np);                                    \
    hipLaunchKernelGGL(k_bfs_count<W>, cgrid, block, 0, st, d, m, S.d_counts, S.d_masks);                                                             \
    hipLaunchKernelGGL((k_bfs_compact<W, MODE>), cgrid, block, 0, st, d, pbegin, m, (uint32_t)nodes, (uint32_t)S.cap_nodes, S.d_counts, S.d_masks, S.d_total)
            if (move_mode == kMoveNf) {
                ACX_BFS_LAUNCH(kMoveNf);
            } else if (move_mode == kMoveNfCyclical) {
                ACX_BFS_LAUNCH(kMoveNfCyclical);
            } else {
                ACX_BFS_LAUNCH(kMoveGeneral);
            }
#undef ACX_BFS_LAUNCH
            hipLaunchKernelGGL(k_decide_tab<W>, dim3(1), dim3(1), 0, st, d, m, np, pbegin, (uint32_t)nodes, (uint32_t)S.cap_nodes, (long long)max_nodes, S.d_total, S.d_dec, 1);
        } else {  // greedy, batch per launch: expand, look up, in-batch dedup in a table sized for this batch, mark, scan, decide, commit
            hipLaunchKernelGGL(k_expand<W>, grid, block, 0, st, d, plist, pbegin, np);
            uint32_t bs = 1024;
            while (bs < 2 * m) bs <<= 1;
            hipLaunchKernelGGL(k_lookup<W>, grid, block, 0, st, d, m);
            ACX_HIP_TRY(hipMemsetAsync(d.bslots, 0xff, (size_t)bs * 4, st));
            hipLaunchKernelGGL(k_insert<W>, grid, block, 0, st, d, d.bslots, bs - 1, m, 1);
            hipLaunchKernelGGL(k_mark<W>, grid, block, 0, st, d, d.bslots, m, bucket_len);
            size_t tb = S.tmp_bytes;
            if (int src = scan_u32_exclusive(S.arena_tmp.p, &tb, d.cflag, d.cpos, m, st)) return src;
            hipLaunchKernelGGL(k_decide<W>, dim3(1), dim3(1), 0, st, d, m, np, (unsigned long long)nodes, (long long)max_nodes, 1, S.d_dec);
            hipLaunchKernelGGL(k_commit<W>, grid, block, 0, st, d, plist, pbegin, S.d_dec, m, (uint32_t)nodes, 1);
        }
        ACX_HIP_TRY(hipGetLastError());
        // one read-back: the decision and (greedy) the total lengths of the nodes this batch may have created
        ACX_HIP_TRY(hipMemcpyAsync(dec, S.d_dec, sizeof(Decision), hipMemcpyDeviceToHost, st));
        if (d.first_len) ACX_HIP_TRY(hipMemcpyAsync(S.h_first, d.first_len, kFirstLen * 8, hipMemcpyDeviceToHost, st));
        if (greedy) ACX_HIP_TRY(hipMemcpyAsync(S.h_pin + sizeof(Decision) + 64 - (sizeof(Decision) % 64), d.tlen + nodes, std::min<uint64_t>(m, S.cap_nodes - nodes), hipMemcpyDeviceToHost, st));
        ACX_HIP_TRY(hipStreamSynchronize(st));
        }  // !have_dec
        uint8_t* hl = S.h_pin + sizeof(Decision) + 64 - (sizeof(Decision) % 64);
        if (dec->err == 0xFE) return fail(ACX_E_CAPACITY, "acx_search: a probe sequence ran through the whole visited table (table full or damaged)");
        if (dec->err) return err_to_rc(dec->err);
        if (debug) fprintf(stderr, "[acx_search]   total=%u p_end=%u committed=%u solved=%u budget_hit=%u\n", dec->total, dec->p_end, dec->committed, dec->solved, dec->budget_hit);
        const uint32_t p_end = dec->p_end, committed = dec->committed;
        min_len = std::min<uint32_t>(min_len, dec->min_len);
        if (d.first_len) S.collect_minima(printed_min, dec->solved ? (unsigned long long)dec->solved_tag : 12ull * (p_end + 1) - 1);

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
    if (stats) {
        stats->nodes = (int64_t)nodes;
        stats->expanded = (int64_t)expanded;
        stats->children = (int64_t)expanded * 12;
        stats->levels = (int64_t)batches;
        stats->min_len = (int32_t)min_len;
        stats->seconds = ms * 1e-3;
    }
    rc = node_digest<W>(SoaKeys<W>{d.k0, d.k1}, d.parent, d.act, std::min<uint64_t>(nodes, S.cap_nodes), st);
    if (rc) return rc;
    if (*path_n > path_cap) return fail(ACX_E_CAPACITY, "path has %lld entries, buffer holds %lld", (long long)*path_n, (long long)path_cap);
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
    if (L > 64) return fail(ACX_E_INVAL, "acx_search handles max_relator_length <= 64, got %d", L);
    if (max_nodes < 0) max_nodes = 0;
    if (L <= 29) return run_search<uint64_t>(kind, h_presentation, L, max_nodes, cyclical, solved, path_action, path_len, path_cap, path_n, stats);
    if (L <= 61) return run_search<u128>(kind, h_presentation, L, max_nodes, cyclical, solved, path_action, path_len, path_cap, path_n, stats);
    return run_search<u128x>(kind, h_presentation, L, max_nodes, cyclical, solved, path_action, path_len, path_cap, path_n, stats);
}

extern "C" int acx_search_minima_enable(int on) {
    t_minima_on = on ? 1 : 0;
    return ACX_OK;
}

extern "C" int acx_search_last_minima(int32_t* lengths, int64_t cap, int64_t* n) {
    if (!n || cap < 0 || (cap > 0 && !lengths)) return fail(ACX_E_INVAL, "acx_search_last_minima: bad argument");
    *n = (int64_t)t_last_minima.size();
    for (int64_t k = 0; k < *n && k < cap; k++) lengths[k] = t_last_minima[(size_t)k];
    return ACX_OK;
}

extern "C" int acx_search_digest_enable(int on) {
    g_digest_on.store(on ? 1 : 0);
    return ACX_OK;
}

extern "C" int acx_search_last_digest(uint64_t* digest) {
    if (!digest) return fail(ACX_E_INVAL, "acx_search_last_digest: null pointer");
    *digest = t_last_digest;
    return ACX_OK;
}

namespace acx {
void host_buffers_trim();  // acx_search_many.hip: the pinned result buffers of the sweeps
}
extern "C" int acx_release_cached_memory(void) {
    block_pool().trim();
    acx::host_buffers_trim();
    return ACX_OK;
}

extern "C" int acx_set_option(int option, int64_t value) {
    if (option < 0 || option >= ACX_OPT_COUNT) return fail(ACX_E_INVAL, "acx_set_option: unknown option %d", option);
    g_options[option].store(value < 0 ? -1 : value);
    return ACX_OK;
}

extern "C" int64_t acx_get_option(int option) {
    if (option < 0 || option >= ACX_OPT_COUNT) return -1;
    return g_options[option].load();
}
