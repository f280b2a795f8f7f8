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
//     breadth_first.py:87-89): a candidate claims an empty entry of the open-addressed table with a CAS on a
//     provisional stamp and equal keys fold with atomicMin; the full key of the occupant is always compared
//     (exact set).  BFS: 32-byte entries with the key inline (acx_frontier.h, TabEntry); greedy: an id table
//   * winners are numbered by an exclusive scan in tag order, which reproduces the reference's insertion
//     order (BFS: inside k_compact_tab, which also writes the nodes -- one pass over the batch; the greedy
//     batch-per-launch path: flag pass + rocprim scan + k_commit); the per-parent budget test (breadth_first.py:91-95) becomes "first parent whose cumulative
//     winner count reaches the budget"; the solved test (:84-85) "minimum tag with total length 2"
//   * greedy additionally stops a batch right after the first parent that inserts a NEW child shorter than
//     the bucket (that child is the heap's next minimum); later parents stay queued
//
// Keys: a relator word and its length share one machine word (length in the top 6 bits):
// W = u64 for L <= 29, u128 for L <= 61.  Roofline: HBM (random table probes); see DESIGN.md.
#include <atomic>
#include <chrono>
#include <string>
#include <thread>

#include "acx_frontier.h"
#include "acx_bfs.h"
#include "acx_bfs_multi.h"
#include "acx_bfs_many.h"
#include "acx_greedy.h"
#include "acx_greedy_mega.h"

namespace acx {

// ---- node-arena digest (repeat-determinism tests) ------------------------------------------------------------------------
// acx_search_digest_enable(1) makes every search of this process finish with one extra pass that folds (id, key, parent,
// action) of all its nodes into a 64-bit sum; acx_search_last_digest returns the calling thread's last one.
static std::atomic<int> g_digest_on{0};
static thread_local uint64_t t_last_digest = 0;
// acx_search_minima_enable(1): every search records the total lengths at which the reference's verbose mode prints "New
// minimal length found" (breadth_first.py:79-82, greedy.py:85-89): each child, in generation order, that is shorter than
// everything generated before it, up to the child that ends the search.  acx_search_last_minima returns the sequence.
static thread_local int t_minima_on = 0;  // per calling thread: a verbose search must not slow down or lose the lines of searches on other threads
static thread_local std::vector<int32_t> t_last_minima;
constexpr int kFirstLen = 128;  // total lengths are <= 2 * 61

template <typename W, typename KEYS>
__global__ void __launch_bounds__(256) k_digest(KEYS keys, const uint32_t* __restrict__ parent, const uint8_t* __restrict__ act, uint32_t n, unsigned long long* __restrict__ out) {
    ACX_VGPR_PAD("v31");
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long h = 0;
    if (i < n) {
        W k0, k1;
        keys(i, k0, k1);
        h = mix64(fold(k0) + 0x9e3779b97f4a7c15ull * (i + 1)) ^ mix64(fold(k1) ^ ((uint64_t)parent[i] << 8 | act[i]));
        h = mix64(h + i);
    }
    for (int o = 32; o > 0; o >>= 1) h += (unsigned long long)__shfl_xor((long long)h, o);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, h);
}
template <typename W> struct SoaKeys {
    const W* k0;
    const W* k1;
    __device__ void operator()(uint32_t i, W& a, W& b) const {
        a = k0[i];
        b = k1[i];
    }
};
template <typename W> struct AosKeys {
    const NodeKey<W>* nk;
    __device__ void operator()(uint32_t i, W& a, W& b) const {
        a = nk[i].k0;
        b = nk[i].k1;
    }
};
template <typename W, typename KEYS> static int node_digest(KEYS keys, const uint32_t* parent, const uint8_t* act, uint64_t n, hipStream_t st) {
    if (!g_digest_on.load()) return ACX_OK;
    DevBuf out;
    if (out.alloc(8)) return ACX_E_NOMEM;
    ACX_HIP_TRY(hipMemsetAsync(out.p, 0, 8, st));
    hipLaunchKernelGGL((k_digest<W, KEYS>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, keys, parent, act, (uint32_t)n, (unsigned long long*)out.p);
    unsigned long long h = 0;
    ACX_HIP_TRY(hipMemcpyAsync(&h, out.p, 8, hipMemcpyDeviceToHost, st));
    ACX_HIP_TRY(hipStreamSynchronize(st));
    t_last_digest = h;
    return ACX_OK;
}

constexpr int kRunAheadSlots = 4, kRunAheadLag = 2;  // pinned snapshots of the BFS cursor / how many batches the host runs ahead of the one it reads

// The streams and events of a search.  Creating and destroying two streams and eight events per search cost ~0.7 ms of host
// time -- of a 9 ms search: finished searches leave theirs here (per device) for the next one.
struct SearchHandles {
    hipStream_t st = nullptr, st_copy = nullptr;
    hipEvent_t ev_cursor[kRunAheadSlots] = {}, ev_batch[kRunAheadSlots] = {};
    int dev = -1;
};
static std::mutex g_handles_mu;
static std::vector<SearchHandles> g_handles_free;
static int search_handles_take(SearchHandles& h) {
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
static void search_handles_give(SearchHandles& h) {
    if (!h.st) return;
    (void)hipStreamSynchronize(h.st_copy);  // (idle when a search ends normally; an error path may have left a copy in flight)
    (void)hipStreamSynchronize(h.st);
    std::lock_guard<std::mutex> lock(g_handles_mu);
    g_handles_free.push_back(h);
    h = SearchHandles();
}

template <typename W> struct Searcher {
    SearchDev<W> d;
    DevBuf arena_nodes, arena_cand, arena_tab, arena_btab, arena_scal, arena_tmp, arena_list, arena_path, arena_status, arena_first, arena_cursor;
    uint8_t* h_cursor = nullptr;  // pinned: kRunAheadSlots x BfsCursor (behind the Decision staging)
    SearchHandles handles;
    hipEvent_t ev_cursor[kRunAheadSlots] = {}, ev_batch[kRunAheadSlots] = {};  // (copies of the handles' events)
    hipStream_t st_copy = nullptr;  // the cursor snapshots travel on a stream of their own: a copy queued on `st` sits between two batches (10 us)
    unsigned long long h_first[kFirstLen];
    unsigned long long* d_status = nullptr;  // one-pass BFS commit: per-tile look-back words
    uint32_t* d_counts = nullptr;            // stamp-table BFS: winners per tile (k_bfs_count -> k_bfs_compact) ...
    uint32_t* d_masks = nullptr;             // ... and one winner bit per candidate
    uint32_t* d_ticket = nullptr;
    uint32_t* d_total = nullptr;
    size_t tmp_bytes = 0;
    uint64_t cap_nodes = 0, cap_cand = 0, n_slots = 0, n_bslots = 0;
    hipStream_t st = nullptr;
    Decision* d_dec = nullptr;   // device
    uint8_t* h_pin = nullptr;    // pinned host staging: Decision followed by the total lengths of the new nodes
    size_t h_pin_bytes = 0;

    ~Searcher() { search_handles_give(handles); }

    // inline_tab: BFS visited table with inline keys (TabEntry, round 1; kept for A/B runs: ACX_BFS_INLINE_TAB=1);
    // stamp_tab: the 8-byte stamp table of acx_bfs.h (no candidate keys at all); otherwise the id table of the greedy paths
    // lean: no key arrays and no table (the persistent greedy frontier keeps its own: GreedyDev::nkeys / tab)
    int init(int L, int cyclical, int64_t max_nodes, uint32_t batch_parents, bool greedy, bool inline_tab = false, bool lean = false, bool stamp_tab = false) {
        memset(&d, 0, sizeof(d));
        if (int rc = search_handles_take(handles)) return rc;
        st = handles.st;
        st_copy = handles.st_copy;
        for (int k = 0; k < kRunAheadSlots; k++) ev_cursor[k] = handles.ev_cursor[k], ev_batch[k] = handles.ev_batch[k];
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
            d.k0 = (W*)take(b, lean ? 0 : cap_nodes * sizeof(W));
            d.k1 = (W*)take(b, lean ? 0 : cap_nodes * sizeof(W));
            d.parent = (uint32_t*)take(b, cap_nodes * 4);
            d.depth = (uint32_t*)take(b, cap_nodes * 4);
            d.act = (uint8_t*)take(b, cap_nodes);
            d.tlen = (uint8_t*)take(b, cap_nodes);
            if (pass == 0 && arena_nodes.alloc(o)) return ACX_E_NOMEM;
        }
        for (int pass = 0; pass < 2; pass++) {
            uint8_t* b = (uint8_t*)arena_cand.p;
            o = 0;
            if (!stamp_tab) {
                d.ck0 = (W*)take(b, cap_cand * sizeof(W));
                d.ck1 = (W*)take(b, cap_cand * sizeof(W));
                d.cslot = (uint32_t*)take(b, cap_cand * 4);
                d.cflag = (uint32_t*)take(b, cap_cand * 4);
                d.cpos = (uint32_t*)take(b, cap_cand * 4);
                d.clen = (uint8_t*)take(b, cap_cand);
                d.cknown = (uint8_t*)take(b, cap_cand);
            }
            if (stamp_tab || (inline_tab && !getenv("ACX_BFS_CLASSIC_COMMIT"))) {  // one-pass BFS commit: byte flags
                d.btook = take(b, cap_cand + 8);
                d.brepl = take(b, cap_cand + 8);
            }
            if (pass == 0 && arena_cand.alloc(o)) return ACX_E_NOMEM;
        }
        if (lean) {
            d.tab = nullptr;
            d.slots = nullptr;
        } else if (stamp_tab) {
            if (arena_tab.alloc(n_slots * 8)) return ACX_E_NOMEM;
            d.stab = (unsigned long long*)arena_tab.p;
            d.stmask = (uint32_t)(n_slots - 1);
        } else if (inline_tab) {
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
        const size_t cursor_off = (sizeof(Decision) + 64 + cap_cand + 255) / 256 * 256;
        h_pin_bytes = cursor_off + (stamp_tab ? kRunAheadSlots * sizeof(BfsCursor) : 0);
        h_pin = pinned_staging(h_pin_bytes);
        if (!h_pin) return fail(ACX_E_NOMEM, "hipHostMalloc(%zu) failed", h_pin_bytes);
        if (stamp_tab) {  // run-ahead batches of the fused BFS (acx_frontier.h: BfsCursor)
            h_cursor = h_pin + cursor_off;
            if (arena_cursor.alloc((1 + kRunAheadSlots) * sizeof(BfsCursor))) return ACX_E_NOMEM;  // the live cursor + one snapshot slot per batch in flight
        }
        d.solved_tag = (unsigned long long*)(sc + 0);
        d.shorter_tag = (unsigned long long*)(sc + 8);
        d.err_tag = (unsigned long long*)(sc + 16);  // reset with the other batch scalars
        d.err = (uint32_t*)(sc + 24);
        d.min_len = (uint32_t*)(sc + 28);
        if (t_minima_on) {
            if (arena_first.alloc(kFirstLen * 8)) return ACX_E_NOMEM;
            d.first_len = (unsigned long long*)arena_first.p;
        }
        if (arena_list.alloc(std::max<uint64_t>(batch_parents, 1024) * 4 * 2)) return ACX_E_NOMEM;
        if (arena_path.alloc(8)) return ACX_E_NOMEM;
        if (!stamp_tab) {  // rocprim temporary storage for the scan over one batch
            size_t need = 0;
            if (rocprim::exclusive_scan(nullptr, need, d.cflag, d.cpos, 0u, cap_cand, rocprim::plus<uint32_t>(), st) != hipSuccess)
                return fail(ACX_E_NODEVICE, "rocprim::exclusive_scan sizing failed");
            tmp_bytes = need + 256;
            if (arena_tmp.alloc(tmp_bytes)) return ACX_E_NOMEM;
        }
        if (!lean) ACX_HIP_TRY(hipMemsetAsync(arena_tab.p, 0xff, n_slots * (stamp_tab ? 8 : inline_tab ? sizeof(TabEntry<W>) : 4), st));
        ACX_HIP_TRY(hipMemsetAsync(arena_scal.p, 0xff, 256, st));
        ACX_HIP_TRY(hipMemsetAsync(d.err, 0, 4, st));
        if (stamp_tab) ACX_HIP_TRY(hipMemsetAsync(d.brepl, 0, cap_cand + 8, st));  // once: k_bfs_compact zeroes what a batch sets
        d_ticket = (uint32_t*)(sc + 128);
        d_total = (uint32_t*)(sc + 132);
        ACX_HIP_TRY(hipMemsetAsync(d_ticket, 0, 8, st));
        if (inline_tab || stamp_tab) {  // status words of k_compact_tab / k_bfs_compact: epoch 0 = never written
            const size_t tiles = cap_cand / kCompactTile + 2;
            if (arena_status.alloc(tiles * 8 + (stamp_tab ? tiles * (4 + 1024) : 0))) return ACX_E_NOMEM;
            d_status = (unsigned long long*)arena_status.p;
            ACX_HIP_TRY(hipMemsetAsync(d_status, 0, tiles * 8, st));
            if (stamp_tab) {
                d_counts = (uint32_t*)(d_status + tiles);
                d_masks = d_counts + tiles;
            }
        }
        return ACX_OK;
    }

    int reset_batch_scalars() {  // solved / shorter / rank tags back to "none"; err and min_len are sticky
        ACX_HIP_TRY(hipMemsetAsync(arena_scal.p, 0xff, 24, st));
        if (d.first_len) ACX_HIP_TRY(hipMemsetAsync(d.first_len, 0xff, kFirstLen * 8, st));
        return ACX_OK;
    }

    // the lengths the reference prints during this batch: children with a tag <= end_tag, in tag order, each shorter than
    // everything before it (call after the batch's stream has been synchronised and h_first read back)
    void collect_minima(uint32_t& running_min, unsigned long long end_tag) {
        std::vector<std::pair<unsigned long long, int>> hits;
        for (int l = 0; l < kFirstLen && (uint32_t)l < running_min; l++)
            if (h_first[l] != ~0ull && h_first[l] <= end_tag) hits.emplace_back(h_first[l], l);
        std::sort(hits.begin(), hits.end());
        for (auto& hit : hits)
            if ((uint32_t)hit.second < running_min) {
                running_min = (uint32_t)hit.second;
                t_last_minima.push_back(hit.second);
            }
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
    DevBuf bk, bitmap, arena, gk0, gk1, gid, nkeys, tab;
    GreedyDev<W> g;
    // `st`: stream for the bucket-table memsets (nullptr = the search's own stream, S.st)
    int setup(const Pres<W>& root, int L, int64_t max_nodes, int cyclical, hipStream_t st) {
        int rc = S.init(L, cyclical, max_nodes, 1024, false, false, true);
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
    const uint64_t sort_cap = 2 * ((uint64_t)std::max<int64_t>(max_nodes, 1) + 64) + 4096;  // a bucket rounded up to a power of two
    auto up = [](uint64_t b) { return (b + 255) / 256 * 256; };
    const uint64_t b_slots = up(n_slots * 8), b_bk = up((uint64_t)nlen * kDepthCap * sizeof(BucketRec)), b_bm = up((uint64_t)nlen * (kDepthCap / 32) * 4);
    const uint64_t b_arena = up(arena_entries * 4), b_key = up(cap_nodes * sizeof(NodeKey<W>)), b_u32 = up(cap_nodes * 4), b_u8 = up(cap_nodes);
    const uint64_t b_gk = up(sort_cap * sizeof(W)), b_gid = up(sort_cap * 4);
    const uint64_t per_rest = b_arena + b_key + 2 * b_u32 + 2 * b_u8 + 2 * b_gk + b_gid;
    const uint64_t total = (uint64_t)n * (b_slots + b_bk + b_bm + per_rest);
    DevBuf big;
    if (big.alloc(total)) return ACX_E_NOMEM;
    uint8_t* base = (uint8_t*)big.p;
    uint8_t* p_slots = base;
    uint8_t* p_bk = p_slots + (uint64_t)n * b_slots;
    uint8_t* p_bm = p_bk + (uint64_t)n * b_bk;
    uint8_t* p_rest = p_bm + (uint64_t)n * b_bm;
    std::vector<GreedyDev<W>> hdev((size_t)n);
    bool all_nf = true;  // every root of the group in normal form: the launch runs the shorter move code
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
        g.tab = (unsigned long long*)(p_slots + (uint64_t)k * b_slots);
        g.tmask = (uint32_t)(n_slots - 1);
        g.bk = (BucketRec*)(p_bk + (uint64_t)k * b_bk);
        g.bitmap = (uint32_t*)(p_bm + (uint64_t)k * b_bm);
        g.arena = (uint32_t*)take(b_arena);
        g.nkeys = (NodeKey<W>*)take(b_key);
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
        g.nf = is_normal_form<W>(root, cyclical != 0) ? 1u : 0u;
        all_nf = all_nf && g.nf != 0;
        g.hand_min = 0;
        g.state = nullptr;
        g.mega_status = nullptr;
        g.hand_ctl = nullptr;
        g.rank_max = 0;
        g.root_k0 = keyops<W>::make(root.w0, root.n0);
        g.root_k1 = keyops<W>::make(root.w1, root.n1);
    }
    const int64_t pc = std::max<int64_t>(path_cap, 1);
    DevBuf ddev, douts, dpa, dpl;
    if (ddev.alloc((size_t)n * sizeof(GreedyDev<W>)) || douts.alloc((size_t)n * sizeof(GreedyOut)) || dpa.alloc((size_t)n * pc * 4) || dpl.alloc((size_t)n * pc * 4))
        return ACX_E_NOMEM;
    ACX_HIP_TRY(hipMemcpyAsync(ddev.p, hdev.data(), (size_t)n * sizeof(GreedyDev<W>), hipMemcpyHostToDevice, st));
    ACX_HIP_TRY(hipMemsetAsync(douts.p, 0, (size_t)n * sizeof(GreedyOut), st));
    EventPair evs;
    ACX_HIP_TRY(evs.create());
    hipEvent_t ev0 = evs.a, ev1 = evs.b;
    ACX_HIP_TRY(hipEventRecord(ev0, st));
    if (all_nf)
        hipLaunchKernelGGL((k_greedy_multi<W, true>), dim3((unsigned)n), dim3(kGreedyMultiThreads), 0, st, (const GreedyDev<W>*)ddev.p, (GreedyOut*)douts.p, (int32_t*)dpa.p,
                           (int32_t*)dpl.p, (long long)pc);
    else
        hipLaunchKernelGGL((k_greedy_multi<W, false>), dim3((unsigned)n), dim3(kGreedyMultiThreads), 0, st, (const GreedyDev<W>*)ddev.p, (GreedyOut*)douts.p, (int32_t*)dpa.p,
                           (int32_t*)dpl.p, (long long)pc);
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
    if (getenv("ACX_DEBUG")) {  // -DACX_GREEDY_PROFILE=1 builds: where the workgroups' cycles go, summed over the group's searches
        unsigned long long tp[12] = {}, tot = 0, batches = 0, sorts = 0, bigs = 0;
        for (int64_t k = 0; k < n; k++) {
            for (int q = 0; q < 12; q++) tp[q] += o[k].t_phase[q];
            batches += o[k].batches;
            sorts += o[k].sorts;
            bigs += o[k].big_sorts;
        }
        for (int q = 0; q < 8; q++) tot += tp[q];
        fprintf(stderr, "[acx_greedy_multi] %lld searches, %llu batches, %llu sorts (%llu of buckets larger than the LDS), group %.2f ms\n", (long long)n, batches, sorts, bigs, ms);
        if (tot)
            fprintf(stderr, "[acx_greedy_multi] %.3e workgroup cycles in all (thread 0's clock, summed over the searches)\n", (double)tot),
            fprintf(stderr, "[acx_greedy_multi] cycles%%: select %.1f sort %.1f expand %.1f probe %.1f scan %.1f commit %.1f file %.1f tail %.1f; of all cycles %.1f%% in sorts of buckets > LDS, %.1f%% in 256 < n <= LDS\n",
                    100.0 * tp[0] / tot, 100.0 * tp[1] / tot, 100.0 * tp[2] / tot, 100.0 * tp[3] / tot, 100.0 * tp[4] / tot, 100.0 * tp[5] / tot, 100.0 * tp[6] / tot, 100.0 * tp[7] / tot,
                    100.0 * tp[10] / tot, 100.0 * tp[11] / tot);
    }
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

// Greedy searches as JOBS on a fixed set of workgroup slots (acx_greedy.h: k_greedy_sched).  `groups`: batches of presentations, each
// with its own max_relator_length (all of one key width W); out0 = index of a batch's first search in the output arrays.
struct SearchGroupIn {
    const int8_t* rows;
    int64_t n;
    int L;
    int64_t out0;
};
static uint32_t greedy_slots_wanted() {
    const char* e = getenv("ACX_GREEDY_SLOTS");
    const long v = e ? atol(e) : 0;
    return v >= 1 ? (uint32_t)std::min<long>(v, 4096) : 256u;  // one workgroup of this kernel per compute unit (its 1024 lanes take the whole register file)
}
template <typename W>
static int run_greedy_sched(const std::vector<SearchGroupIn>& groups, int64_t max_nodes, int cyclical, int32_t* solved, int32_t* path_action, int32_t* path_len,
                            int64_t path_cap, int64_t* path_n, acx_search_stats* stats, int32_t* rc_out, uint8_t* need_rerun, uint32_t slots_cap) {
    int64_t n_all = 0;
    int L_max = 1;
    for (const auto& gr : groups) n_all += gr.n, L_max = std::max(L_max, gr.L);
    if (n_all == 0) return ACX_OK;
    const auto t_begin = std::chrono::steady_clock::now();
    auto since = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count(); };
    // a slot: the node arrays, the id table, the bucket table / bitmap / arena and the sort scratch of ONE search (as run_greedy_group lays them out)
    const uint64_t cap_nodes = (uint64_t)max_nodes + 64 + 12 * 1024;
    uint64_t n_slots = 1024;
    while (n_slots < 2 * (cap_nodes + 12288)) n_slots <<= 1;
    if (n_slots > (1ull << 31)) return fail(ACX_E_INVAL, "acx_search_many: budget too large for 32-bit node ids");
    const uint32_t nlen_max = (uint32_t)(2 * L_max + 1);
    const uint64_t arena_entries = std::min<uint64_t>(8ull * (uint64_t)std::max<int64_t>(max_nodes, 1) + (1ull << 20), 1ull << 31);
    const uint64_t sort_cap = 2 * ((uint64_t)std::max<int64_t>(max_nodes, 1) + 64) + 4096;
    auto up = [](uint64_t b) { return (b + 255) / 256 * 256; };
    const uint64_t b_slots = up(n_slots * 8), b_bk = up((uint64_t)nlen_max * kDepthCap * sizeof(BucketRec)), b_bm = up((uint64_t)nlen_max * (kDepthCap / 32) * 4);
    const uint64_t b_arena = up(arena_entries * 4), b_key = up(cap_nodes * sizeof(NodeKey<W>)), b_u32 = up(cap_nodes * 4), b_u8 = up(cap_nodes);
    const uint64_t b_gk = up(sort_cap * sizeof(W)), b_gid = up(sort_cap * 4);
    const uint64_t per_rest = b_arena + b_key + 2 * b_u32 + 2 * b_u8 + 2 * b_gk + b_gid;
    const uint64_t per_slot = b_slots + b_bk + b_bm + per_rest;
    size_t free_b = 0, total_b = 0;
    double avail = 64e9;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) avail = (double)free_b + (double)block_pool().cached_on(BlockPool::current_device());
    const double budget = std::max(2.0 * (double)per_slot, std::min(64e9, avail / 5.0));
    // the jobs, by move code: a root in normal form keeps its search in normal form (the shorter move code); one launch per code
    std::vector<GreedyJob<W>> jobs[2];
    std::vector<int64_t> where[2];  // job -> index in the output arrays
    for (const auto& gr : groups)
        for (int64_t k = 0; k < gr.n; k++) {
            const int64_t o = gr.out0 + k;
            need_rerun[o] = 0;
            rc_out[o] = ACX_OK;
            solved[o] = 0;
            path_n[o] = 0;
            Pres<W> root;
            bool ok = pack_relator<W>(gr.rows + k * 2 * gr.L, gr.L, root.w0, root.n0);
            ok = pack_relator<W>(gr.rows + k * 2 * gr.L + gr.L, gr.L, root.w1, root.n1) && ok;
            if (!ok) return fail(ACX_E_ROWERR, "acx_search_many: presentation %lld is not a zero-padded word pair over {+-1,+-2}", (long long)o);
            GreedyJob<W> jb;
            memset(&jb, 0, sizeof(jb));
            jb.root_k0 = keyops<W>::make(root.w0, root.n0);
            jb.root_k1 = keyops<W>::make(root.w1, root.n1);
            jb.root_len = (uint32_t)(root.n0 + root.n1);
            jb.nlen = (uint32_t)(2 * gr.L + 1);
            jb.L = gr.L;
            const int code = is_normal_form<W>(root, cyclical != 0) ? 1 : 0;
            jobs[code].push_back(jb);
            where[code].push_back(o);
        }
    // longest first, as far as one can tell beforehand: a search that is still running when the others are done has the chip to itself
    // (measured on the Miller-Schupp sweep at 1e6 nodes: an UNSOLVED search costs 1.33e8 workgroup cycles at max_relator_length 18, 1.20e8
    // at 20, 8.2e7 at 24, 6.9e7 at 28 -- the tighter the length bound, the more batches a node costs -- and a solved one a tenth of that;
    // which searches stay unsolved is not known beforehand, so: the smaller max_relator_length first, the longer relators first)
    for (int code = 0; code < 2; code++) {
        std::vector<size_t> order(jobs[code].size());
        for (size_t i = 0; i < order.size(); i++) order[i] = i;
        std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) {
            const GreedyJob<W>&x = jobs[code][a], &y = jobs[code][b];
            return x.L != y.L ? x.L < y.L : x.root_len > y.root_len;
        });
        std::vector<GreedyJob<W>> js(order.size());
        std::vector<int64_t> ws(order.size());
        for (size_t i = 0; i < order.size(); i++) js[i] = jobs[code][order[i]], ws[i] = where[code][order[i]];
        jobs[code].swap(js);
        where[code].swap(ws);
    }
    const uint32_t n_most = (uint32_t)std::max(jobs[0].size(), jobs[1].size());
    uint32_t R = (uint32_t)std::max<double>(1.0, std::min<double>(std::min<uint32_t>(n_most, std::max<uint32_t>(slots_cap, 1u)), budget / (double)per_slot));
    DevBuf big, dslots, djobs, dcounter, douts, dpa, dpl;
    while (big.alloc((uint64_t)R * per_slot)) {  // (the estimate of the free memory was too good: fewer slots -- the jobs just take longer)
        if (R == 1) return ACX_E_NOMEM;
        (void)hipGetLastError();  // (the failed hipMalloc's error must not meet the launch check below)
        R = (R + 1) / 2;
    }
    uint8_t* p_slots = (uint8_t*)big.p;
    uint8_t* p_bk = p_slots + (uint64_t)R * b_slots;
    uint8_t* p_bm = p_bk + (uint64_t)R * b_bk;
    uint8_t* p_rest = p_bm + (uint64_t)R * b_bm;
    std::vector<GreedyDev<W>> hslots((size_t)R);
    for (uint32_t r = 0; r < R; r++) {
        GreedyDev<W>& g = hslots[r];
        memset(&g, 0, sizeof(g));
        uint8_t* q = p_rest + (uint64_t)r * per_rest;
        auto take = [&](uint64_t bytes) {
            uint8_t* x = q;
            q += bytes;
            return x;
        };
        g.tab = (unsigned long long*)(p_slots + (uint64_t)r * b_slots);
        g.tmask = (uint32_t)(n_slots - 1);
        g.bk = (BucketRec*)(p_bk + (uint64_t)r * b_bk);
        g.bitmap = (uint32_t*)(p_bm + (uint64_t)r * b_bm);
        g.arena = (uint32_t*)take(b_arena);
        g.nkeys = (NodeKey<W>*)take(b_key);
        g.d.parent = (uint32_t*)take(b_u32);
        g.d.depth = (uint32_t*)take(b_u32);
        g.d.act = take(b_u8);
        g.d.tlen = take(b_u8);
        g.gk0 = (W*)take(b_gk);
        g.gk1 = (W*)take(b_gk);
        g.gid = (uint32_t*)take(b_gid);
        g.d.cyclical = cyclical;
        g.arena_cap = (uint32_t)arena_entries;
        g.nlen = nlen_max;  // (rows of the slot's bucket table; a job runs with its own 2 L + 1)
        g.max_nodes = (long long)max_nodes;
    }
    hipStream_t st = nullptr;
    {  // the 128-bit searches are the longer ones: when both widths are in flight their workgroups get a free compute unit first
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        ACX_HIP_TRY(hipStreamCreateWithPriority(&st, hipStreamNonBlocking, sizeof(W) > 8 ? hi : lo));
    }
    struct StreamGuard {
        hipStream_t s;
        ~StreamGuard() {
            (void)hipStreamSynchronize(s);
            (void)hipStreamDestroy(s);
        }
    } guard{st};
    const int64_t pc = std::max<int64_t>(path_cap, 1);
    if (dslots.alloc((size_t)R * sizeof(GreedyDev<W>)) || djobs.alloc((size_t)n_most * sizeof(GreedyJob<W>)) || dcounter.alloc(256) ||
        douts.alloc((size_t)n_most * sizeof(GreedyOut)) || dpa.alloc((size_t)n_most * pc * 4) || dpl.alloc((size_t)n_most * pc * 4))
        return ACX_E_NOMEM;
    EventPair evs;
    ACX_HIP_TRY(evs.create());
    for (int code = 1; code >= 0; code--) {
        const size_t nj = jobs[code].size();
        if (!nj) continue;
        for (uint32_t r = 0; r < R; r++) hslots[r].nf = (uint32_t)code;
        // the slots as a search expects them: table free, bucket records and bitmaps zero (between two jobs the workgroup does it itself)
        ACX_HIP_TRY(hipMemsetAsync(p_slots, 0xff, (uint64_t)R * b_slots, st));
        ACX_HIP_TRY(hipMemsetAsync(p_bk, 0, (uint64_t)R * (b_bk + b_bm), st));
        ACX_HIP_TRY(hipMemcpyAsync(dslots.p, hslots.data(), (size_t)R * sizeof(GreedyDev<W>), hipMemcpyHostToDevice, st));
        ACX_HIP_TRY(hipMemcpyAsync(djobs.p, jobs[code].data(), nj * sizeof(GreedyJob<W>), hipMemcpyHostToDevice, st));
        ACX_HIP_TRY(hipMemsetAsync(dcounter.p, 0, 256, st));
        ACX_HIP_TRY(hipMemsetAsync(douts.p, 0, nj * sizeof(GreedyOut), st));
        ACX_HIP_TRY(hipEventRecord(evs.a, st));
        const unsigned grid = (unsigned)std::min<size_t>(R, nj);
        if (code)
            hipLaunchKernelGGL((k_greedy_sched<W, true>), dim3(grid), dim3(kGreedyMultiThreads), 0, st, (const GreedyDev<W>*)dslots.p, (const GreedyJob<W>*)djobs.p, (uint32_t)nj,
                               (uint32_t*)dcounter.p, (GreedyOut*)douts.p, (int32_t*)dpa.p, (int32_t*)dpl.p, (long long)pc);
        else
            hipLaunchKernelGGL((k_greedy_sched<W, false>), dim3(grid), dim3(kGreedyMultiThreads), 0, st, (const GreedyDev<W>*)dslots.p, (const GreedyJob<W>*)djobs.p, (uint32_t)nj,
                               (uint32_t*)dcounter.p, (GreedyOut*)douts.p, (int32_t*)dpa.p, (int32_t*)dpl.p, (long long)pc);
        ACX_HIP_TRY(hipGetLastError());
        ACX_HIP_TRY(hipEventRecord(evs.b, st));
        const double t_launched = since();
        std::vector<GreedyOut> o(nj);
        std::vector<int32_t> pa(nj * (size_t)pc), pl(nj * (size_t)pc);
        ACX_HIP_TRY(hipMemcpyAsync(o.data(), douts.p, nj * sizeof(GreedyOut), hipMemcpyDeviceToHost, st));
        ACX_HIP_TRY(hipMemcpyAsync(pa.data(), dpa.p, nj * (size_t)pc * 4, hipMemcpyDeviceToHost, st));
        ACX_HIP_TRY(hipMemcpyAsync(pl.data(), dpl.p, nj * (size_t)pc * 4, hipMemcpyDeviceToHost, st));
        ACX_HIP_TRY(hipStreamSynchronize(st));
        float ms = 0;
        ACX_HIP_TRY(hipEventElapsedTime(&ms, evs.a, evs.b));
        if (getenv("ACX_DEBUG")) {  // -DACX_GREEDY_PROFILE=1 builds: where the workgroups' cycles go, summed over the launch's searches
            unsigned long long tp[12] = {}, tot = 0, batches = 0, sorts = 0, bigs = 0;
            for (size_t j = 0; j < nj; j++) {
                for (int q = 0; q < 12; q++) tp[q] += o[j].t_phase[q];
                batches += o[j].batches;
                sorts += o[j].sorts;
                bigs += o[j].big_sorts;
            }
            for (int q = 0; q < 8; q++) tot += tp[q];
            {  // the longest searches of the launch (their share of the launch's cycles decides how well any order can pack them)
                std::vector<unsigned long long> cyc(nj, 0);
                for (size_t j = 0; j < nj; j++)
                    for (int q = 0; q < 8; q++) cyc[j] += o[j].t_phase[q];
                std::vector<unsigned long long> sorted_c(cyc);
                std::sort(sorted_c.begin(), sorted_c.end());
                size_t first_long = nj;  // position (in job order) of the first search longer than half the longest
                for (size_t j = 0; j < nj && first_long == nj; j++)
                    if (2 * cyc[j] > sorted_c[nj - 1]) first_long = j;
                size_t last_long = 0;
                for (size_t j = 0; j < nj; j++)
                    if (2 * cyc[j] > sorted_c[nj - 1]) last_long = j;
                if (tot) {
                    std::vector<size_t> idx(nj);
                    for (size_t j = 0; j < nj; j++) idx[j] = j;
                    std::sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return cyc[a] > cyc[b]; });
                    for (size_t q = 0; q < std::min<size_t>(nj, 12); q++)
                        fprintf(stderr, "[acx_greedy_sched]   job %zu: L %d, root length %u, %.3e cycles, %u nodes, %llu batches, status %u\n", idx[q], jobs[code][idx[q]].L,
                                jobs[code][idx[q]].root_len, (double)cyc[idx[q]], o[idx[q]].nodes, (unsigned long long)o[idx[q]].batches, o[idx[q]].status);
                    // mean cycles by max_relator_length and outcome
                    for (int L0 = 1; L0 <= 61; L0++) {
                        double c[2] = {0, 0};
                        size_t cnt[2] = {0, 0};
                        for (size_t j = 0; j < nj; j++)
                            if (jobs[code][j].L == L0) c[o[j].status == GREEDY_SOLVED] += (double)cyc[j], cnt[o[j].status == GREEDY_SOLVED]++;
                        if (cnt[0] + cnt[1]) fprintf(stderr, "[acx_greedy_sched]   L %d: %zu unsolved, mean %.3e cycles; %zu solved, mean %.3e\n", L0, cnt[0], cnt[0] ? c[0] / cnt[0] : 0.0, cnt[1], cnt[1] ? c[1] / cnt[1] : 0.0);
                    }
                }
                if (tot)
                    fprintf(stderr, "[acx_greedy_sched] job cycles: longest %.3e, median %.3e, p90 %.3e; searches longer than half the longest: first at job %zu, last at job %zu of %zu\n",
                            (double)sorted_c[nj - 1], (double)sorted_c[nj / 2], (double)sorted_c[nj * 9 / 10], first_long, last_long, nj);
            }
            fprintf(stderr, "[acx_greedy_sched] %zu searches on %u slots (%s move code), %llu batches, %llu sorts (%llu of buckets larger than the LDS), launch %.2f ms; host: launched at %.1f ms, results at %.1f ms\n", nj, grid,
                    code ? "normal-form" : "general", batches, sorts, bigs, ms, t_launched, since());
            if (tot)
                fprintf(stderr, "[acx_greedy_sched] %.3e workgroup cycles; cycles%%: select %.1f sort %.1f expand %.1f probe %.1f scan %.1f commit %.1f file %.1f tail %.1f; %.1f%% in sorts of buckets > LDS, %.1f%% in 256 < n <= LDS\n",
                        (double)tot, 100.0 * tp[0] / tot, 100.0 * tp[1] / tot, 100.0 * tp[2] / tot, 100.0 * tp[3] / tot, 100.0 * tp[4] / tot, 100.0 * tp[5] / tot, 100.0 * tp[6] / tot,
                        100.0 * tp[7] / tot, 100.0 * tp[10] / tot, 100.0 * tp[11] / tot);
        }
        for (size_t j = 0; j < nj; j++) {
            const int64_t k = where[code][j];
            const GreedyOut& r = o[j];
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
                memcpy(path_action + k * path_cap, pa.data() + j * (size_t)pc, (size_t)r.path_n * 4);
                memcpy(path_len + k * path_cap, pl.data() + j * (size_t)pc, (size_t)r.path_n * 4);
            }
            if (stats) {
                stats[k].nodes = (int64_t)r.nodes;
                stats[k].expanded = (int64_t)r.expanded;
                stats[k].children = (int64_t)r.expanded * 12;
                stats[k].levels = (int64_t)r.batches;
                stats[k].min_len = (int32_t)r.min_len;
                stats[k].seconds = ms * 1e-3;  // of the whole launch
            }
        }
    }
    return ACX_OK;
}

// A group of independent breadth-first searches in one launch of k_bfs_multi (one workgroup each).  rc_out[k] = ACX_OK /
// ACX_E_CAPACITY (path buffer, or a probe sequence that ran through the whole table) / ACX_E_ROWERR (the reference raises).
template <typename W>
static int run_bfs_group(const int8_t* rows, int64_t n, int L, int64_t max_nodes, int cyclical, int32_t* solved, int32_t* path_action, int32_t* path_len,
                         int64_t path_cap, int64_t* path_n, acx_search_stats* stats, int32_t* rc_out) {
    const uint64_t cap_nodes = (uint64_t)std::max<int64_t>(max_nodes, 1) + 12 * kBmMaxParents + 64;
    uint64_t n_slots = 1024;
    while (n_slots < 2 * cap_nodes) n_slots <<= 1;
    if (n_slots > (1ull << 31)) return fail(ACX_E_INVAL, "acx_search_many: budget too large for 32-bit node ids");
    auto up = [](uint64_t b) { return (b + 255) / 256 * 256; };
    const uint64_t b_tab = up(n_slots * 8), b_key = up(cap_nodes * sizeof(W)), b_u32 = up(cap_nodes * 4), b_u8 = up(cap_nodes);
    const uint64_t per_rest = 2 * b_key + 2 * b_u32 + 2 * b_u8;
    DevBuf big;
    if (big.alloc((uint64_t)n * (b_tab + per_rest))) return ACX_E_NOMEM;
    uint8_t* p_tab = (uint8_t*)big.p;
    uint8_t* p_rest = p_tab + (uint64_t)n * b_tab;
    hipStream_t st = nullptr;
    ACX_HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    struct StreamGuard {
        hipStream_t s;
        ~StreamGuard() { (void)hipStreamDestroy(s); }
    } guard{st};
    ACX_HIP_TRY(hipMemsetAsync(p_tab, 0xff, (uint64_t)n * b_tab, st));
    // a root in normal form keeps its whole search in normal form (acx_bfs.h): those searches run the shorter move code
    std::vector<BfsJob<W>> hjobs((size_t)n);
    std::vector<int64_t> order[2];  // [0] general move code, [1] normal form
    for (int64_t k = 0; k < n; k++) {
        rc_out[k] = ACX_OK;
        solved[k] = 0;
        path_n[k] = 0;
        Pres<W> root;
        bool ok = pack_relator<W>(rows + k * 2 * L, L, root.w0, root.n0);
        ok = pack_relator<W>(rows + k * 2 * L + L, L, root.w1, root.n1) && ok;
        if (!ok) return fail(ACX_E_ROWERR, "acx_search_many: presentation %lld is not a zero-padded word pair over {+-1,+-2}", (long long)k);
        order[is_normal_form<W>(root, cyclical != 0) && !getenv("ACX_BFS_GENERAL_MOVE") ? 1 : 0].push_back(k);
        BfsJob<W>& g = hjobs[(size_t)k];
        memset(&g, 0, sizeof(g));
        uint8_t* q = p_rest + (uint64_t)k * per_rest;
        auto take = [&](uint64_t bytes) {
            uint8_t* r = q;
            q += bytes;
            return r;
        };
        g.stab = (unsigned long long*)(p_tab + (uint64_t)k * b_tab);
        g.stmask = (uint32_t)(n_slots - 1);
        g.k0 = (W*)take(b_key);
        g.k1 = (W*)take(b_key);
        g.parent = (uint32_t*)take(b_u32);
        g.depth = (uint32_t*)take(b_u32);
        g.act = take(b_u8);
        g.tlen = take(b_u8);
        g.cap_nodes = (uint32_t)cap_nodes;
        g.root_k0 = keyops<W>::make(root.w0, root.n0);
        g.root_k1 = keyops<W>::make(root.w1, root.n1);
        g.max_nodes = (long long)max_nodes;
        g.L = L;
        g.cyclical = cyclical;
    }
    // jobs of one move code are contiguous on the device: [general ...][normal form ...]; slot j of the launch arrays
    std::vector<BfsJob<W>> sorted_jobs;
    std::vector<int64_t> slot_of;  // launch slot -> search index
    for (int m = 0; m < 2; m++)
        for (int64_t k : order[m]) {
            sorted_jobs.push_back(hjobs[(size_t)k]);
            slot_of.push_back(k);
        }
    const int64_t pc = std::max<int64_t>(path_cap, 1);
    DevBuf djobs, douts, dpa, dpl;
    if (djobs.alloc((size_t)n * sizeof(BfsJob<W>)) || douts.alloc((size_t)n * sizeof(BfsOut)) || dpa.alloc((size_t)n * pc * 4) || dpl.alloc((size_t)n * pc * 4)) return ACX_E_NOMEM;
    ACX_HIP_TRY(hipMemcpyAsync(djobs.p, sorted_jobs.data(), (size_t)n * sizeof(BfsJob<W>), hipMemcpyHostToDevice, st));
    ACX_HIP_TRY(hipMemsetAsync(douts.p, 0, (size_t)n * sizeof(BfsOut), st));
    EventPair evs;
    ACX_HIP_TRY(evs.create());
    ACX_HIP_TRY(hipEventRecord(evs.a, st));
    const int64_t n_gen = (int64_t)order[0].size(), n_nf = (int64_t)order[1].size();
    const BfsJob<W>* dj = (const BfsJob<W>*)djobs.p;
    BfsOut* dout = (BfsOut*)douts.p;
    if (n_gen)
        hipLaunchKernelGGL((k_bfs_multi<W, kMoveGeneral>), dim3((unsigned)n_gen), dim3(kBmT), 0, st, dj, dout, (int32_t*)dpa.p, (int32_t*)dpl.p, (long long)pc);
    if (n_nf) {
        if (cyclical)
            hipLaunchKernelGGL((k_bfs_multi<W, kMoveNfCyclical>), dim3((unsigned)n_nf), dim3(kBmT), 0, st, dj + n_gen, dout + n_gen, (int32_t*)dpa.p + n_gen * pc,
                               (int32_t*)dpl.p + n_gen * pc, (long long)pc);
        else
            hipLaunchKernelGGL((k_bfs_multi<W, kMoveNf>), dim3((unsigned)n_nf), dim3(kBmT), 0, st, dj + n_gen, dout + n_gen, (int32_t*)dpa.p + n_gen * pc,
                               (int32_t*)dpl.p + n_gen * pc, (long long)pc);
    }
    ACX_HIP_TRY(hipGetLastError());
    ACX_HIP_TRY(hipEventRecord(evs.b, st));
    std::vector<BfsOut> o((size_t)n);
    std::vector<int32_t> pa((size_t)n * pc), pl((size_t)n * pc);
    ACX_HIP_TRY(hipMemcpyAsync(o.data(), douts.p, (size_t)n * sizeof(BfsOut), hipMemcpyDeviceToHost, st));
    ACX_HIP_TRY(hipMemcpyAsync(pa.data(), dpa.p, (size_t)n * pc * 4, hipMemcpyDeviceToHost, st));
    ACX_HIP_TRY(hipMemcpyAsync(pl.data(), dpl.p, (size_t)n * pc * 4, hipMemcpyDeviceToHost, st));
    ACX_HIP_TRY(hipStreamSynchronize(st));
    float ms = 0;
    ACX_HIP_TRY(hipEventElapsedTime(&ms, evs.a, evs.b));
    if (getenv("ACX_DEBUG")) {  // -DACX_BFS_MULTI_PROFILE=1 builds: where a chunk's time goes, summed over the group's searches
        unsigned long long tp[6] = {}, tot = 0, chunks = 0;
        for (int64_t j = 0; j < n; j++) {
            for (int q = 0; q < 6; q++) tp[q] += o[(size_t)j].t_phase[q];
            chunks += o[(size_t)j].batches;
        }
        for (int q = 0; q < 6; q++) tot += tp[q];
        if (tot)
            fprintf(stderr, "[acx_bfs_multi] %lld searches, %llu chunks, %.0f cycles per chunk: expand %.1f %% fold %.1f table %.1f number+decide %.1f commit %.1f tail %.1f; group %.2f ms\n",
                    (long long)n, chunks, (double)tot / (double)(chunks ? chunks : 1), 100.0 * tp[0] / tot, 100.0 * tp[1] / tot, 100.0 * tp[2] / tot, 100.0 * tp[3] / tot,
                    100.0 * tp[4] / tot, 100.0 * tp[5] / tot, ms);
    }
    for (int64_t j = 0; j < n; j++) {
        const int64_t k = slot_of[(size_t)j];
        const BfsOut& r = o[(size_t)j];
        if (r.status == BFS_MOVE_ERROR) {
            rc_out[k] = err_to_rc(r.err);
            continue;
        }
        if (r.status == BFS_TABLE_FULL) {
            rc_out[k] = fail(ACX_E_CAPACITY, "acx_search_many: a probe sequence ran through the whole visited table of search %lld", (long long)k);
            continue;
        }
        if (r.status != BFS_SOLVED && r.status != BFS_BUDGET && r.status != BFS_EXHAUSTED) {
            rc_out[k] = fail(ACX_E_NODEVICE, "bfs frontier kernel ended in state %u", r.status);
            continue;
        }
        solved[k] = r.status == BFS_SOLVED ? 1 : 0;
        path_n[k] = r.path_n;
        if ((int64_t)r.path_n > path_cap) {
            rc_out[k] = fail(ACX_E_CAPACITY, "path has %u entries, buffer holds %lld", r.path_n, (long long)path_cap);
        } else if (path_action && path_len && r.path_n) {
            memcpy(path_action + k * path_cap, pa.data() + j * pc, (size_t)r.path_n * 4);
            memcpy(path_len + k * path_cap, pl.data() + j * pc, (size_t)r.path_n * 4);
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

// A group of independent breadth-first searches, level-synchronous on the kernels of the fused single search (acx_bfs_many.h): one
// round of launches advances every running search by one batch of at most `bmax` parents.  Same contract as run_bfs_group.
static uint32_t bfs_many_bmax() {
    const char* e = getenv("ACX_BFS_MANY_BMAX");
    const long v = e ? atol(e) : 0;
    return v >= 128 ? (uint32_t)std::min<long>(v, 1l << 22) : (1u << 15);
}
template <typename W>
static int run_bfs_group_fused(const int8_t* rows, int64_t n, int L, int64_t max_nodes, int cyclical, int32_t* solved, int32_t* path_action, int32_t* path_len,
                               int64_t path_cap, int64_t* path_n, acx_search_stats* stats, int32_t* rc_out) {
    if (n <= 0) return ACX_OK;
    const uint32_t bmax = (uint32_t)std::min<int64_t>(std::max<int64_t>(max_nodes / 4, 1024), bfs_many_bmax());
    const uint64_t cap_nodes = (uint64_t)std::max<int64_t>(max_nodes, 0) + 64, cap_cand = 12ull * bmax;
    uint64_t n_slots = 1024;
    static const double load = getenv("ACX_BFS_MANY_LOAD") ? atof(getenv("ACX_BFS_MANY_LOAD")) : 2.0;  // slots per stamp the table may ever hold
    while ((double)n_slots < load * (double)(cap_nodes + cap_cand)) n_slots <<= 1;
    if (n_slots > (1ull << 31)) return fail(ACX_E_INVAL, "acx_search_many: budget too large for 32-bit node ids");
    auto up = [](uint64_t b) { return (b + 255) / 256 * 256; };
    const uint64_t tiles = cap_cand / kCompactTile + 2;
    const uint64_t b_tab = up(n_slots * 8), b_key = up(cap_nodes * sizeof(W)), b_u32 = up(cap_nodes * 4), b_u8 = up(cap_nodes), b_flag = up(cap_cand + 8);
    const uint64_t b_counts = up(tiles * 4), b_masks = up(tiles * 1024), b_scal = 256;
    const uint64_t per_rest = 2 * b_key + 2 * b_u32 + 2 * b_u8 + b_flag + b_counts + b_masks + b_scal;
    // [tables of all searches][replaced-flags of all searches] (one memset each), then the rest search by search
    // (every device block is declared before the stream guard below: an error return first waits for the streams, then frees)
    DevBuf big, dcur, dq, droots, dstatus, dwant, dpa, dpl, dpn;
    if (big.alloc((uint64_t)n * (b_tab + b_flag + per_rest))) return ACX_E_NOMEM;
    uint8_t* p_tab = (uint8_t*)big.p;
    uint8_t* p_repl = p_tab + (uint64_t)n * b_tab;
    uint8_t* p_rest = p_repl + (uint64_t)n * b_flag;
    SearchHandles H;
    if (int rc = search_handles_take(H)) return rc;
    struct Give {
        SearchHandles& h;
        ~Give() { search_handles_give(h); }
    } give{H};
    hipStream_t st = H.st;
    ACX_HIP_TRY(hipMemsetAsync(p_tab, 0xff, (uint64_t)n * b_tab, st));
    ACX_HIP_TRY(hipMemsetAsync(p_repl, 0, (uint64_t)n * b_flag, st));  // once: k_bfs_count zeroes what a batch sets
    std::vector<int64_t> order[2];  // [0] general move code, [1] normal form (a root in normal form keeps its whole search there)
    std::vector<Pres<W>> roots((size_t)n);
    for (int64_t k = 0; k < n; k++) {
        rc_out[k] = ACX_OK;
        solved[k] = 0;
        path_n[k] = 0;
        Pres<W>& root = roots[(size_t)k];
        bool ok = pack_relator<W>(rows + k * 2 * L, L, root.w0, root.n0);
        ok = pack_relator<W>(rows + k * 2 * L + L, L, root.w1, root.n1) && ok;
        if (!ok) return fail(ACX_E_ROWERR, "acx_search_many: presentation %lld is not a zero-padded word pair over {+-1,+-2}", (long long)k);
        order[is_normal_form<W>(root, cyclical != 0) && !getenv("ACX_BFS_GENERAL_MOVE") ? 1 : 0].push_back(k);
    }
    if (dcur.alloc((size_t)n * sizeof(BfsCursor))) return ACX_E_NOMEM;  // the searches' cursors, contiguous: one copy brings them all back
    std::vector<BfsMany<W>> hq;
    std::vector<W> hroots;
    std::vector<int64_t> slot_of;  // launch slot -> search index: [general ...][normal form ...]
    for (int mode = 0; mode < 2; mode++)
        for (int64_t k : order[mode]) {
            const size_t j = slot_of.size();
            slot_of.push_back(k);
            BfsMany<W> e;
            memset(&e, 0, sizeof(e));
            uint8_t* q = p_rest + (uint64_t)j * per_rest;
            auto take = [&](uint64_t bytes) {
                uint8_t* r = q;
                q += bytes;
                return r;
            };
            SearchDev<W>& d = e.d;
            d.L = L;
            d.cyclical = cyclical;
            d.stab = (unsigned long long*)(p_tab + (uint64_t)j * b_tab);
            d.stmask = (uint32_t)(n_slots - 1);
            d.brepl = p_repl + (uint64_t)j * b_flag;
            d.k0 = (W*)take(b_key);
            d.k1 = (W*)take(b_key);
            d.parent = (uint32_t*)take(b_u32);
            d.depth = (uint32_t*)take(b_u32);
            d.act = take(b_u8);
            d.tlen = take(b_u8);
            d.btook = take(b_flag);
            e.counts = (uint32_t*)take(b_counts);
            e.masks = (uint32_t*)take(b_masks);
            uint8_t* sc = take(b_scal);
            d.solved_tag = (unsigned long long*)(sc + 0);
            d.shorter_tag = (unsigned long long*)(sc + 8);
            d.err_tag = (unsigned long long*)(sc + 16);
            d.err = (uint32_t*)(sc + 24);
            d.min_len = (uint32_t*)(sc + 28);
            e.dec = (Decision*)(sc + 64);
            e.total = (uint32_t*)(sc + 132);
            e.cur = (BfsCursor*)dcur.p + j;
            hq.push_back(e);
            hroots.push_back(keyops<W>::make(roots[(size_t)k].w0, roots[(size_t)k].n0));
            hroots.push_back(keyops<W>::make(roots[(size_t)k].w1, roots[(size_t)k].n1));
        }
    const int64_t pc = std::max<int64_t>(path_cap, 1);
    const uint32_t un = (uint32_t)n;
    if (dq.alloc((size_t)n * sizeof(BfsMany<W>)) || droots.alloc((size_t)n * 2 * sizeof(W)) || dstatus.alloc((size_t)n * 4 * kRunAheadSlots) || dwant.alloc((size_t)n * 4) ||
        dpa.alloc((size_t)n * pc * 4) || dpl.alloc((size_t)n * pc * 4) || dpn.alloc((size_t)n * 4))
        return ACX_E_NOMEM;
    const size_t cur_off = up((size_t)n * 4 * kRunAheadSlots);
    uint8_t* pin = pinned_staging(cur_off + (size_t)n * sizeof(BfsCursor));
    if (!pin) return fail(ACX_E_NOMEM, "hipHostMalloc failed");
    uint32_t* h_status = (uint32_t*)pin;
    ACX_HIP_TRY(hipMemcpyAsync(dq.p, hq.data(), (size_t)n * sizeof(BfsMany<W>), hipMemcpyHostToDevice, st));
    ACX_HIP_TRY(hipMemcpyAsync(droots.p, hroots.data(), (size_t)n * 2 * sizeof(W), hipMemcpyHostToDevice, st));
    EventPair evs;
    ACX_HIP_TRY(evs.create());
    ACX_HIP_TRY(hipEventRecord(evs.a, st));
    const BfsMany<W>* q = (const BfsMany<W>*)dq.p;
    const uint32_t n_gen = (uint32_t)order[0].size(), n_nf = (uint32_t)order[1].size();
    const dim3 sgrid((un + 63) / 64), sblock(64);
    hipLaunchKernelGGL(k_bfs_root_many<W>, sgrid, sblock, 0, st, q, (const W*)droots.p, un);
    const uint32_t mcap = 12u * bmax;
    uint64_t bound = 1;  // no search has more than `bound` parents queued in this round (a batch multiplies the nodes by at most 13)
    uint64_t rounds = 0;
    for (uint64_t k = 0;; k++) {
        if (k + 1 >= (1ull << 30)) return fail(ACX_E_CAPACITY, "acx_search_many: more than 2^30 batches");
        const uint32_t bp = (uint32_t)std::min<uint64_t>(bound, bmax);
        const unsigned ex = (bp + kBfsParents - 1) / kBfsParents, cx = (12u * bp + kCompactTile - 1) / kCompactTile;
        if (n_gen) hipLaunchKernelGGL((k_bfs_expand_insert_many<W, kMoveGeneral>), dim3(ex, n_gen), dim3(kBfsThreads), 0, st, q, bmax);
        if (n_nf) {
            if (cyclical) hipLaunchKernelGGL((k_bfs_expand_insert_many<W, kMoveNfCyclical>), dim3(ex, n_nf), dim3(kBfsThreads), 0, st, q + n_gen, bmax);
            else hipLaunchKernelGGL((k_bfs_expand_insert_many<W, kMoveNf>), dim3(ex, n_nf), dim3(kBfsThreads), 0, st, q + n_gen, bmax);
        }
        hipLaunchKernelGGL(k_bfs_count_many<W>, dim3(cx, un), dim3(256), 0, st, q, mcap);
        if (n_gen) hipLaunchKernelGGL((k_bfs_compact_many<W, kMoveGeneral>), dim3(cx, n_gen), dim3(256), 0, st, q, mcap, (uint32_t)cap_nodes);
        if (n_nf) {
            if (cyclical) hipLaunchKernelGGL((k_bfs_compact_many<W, kMoveNfCyclical>), dim3(cx, n_nf), dim3(256), 0, st, q + n_gen, mcap, (uint32_t)cap_nodes);
            else hipLaunchKernelGGL((k_bfs_compact_many<W, kMoveNf>), dim3(cx, n_nf), dim3(256), 0, st, q + n_gen, mcap, (uint32_t)cap_nodes);
        }
        const int slot = (int)(k % kRunAheadSlots);
        uint32_t* dst = (uint32_t*)dstatus.p + (size_t)slot * n;
        hipLaunchKernelGGL(k_decide_tab_many<W>, sgrid, sblock, 0, st, q, un, mcap, bmax, (uint32_t)cap_nodes, (long long)max_nodes, dst);
        ACX_HIP_TRY(hipGetLastError());
        // the round's status words, copied on the side stream behind an event of the main one (the slot is reused four rounds later, which
        // is only enqueued after the host has waited for this copy)
        ACX_HIP_TRY(hipEventRecord(H.ev_batch[slot], st));
        ACX_HIP_TRY(hipStreamWaitEvent(H.st_copy, H.ev_batch[slot], 0));
        ACX_HIP_TRY(hipMemcpyAsync(h_status + (size_t)slot * n, dst, (size_t)n * 4, hipMemcpyDeviceToHost, H.st_copy));
        ACX_HIP_TRY(hipEventRecord(H.ev_cursor[slot], H.st_copy));
        bound = std::min<uint64_t>(bound * 13, 1ull << 40);
        rounds = k + 1;
        if (k >= kRunAheadLag) {
            const int old = (int)((k - kRunAheadLag) % kRunAheadSlots);
            ACX_HIP_TRY(hipEventSynchronize(H.ev_cursor[old]));
            bool all = true;
            for (int64_t j = 0; j < n && all; j++) all = h_status[(size_t)old * n + j] != 0;
            if (all) break;  // (the rounds enqueued behind it found every cursor ended and left them alone)
        }
    }
    // every search has ended: its cursor says how (status 3: the queue ran empty; 1: `term` is the batch that ended it, not applied)
    BfsCursor* hc = (BfsCursor*)(pin + cur_off);
    ACX_HIP_TRY(hipMemcpyAsync(hc, dcur.p, (size_t)n * sizeof(BfsCursor), hipMemcpyDeviceToHost, st));
    ACX_HIP_TRY(hipStreamSynchronize(st));
    std::vector<uint32_t> want((size_t)n, kEmpty);
    bool any_path = false;
    for (int64_t j = 0; j < n; j++) {
        const BfsCursor& c = hc[j];
        if (c.status == 1 && c.term.solved && !c.term.err) {
            want[(size_t)j] = c.term_pbegin + c.term.solved_tag / 12;
            any_path = true;
        }
    }
    std::vector<int32_t> pa, pl;
    std::vector<uint32_t> pn((size_t)n, 0);
    if (any_path) {
        pa.resize((size_t)n * pc);
        pl.resize((size_t)n * pc);
        ACX_HIP_TRY(hipMemcpyAsync(dwant.p, want.data(), (size_t)n * 4, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_paths_many<W>, sgrid, sblock, 0, st, q, un, (const uint32_t*)dwant.p, (int32_t*)dpa.p, (int32_t*)dpl.p, (uint32_t*)dpn.p, (long long)pc);
        ACX_HIP_TRY(hipGetLastError());
        ACX_HIP_TRY(hipMemcpyAsync(pn.data(), dpn.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
        ACX_HIP_TRY(hipMemcpyAsync(pa.data(), dpa.p, (size_t)n * pc * 4, hipMemcpyDeviceToHost, st));
        ACX_HIP_TRY(hipMemcpyAsync(pl.data(), dpl.p, (size_t)n * pc * 4, hipMemcpyDeviceToHost, st));
    }
    ACX_HIP_TRY(hipEventRecord(evs.b, st));
    ACX_HIP_TRY(hipStreamSynchronize(st));
    float ms = 0;
    ACX_HIP_TRY(hipEventElapsedTime(&ms, evs.a, evs.b));
    if (getenv("ACX_DEBUG")) fprintf(stderr, "[acx_bfs_many] %lld searches (%u general), %llu rounds of <= %u parents, group %.2f ms\n", (long long)n, n_gen, (unsigned long long)rounds, bmax, ms);
    for (int64_t j = 0; j < n; j++) {
        const int64_t k = slot_of[(size_t)j];
        const BfsCursor& c = hc[j];
        uint64_t nodes = c.nodes, expanded = c.expanded;
        uint32_t min_len = c.min_len;
        if (c.status == 1) {
            const Decision& dec = c.term;
            if (dec.err == 0xFE) {
                rc_out[k] = fail(ACX_E_CAPACITY, "acx_search_many: a probe sequence ran through the whole visited table of search %lld", (long long)k);
                continue;
            }
            if (dec.err) {
                rc_out[k] = err_to_rc(dec.err);
                continue;
            }
            min_len = std::min<uint32_t>(min_len, dec.min_len);
            nodes += dec.committed;
            if (dec.solved) {  // success: path of the parent + (action, 2); checked before dedup and before the budget test
                const uint32_t ps = dec.solved_tag / 12, as = dec.solved_tag % 12;
                const int64_t len = (int64_t)pn[(size_t)j];
                if (path_action && path_len) {
                    const int64_t w = std::min<int64_t>(len, path_cap);
                    if (w > 0) {
                        memcpy(path_action + k * path_cap, pa.data() + j * pc, (size_t)w * 4);
                        memcpy(path_len + k * path_cap, pl.data() + j * pc, (size_t)w * 4);
                    }
                    if (len < path_cap) {
                        path_action[k * path_cap + len] = (int32_t)as;
                        path_len[k * path_cap + len] = 2;
                    }
                }
                path_n[k] = len + 1;
                solved[k] = 1;
                expanded += ps + 1;
                min_len = 2;
                if (path_n[k] > path_cap) rc_out[k] = fail(ACX_E_CAPACITY, "path has %lld entries, buffer holds %lld", (long long)path_n[k], (long long)path_cap);
            } else {
                expanded += (uint64_t)dec.p_end + 1;
            }
        } else if (c.status != 3) {
            rc_out[k] = fail(ACX_E_NODEVICE, "bfs cursor of search %lld ended in state %u", (long long)k, c.status);
            continue;
        }
        if (stats) {
            stats[k].nodes = (int64_t)nodes;
            stats[k].expanded = (int64_t)expanded;
            stats[k].children = (int64_t)expanded * 12;
            stats[k].levels = (int64_t)c.batches;
            stats[k].min_len = (int32_t)min_len;
            stats[k].seconds = ms * 1e-3;  // of the whole group
        }
    }
    return ACX_OK;
}

template <typename W> static void launch_greedy_persistent(const GreedyDev<W>& g, GreedyOut* out, hipStream_t st) {
    if (g.nf) hipLaunchKernelGGL((k_greedy_persistent<W, true>), dim3(1), dim3(kGT), 0, st, g, out);
    else hipLaunchKernelGGL((k_greedy_persistent<W, false>), dim3(1), dim3(kGT), 0, st, g, out);
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
    // big buckets go to the whole-GPU kernels of acx_greedy_mega.h (0: the persistent workgroup does everything)
    // (chained hand-offs cost ~95 us a cycle instead of ~120: buckets from 512 parents pay; measured 256 .. 1024: 179.9 / 177.8 / 177.3 / 177.5 / 179.8 ms)
    uint32_t hand_min = getenv("ACX_GREEDY_NO_CHAIN") ? 1024 : 512;
    if (const char* hm = getenv("ACX_GREEDY_HAND_MIN")) hand_min = (uint32_t)strtoul(hm, nullptr, 10);
    uint32_t rank_max = kMegaRankMax;  // handed-off buckets up to this size are ordered by counting (k_gm_rank)
    if (const char* rm = getenv("ACX_MEGA_RANK_MAX")) rank_max = std::max<uint32_t>(256, (uint32_t)strtoul(rm, nullptr, 10));
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
    // ACX_GREEDY_NO_CHAIN=1: round 3's form, one synchronisation per hand-off (A/B runs)
    const bool chain = hand_min && !getenv("ACX_GREEDY_NO_CHAIN");
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
            hipLaunchKernelGGL(k_gm_rank<W>, dim3(1024), dim3(256), 0, st, md, 0u, 1u);
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
    while (o.status == GREEDY_HANDOFF) {
        handoffs++;
        // ---- the selected bucket on the whole GPU: order it, then mega-batches until it is used up, cut, or the search ends ----
        uint32_t live = o.hand_live, counted = 0;  // counted: k_gm_rank has run, the first k_gm_begin places the ids
        if (o.hand_sort) {
            const uint32_t n = live, SC = kMegaRun;
            if (n <= rank_max) {
                hipLaunchKernelGGL(k_gm_rank<W>, dim3(std::min<uint32_t>(1024, ((n + 255) / 256) * ((n + kRankTile - 1) / kRankTile))), dim3(256), 0, st, md, n, 0u);
                counted = n;
            } else {
                hipLaunchKernelGGL(k_gm_runsort<W>, dim3((n + SC - 1) / SC), dim3(kGT), 0, st, md, n, 0u);
                hipLaunchKernelGGL(k_gm_merge<W>, dim3((n + 255) / 256), dim3(256), 0, st, md, n, 0u);
            }
        }
        for (;;) {
            const uint32_t np = std::min<uint32_t>(live, kMegaParents), m = 12u * np;
            uint32_t slots = 1024;
            while (slots < 2 * m) slots <<= 1;
            hipLaunchKernelGGL(k_gm_begin<W>, dim3(std::max<uint32_t>(1, slots / 1024)), dim3(256), 0, st, md, slots, counted, 0u);
            counted = 0;
            hipLaunchKernelGGL(k_gm_expand<W>, dim3((m + 255) / 256), dim3(256), 0, st, md, m, slots - 1, 0u);
            hipLaunchKernelGGL(k_gm_mark<W>, dim3((m + kMegaTile - 1) / kMegaTile), dim3(kMegaTile), 0, st, md, m, 0u);
            hipLaunchKernelGGL(k_gm_decide<W>, dim3(1), dim3(256), 0, st, md, np, m, 0u);
            hipLaunchKernelGGL(k_gm_commit<W>, dim3((m + kMegaTile - 1) / kMegaTile), dim3(kMegaTile), 0, st, md, m, 0u);
            hipLaunchKernelGGL(k_gm_file<W>, dim3(1), dim3(256), 0, st, md, np, 0u);
            hipLaunchKernelGGL(k_gm_push<W>, dim3((m + 255) / 256), dim3(256), 0, st, md, m, 0u);
            uint32_t res[3];  // status, cut, remaining (consecutive in MegaScalars)
            ACX_HIP_TRY(hipMemcpyAsync(res, (const uint8_t*)mscal.p + offsetof(MegaScalars, status), sizeof(res), hipMemcpyDeviceToHost, st));
            // the frontier kernel again, behind the batch and WITHOUT waiting for its outcome: most buckets take one mega-batch,
            // and when this one needs another the launch finds that in the scalars and does nothing (GREEDY_MEGA_MORE) --
            // one synchronisation per hand-off instead of two
            launch_greedy_persistent<W>(g, (GreedyOut*)outb.p, st);
            ACX_HIP_TRY(hipGetLastError());
            ACX_HIP_TRY(hipMemcpyAsync(&o, outb.p, sizeof(o), hipMemcpyDeviceToHost, st));
            ACX_HIP_TRY(hipStreamSynchronize(st));
            const bool more = res[0] == GREEDY_RUNNING && !res[1] && res[2] != 0;
            if (more != (o.status == GREEDY_MEGA_MORE)) return fail(ACX_E_NODEVICE, "greedy hand-off: the frontier kernel and the host disagree about the bucket (status %u)", o.status);
            if (!more) break;
            live = res[2];
        }
    }
    ACX_HIP_TRY(hipEventRecord(ev1, st));
    ACX_HIP_TRY(hipEventSynchronize(ev1));
    float ms = 0;
    ACX_HIP_TRY(hipEventElapsedTime(&ms, ev0, ev1));
    if (hand_min && getenv("ACX_DEBUG")) {
        ACX_HIP_TRY(hipMemcpy(&hstate, stateb.p, sizeof(hstate), hipMemcpyDeviceToHost));
        fprintf(stderr, "[acx_greedy] hand-offs=%llu mega-batches=%llu with %llu parents\n", handoffs, hstate.mega_batches, hstate.mega_parents);
    }
    if (getenv("ACX_DEBUG"))
        fprintf(stderr, "[acx_greedy] status=%u nodes=%u batches=%llu expanded=%llu sorts=%llu big_sorts=%llu max_bucket=%u reason=%u %.3f ms\n", o.status, o.nodes,
                o.batches, o.expanded, o.sorts, o.big_sorts, o.max_bucket, o.fallback_reason, ms);
    if (getenv("ACX_DEBUG")) {
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
    // (verbose searches run greedy batch by batch: the per-improvement lengths need each batch's decision on the host)
    if (greedy && !getenv("ACX_GREEDY_HOST") && !t_minima_on) {  // device-resident priority frontier; falls through when it hits a capacity
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
    // BFS: the stamp table of acx_bfs.h; ACX_BFS_INLINE_TAB=1 selects round 1's inline-key table (A/B measurements)
    const bool stamp = !greedy && !getenv("ACX_BFS_INLINE_TAB") && !getenv("ACX_BFS_CLASSIC_COMMIT");
    // a root in normal form keeps the whole search in normal form: the kernels then run the shorter move code (acx_bfs.h)
    const int move_mode = !is_normal_form<W>(root, cyclical != 0) || getenv("ACX_BFS_GENERAL_MOVE") ? kMoveGeneral : (cyclical ? kMoveNfCyclical : kMoveNf);
    Searcher<W> S;
    int rc = S.init(L, cyclical, max_nodes, bmax, greedy, !greedy && !stamp, false, stamp);
    if (rc) return rc;
    SearchDev<W>& d = S.d;
    hipStream_t st = S.st;
    EventPair evs;
    ACX_HIP_TRY(evs.create());
    hipEvent_t ev0 = evs.a, ev1 = evs.b;
    ACX_HIP_TRY(hipEventRecord(ev0, st));

    const uint32_t tl0 = (uint32_t)(root.n0 + root.n1);
    if (greedy) hipLaunchKernelGGL(k_root<W>, dim3(1), dim3(1), 0, st, d, keyops<W>::make(root.w0, root.n0), keyops<W>::make(root.w1, root.n1), tl0);
    else if (stamp) hipLaunchKernelGGL(k_bfs_root<W>, dim3(1), dim3(1), 0, st, d, keyops<W>::make(root.w0, root.n0), keyops<W>::make(root.w1, root.n1), tl0);
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
    const bool no_runahead = getenv("ACX_BFS_NO_RUNAHEAD") != nullptr;  // A/B: every batch's decision read back, as in round 2
    const bool classic_commit = getenv("ACX_BFS_CLASSIC_COMMIT") != nullptr;  // A/B: mark + scan + decide + commit as four launches
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
#ifndef ACX_BFS_EXPAND_MODE
#define ACX_BFS_EXPAND_MODE(M) M
#endif
#ifndef ACX_BFS_COMPACT_MODE
#define ACX_BFS_COMPACT_MODE(M) M
#endif
#define ACX_BFS_LAUNCH(MODE)                                                                                                                          \
    hipLaunchKernelGGL((k_bfs_expand_insert<W, ACX_BFS_EXPAND_MODE(MODE)>), egrid, eblock, 0, st, d, pbegin, np);                                    \
    hipLaunchKernelGGL(k_bfs_count<W>, cgrid, block, 0, st, d, m, S.d_counts, S.d_masks);                                                             \
    hipLaunchKernelGGL((k_bfs_compact<W, ACX_BFS_COMPACT_MODE(MODE)>), cgrid, block, 0, st, d, pbegin, m, (uint32_t)nodes, (uint32_t)S.cap_nodes, S.d_counts, S.d_masks, S.d_total)
            if (move_mode == kMoveNf) {
                ACX_BFS_LAUNCH(kMoveNf);
            } else if (move_mode == kMoveNfCyclical) {
                ACX_BFS_LAUNCH(kMoveNfCyclical);
            } else {
                ACX_BFS_LAUNCH(kMoveGeneral);
            }
#undef ACX_BFS_LAUNCH
            hipLaunchKernelGGL(k_decide_tab<W>, dim3(1), dim3(1), 0, st, d, m, np, pbegin, (uint32_t)nodes, (uint32_t)S.cap_nodes, (long long)max_nodes, S.d_total, S.d_dec, 1);
        } else {
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
            if (classic_commit) hipLaunchKernelGGL(k_mark_tab<W>, grid, block, 0, st, d, m);
        }
        if (!greedy && !classic_commit) {
            // winners -> nodes in one pass, then the decision from the written nodes
            hipLaunchKernelGGL(k_compact_tab<W>, dim3((m + kCompactTile - 1) / kCompactTile), block, 0, st, d, pbegin, m, (uint32_t)nodes, (uint32_t)S.cap_nodes,
                               (uint32_t)batches, S.d_status, S.d_ticket, S.d_total);
            hipLaunchKernelGGL(k_decide_tab<W>, dim3(1), dim3(1), 0, st, d, m, np, pbegin, (uint32_t)nodes, (uint32_t)S.cap_nodes, (long long)max_nodes, S.d_total, S.d_dec, 1);
        } else {
            size_t tb = S.tmp_bytes;
            if (rocprim::exclusive_scan(S.arena_tmp.p, tb, d.cflag, d.cpos, 0u, m, rocprim::plus<uint32_t>(), st) != hipSuccess)
                return fail(ACX_E_NODEVICE, "rocprim::exclusive_scan failed");
            hipLaunchKernelGGL(k_decide<W>, dim3(1), dim3(1), 0, st, d, m, np, (unsigned long long)nodes, (long long)max_nodes, greedy ? 1 : 0, S.d_dec);
            hipLaunchKernelGGL(k_commit<W>, grid, block, 0, st, d, plist, pbegin, S.d_dec, m, (uint32_t)nodes, greedy ? 1 : 0);
        }
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
    if (L > 61) return fail(ACX_E_INVAL, "acx_search handles max_relator_length <= 61, got %d", L);
    if (max_nodes < 0) max_nodes = 0;
    if (L <= 29) return run_search<uint64_t>(kind, h_presentation, L, max_nodes, cyclical, solved, path_action, path_len, path_cap, path_n, stats);
    return run_search<u128>(kind, h_presentation, L, max_nodes, cyclical, solved, path_action, path_len, path_cap, path_n, stats);
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

extern "C" int acx_release_cached_memory(void) {
    block_pool().trim();
    return ACX_OK;
}

// ------------------------------------------------------------------ many independent searches ----
#include <thread>

extern "C" int acx_search_many(int kind, const int8_t* h_presentations, int64_t n, int L, int64_t max_nodes, int cyclical, int n_threads,
                               int32_t* solved, int32_t* path_action, int32_t* path_len, int64_t path_cap, int64_t* path_n,
                               acx_search_stats* stats, int32_t* rc_out);

extern "C" int acx_search_groups(int kind, int n_groups, const int8_t* const* h_presentations, const int64_t* n, const int32_t* L, int64_t max_nodes, int cyclical,
                                 int32_t* solved, int32_t* path_action, int32_t* path_len, int64_t path_cap, int64_t* path_n, acx_search_stats* stats,
                                 int32_t* rc_out) {
    if (!have_device()) return ACX_E_NODEVICE;
    if (n_groups < 0 || (n_groups && (!h_presentations || !n || !L)) || !solved || !path_n || !rc_out || path_cap < 0)
        return fail(ACX_E_INVAL, "acx_search_groups: bad argument");
    if (kind != ACX_SEARCH_BFS && kind != ACX_SEARCH_GREEDY) return fail(ACX_E_INVAL, "acx_search_groups: bad kind");
    std::vector<int64_t> out0((size_t)n_groups + 1, 0);
    for (int g = 0; g < n_groups; g++) {
        if (n[g] < 0 || L[g] < 1 || (n[g] && !h_presentations[g])) return fail(ACX_E_INVAL, "acx_search_groups: bad group %d", g);
        out0[(size_t)g + 1] = out0[(size_t)g] + n[g];
    }
    if (max_nodes < 0) max_nodes = 0;
    bool sched = kind == ACX_SEARCH_GREEDY && !getenv("ACX_GREEDY_HOST") && !getenv("ACX_GREEDY_MULTI_STATIC") && !t_minima_on && !g_digest_on.load();
    for (int g = 0; g < n_groups; g++) sched = sched && L[g] <= 61;
    if (!sched) {  // one batch after the other through acx_search_many (bfs: a batch fills the GPU by itself, acx_bfs_many.h)
        for (int g = 0; g < n_groups; g++) {
            const int64_t o = out0[(size_t)g];
            const int rc = acx_search_many(kind, h_presentations[g], n[g], L[g], max_nodes, cyclical, 16, solved + o, path_action ? path_action + o * path_cap : nullptr,
                                           path_len ? path_len + o * path_cap : nullptr, path_cap, path_n + o, stats ? stats + o : nullptr, rc_out + o);
            if (rc != ACX_OK) return rc;
        }
        return ACX_OK;
    }
    // greedy_search: ALL batches as jobs of one launch per key width (64-bit keys up to max_relator_length 29, 128-bit above), the two
    // launches side by side
    std::vector<SearchGroupIn> narrow, wide;
    for (int g = 0; g < n_groups; g++)
        if (n[g]) (L[g] <= 29 ? narrow : wide).push_back(SearchGroupIn{h_presentations[g], n[g], L[g], out0[(size_t)g]});
    const int64_t n_all = out0[(size_t)n_groups];
    std::vector<uint8_t> rerun((size_t)n_all, 0);
    int dev = 0;
    (void)hipGetDevice(&dev);
    int rc_wide = ACX_OK;
    std::string err_wide;
    std::thread side;
    // Both widths in flight: the slots are SHARED OUT -- a workgroup of this kernel fills a compute unit and stays until the launch's
    // jobs are used up, so whatever is launched beyond the chip's 256 compute units (the fills of the other launch's tables included)
    // waits for one of them to end.  In proportion to the expected work: a 128-bit search costs ~1.7 x a 64-bit one (measured on the
    // Miller-Schupp sweep: 7.0e7 against 4.2e7 workgroup cycles).
    const uint32_t total = greedy_slots_wanted();
    int64_t n_narrow = 0, n_wide = 0;
    for (const auto& gr : narrow) n_narrow += gr.n;
    for (const auto& gr : wide) n_wide += gr.n;
    uint32_t slots_wide = total, slots_narrow = total;
    if (n_narrow && n_wide) {
        const double share = 1.7 * (double)n_wide / (1.7 * (double)n_wide + (double)n_narrow);
        slots_wide = (uint32_t)std::min<double>(std::max<double>(1.0, share * total + 0.5), (double)total - 1.0);
        slots_narrow = total - slots_wide;
    }
    if (!wide.empty() && !narrow.empty())
        side = std::thread([&]() {
            (void)hipSetDevice(dev);
            rc_wide = run_greedy_sched<u128>(wide, max_nodes, cyclical, solved, path_action, path_len, path_cap, path_n, stats, rc_out, rerun.data(), slots_wide);
            if (rc_wide != ACX_OK) err_wide = acx_last_error();
        });
    int rc = ACX_OK;
    if (!narrow.empty()) rc = run_greedy_sched<uint64_t>(narrow, max_nodes, cyclical, solved, path_action, path_len, path_cap, path_n, stats, rc_out, rerun.data(), slots_narrow);
    if (side.joinable()) side.join();
    else if (!wide.empty()) rc_wide = run_greedy_sched<u128>(wide, max_nodes, cyclical, solved, path_action, path_len, path_cap, path_n, stats, rc_out, rerun.data(), slots_wide);
    if (rc != ACX_OK) return rc;
    if (rc_wide != ACX_OK) return err_wide.empty() ? rc_wide : fail(rc_wide, "%s", err_wide.c_str());
    for (int g = 0; g < n_groups; g++)
        for (int64_t k = 0; k < n[g]; k++) {
            const int64_t o = out0[(size_t)g] + k;
            if (rerun[(size_t)o])  // a search that outgrew a capacity of its workgroup: alone through acx_search
                rc_out[o] = acx_search(kind, h_presentations[g] + k * 2 * L[g], L[g], max_nodes, cyclical, solved + o, path_action ? path_action + o * path_cap : nullptr,
                                       path_len ? path_len + o * path_cap : nullptr, path_cap, path_n + o, stats ? stats + o : nullptr);
        }
    for (int64_t k = 0; k < n_all; k++)
        if (rc_out[k] != ACX_OK && rc_out[k] != ACX_E_CAPACITY) return fail(ACX_E_ROWERR, "acx_search_groups: search %lld failed with code %d", (long long)k, rc_out[k]);
    return ACX_OK;
}

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
        const int64_t group = (int64_t)std::max(1.0, std::min(256.0, group_byte_budget(12e9) / per_search));
        std::vector<uint8_t> rerun((size_t)n, 0);
        if (!getenv("ACX_GREEDY_MULTI_STATIC")) {  // the searches as jobs on a fixed set of workgroup slots (k_greedy_sched); the env switch: round 3's one workgroup per search, launch by launch
            const std::vector<SearchGroupIn> one{SearchGroupIn{h_presentations, n, L, 0}};
            const int rc = L <= 29 ? run_greedy_sched<uint64_t>(one, max_nodes, cyclical, solved, path_action, path_len, path_cap, path_n, stats, rc_out, rerun.data(), greedy_slots_wanted())
                                   : run_greedy_sched<u128>(one, max_nodes, cyclical, solved, path_action, path_len, path_cap, path_n, stats, rc_out, rerun.data(), greedy_slots_wanted());
            if (rc != ACX_OK) return rc;
        } else
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
        for (int64_t k = 0; k < n; k++)
            if (rc_out[k] != ACX_OK && rc_out[k] != ACX_E_CAPACITY) return fail(ACX_E_ROWERR, "acx_search_many: search %lld failed with code %d", (long long)k, rc_out[k]);
        return ACX_OK;
    }
    if (kind == ACX_SEARCH_BFS && n > 1 && L >= 1 && L <= 61 && !getenv("ACX_BFS_MANY_STREAMS") && !t_minima_on && !g_digest_on.load()) {
        // bfs: groups of searches sharing the launches of the fused single search, a batch of every search per round (acx_bfs_many.h);
        // ACX_BFS_MANY=multi: round 3's one persistent workgroup per search in ONE launch (acx_bfs_multi.h), kept for A/B runs
        if (max_nodes < 0) max_nodes = 0;
        const char* many_env = getenv("ACX_BFS_MANY");
        const bool fused = !(many_env && !strcmp(many_env, "multi"));
        const double nn = (double)std::max<int64_t>(max_nodes, 1);
        const double per_search = fused ? (L <= 29 ? 26.0 : 42.0) * nn + 32.0 * (nn + 12.0 * bfs_many_bmax()) + 64.0 * bfs_many_bmax() + 1e6
                                        : (L <= 29 ? 26.0 : 42.0) * nn + 16.0 * 2.0 * nn + 4e6;
        const int64_t group = (int64_t)std::max(1.0, std::min(4096.0, group_byte_budget(48e9) / per_search));
        for (int64_t k0 = 0; k0 < n; k0 += group) {
            const int64_t m = std::min<int64_t>(group, n - k0);
            int32_t* pa = path_action ? path_action + k0 * path_cap : nullptr;
            int32_t* pl = path_len ? path_len + k0 * path_cap : nullptr;
            acx_search_stats* ps = stats ? stats + k0 : nullptr;
            const int8_t* pr = h_presentations + k0 * 2 * L;
            const int rc = fused ? (L <= 29 ? run_bfs_group_fused<uint64_t>(pr, m, L, max_nodes, cyclical, solved + k0, pa, pl, path_cap, path_n + k0, ps, rc_out + k0)
                                            : run_bfs_group_fused<u128>(pr, m, L, max_nodes, cyclical, solved + k0, pa, pl, path_cap, path_n + k0, ps, rc_out + k0))
                                 : (L <= 29 ? run_bfs_group<uint64_t>(pr, m, L, max_nodes, cyclical, solved + k0, pa, pl, path_cap, path_n + k0, ps, rc_out + k0)
                                            : run_bfs_group<u128>(pr, m, L, max_nodes, cyclical, solved + k0, pa, pl, path_cap, path_n + k0, ps, rc_out + k0));
            if (rc != ACX_OK) return rc;
        }
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
    };
    std::vector<std::thread> pool;
    for (int t = 0; t < n_threads; t++) pool.emplace_back(work);
    for (auto& t : pool) t.join();
    for (int64_t k = 0; k < n; k++)
        if (rc_out[k] != ACX_OK && rc_out[k] != ACX_E_CAPACITY) return fail(ACX_E_ROWERR, "acx_search_many: search %lld failed with code %d", (long long)k, rc_out[k]);
    return ACX_OK;
}
