// acx_searcher.h -- what the translation units of the single-GPU searches share (acx_search.hip: one search; acx_search_greedy.hip: the
// device-resident greedy frontier of one search; acx_search_many.hip: many searches per call): the per-search device arenas
// (Searcher), the pooled streams and events, the node digest and verbose-minima hooks, and the library's options.
#pragma once
#include <atomic>
#include <chrono>
#include <string>
#include <thread>

#include "acx_frontier.h"

namespace acx {

// ---- options (include/acx.h: acx_set_option) ---------------------------------------------------------------------------------
// Tuning and test knobs are set through the C ABI; nothing on a call path reads the environment.  -1 = the built-in default.
extern std::atomic<int64_t> g_options[ACX_OPT_COUNT];
inline int64_t option(int which, int64_t dflt) {
    const int64_t v = g_options[which].load(std::memory_order_relaxed);
    return v < 0 ? dflt : v;
}
// ACX_DEBUG in the environment when the library was loaded: diagnostics on stderr (read once, acx_search.hip)
extern const bool g_debug;

// ---- node-arena digest (repeat-determinism tests) ------------------------------------------------------------------------
// acx_search_digest_enable(1) makes every search of this process finish with one extra pass that folds (id, key, parent,
// action) of all its nodes into a 64-bit sum; acx_search_last_digest returns the calling thread's last one.
extern std::atomic<int> g_digest_on;
extern thread_local uint64_t t_last_digest;
// acx_search_minima_enable(1): every search records the total lengths at which the reference's verbose mode prints "New
// minimal length found" (breadth_first.py:79-82, greedy.py:85-89): each child, in generation order, that is shorter than
// everything generated before it, up to the child that ends the search.  acx_search_last_minima returns the sequence.
extern thread_local int t_minima_on;  // per calling thread: a verbose search must not slow down or lose the lines of searches on other threads
extern thread_local std::vector<int32_t> t_last_minima;
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
int search_handles_take(SearchHandles& h);
void search_handles_give(SearchHandles& h);

template <typename W> struct Searcher {
    SearchDev<W> d;
    StampBuf stamp_tab_buf;  // the fused BFS's stamp table: handed from search to search with a new epoch, no refill (acx_frontier.h)
    DevBuf arena_nodes, arena_cand, arena_tab, arena_btab, arena_scal, arena_tmp, arena_list, arena_path, arena_status, arena_first, arena_cursor;
    uint8_t* h_cursor = nullptr;  // pinned: kRunAheadSlots x BfsCursor (behind the Decision staging)
    SearchHandles handles;
    hipEvent_t ev_cursor[kRunAheadSlots] = {}, ev_batch[kRunAheadSlots] = {};  // (copies of the handles' events)
    hipStream_t st_copy = nullptr;  // the cursor snapshots travel on a stream of their own: a copy queued on `st` sits between two batches (10 us)
    unsigned long long h_first[kFirstLen];
    uint32_t* d_counts = nullptr;            // stamp-table BFS: winners per tile (k_bfs_count -> k_bfs_compact) ...
    uint32_t* d_masks = nullptr;             // ... and one winner bit per candidate
    uint32_t* d_total = nullptr;
    size_t tmp_bytes = 0;
    uint64_t cap_nodes = 0, cap_cand = 0, n_slots = 0, n_bslots = 0;
    hipStream_t st = nullptr;
    Decision* d_dec = nullptr;   // device
    uint8_t* h_pin = nullptr;    // pinned host staging: Decision followed by the total lengths of the new nodes
    size_t h_pin_bytes = 0;

    ~Searcher() { search_handles_give(handles); }

    // stamp_tab: the 8-byte stamp table of the fused BFS (acx_bfs.h: no candidate keys at all); otherwise the id table of the
    // batch-per-launch greedy path; lean: no key arrays and no table (the persistent greedy frontier keeps its own: GreedyDev::nkeys / tab)
    int init(int L, int cyclical, int64_t max_nodes, uint32_t batch_parents, bool greedy, bool lean = false, bool stamp_tab = false) {
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
            if (stamp_tab) {  // byte flags of the fused BFS
                d.btook = take(b, cap_cand + 8);
                d.brepl = take(b, cap_cand + 8);
            }
            if (pass == 0 && arena_cand.alloc(o)) return ACX_E_NOMEM;
        }
        if (lean) {
            d.slots = nullptr;
        } else if (stamp_tab) {
            if (int rc = stamp_tab_buf.alloc(n_slots * 8, st)) return rc;
            d.stab = (unsigned long long*)stamp_tab_buf.p;
            d.epoch = stamp_tab_buf.epoch;
            d.stmask = (uint32_t)(n_slots - 1);
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
        if (!stamp_tab) {  // temporary storage of the scan over one batch
            size_t need = 0;
            if (int rc = scan_u32_exclusive(nullptr, &need, d.cflag, d.cpos, cap_cand, st)) return rc;
            tmp_bytes = need + 256;
            if (arena_tmp.alloc(tmp_bytes)) return ACX_E_NOMEM;
        }
        if (!lean && !stamp_tab) ACX_HIP_TRY(hipMemsetAsync(arena_tab.p, 0xff, n_slots * 4, st));
        ACX_HIP_TRY(hipMemsetAsync(arena_scal.p, 0xff, 256, st));
        ACX_HIP_TRY(hipMemsetAsync(d.err, 0, 4, st));
        if (stamp_tab) ACX_HIP_TRY(hipMemsetAsync(d.brepl, 0, cap_cand + 8, st));  // once: k_bfs_compact zeroes what a batch sets
        d_total = (uint32_t*)(sc + 132);
        ACX_HIP_TRY(hipMemsetAsync(sc + 128, 0, 8, st));
        if (stamp_tab) {  // k_bfs_count -> k_bfs_compact: winners per tile and one winner bit per candidate
            const size_t tiles = cap_cand / kCompactTile + 2;
            if (arena_status.alloc(tiles * (4 + 1024))) return ACX_E_NOMEM;
            d_counts = (uint32_t*)arena_status.p;
            d_masks = d_counts + tiles;
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

inline int err_to_rc(uint32_t e) {
    return fail(ACX_E_ROWERR, "a move emptied a relator during the search: the reference raises %s here",
                (e & ACX_ERR_INDEX) && !(e & ACX_ERR_ASSERT) ? "IndexError" : "AssertionError");
}

// greedy_search of ONE presentation on the device-resident priority frontier (acx_search_greedy.hip; instantiated there for both key
// widths).  *handled = false when the persistent kernel ran out of one of its capacities: the caller reruns the search batch by batch.
template <typename W>
int run_greedy_device(const Pres<W>& root, int L, int64_t max_nodes, int cyclical, int32_t* solved, int32_t* path_action, int32_t* path_len, int64_t path_cap,
                      int64_t* path_n, acx_search_stats* stats, bool* handled);

}  // namespace acx
