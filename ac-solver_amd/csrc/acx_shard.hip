// acx_shard.hip -- per-GPU engine of the sharded BFS frontier (multi-GPU form of bfs, breadth_first.py:15-97) and its
// C ABI (acx_shard_*).  Orchestration across ranks: ac-solver_amd/ac_solver/search/sharded.py.
#include "acx_frontier.h"

namespace acx {

// =============================================================================================
// Sharded frontier: one engine per GPU, states partitioned by hash(key) mod world.  The host side
// (ac_solver/search/sharded.py) moves candidate records between ranks with an RCCL all-to-all and
// broadcasts winner tags; everything per rank happens in the kernels below.  A record is KW+2 int64:
// the key words, tag = 12 * global_parent_position + action, parent_ref = rank << 40 | local id.
// =============================================================================================
template <typename W> struct recio;
template <> struct recio<uint64_t> {
    static constexpr int KW = 2;
    static ACX_HD void put(int64_t* r, uint64_t k0, uint64_t k1) { r[0] = (int64_t)k0; r[1] = (int64_t)k1; }
    static ACX_HD void get(const int64_t* r, uint64_t& k0, uint64_t& k1) { k0 = (uint64_t)r[0]; k1 = (uint64_t)r[1]; }
};
template <> struct recio<u128> {
    static constexpr int KW = 4;
    static ACX_HD void put(int64_t* r, u128 k0, u128 k1) {
        r[0] = (int64_t)(uint64_t)k0; r[1] = (int64_t)(uint64_t)(k0 >> 64);
        r[2] = (int64_t)(uint64_t)k1; r[3] = (int64_t)(uint64_t)(k1 >> 64);
    }
    static ACX_HD void get(const int64_t* r, u128& k0, u128& k1) {
        k0 = ((u128)(uint64_t)r[1] << 64) | (uint64_t)r[0];
        k1 = ((u128)(uint64_t)r[3] << 64) | (uint64_t)r[2];
    }
};

template <typename W>
__global__ void __launch_bounds__(256) k_shard_expand(SearchDev<W> d, const int64_t* __restrict__ ids, const int64_t* __restrict__ gpos, int64_t np,
                                                      int64_t pref_hi, int64_t* __restrict__ rec, unsigned long long* __restrict__ solved) {
    ACX_VGPR_PAD_W(W, "v39", "v55");
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 12 * np) return;
    const int64_t p = t / 12;
    const int a = (int)(t - 12 * p);
    const int64_t id = ids[p];
    Pres<W> s;
    key_to_pres<W>(d.k0[id], d.k1[id], s);
    const int e = apply_move<W, kSearchSafe>(s, a, d.L, d.cyclical != 0);
    if (e) atomicMin(solved + 1, ((unsigned long long)(12 * gpos[p] + a) << 8) | (unsigned long long)e);  // first erroring move (global tag)
    int64_t* r = rec + t * (recio<W>::KW + 2);
    recio<W>::put(r, keyops<W>::make(s.w0, s.n0), keyops<W>::make(s.w1, s.n1));
    const int64_t tag = 12 * gpos[p] + a;
    r[recio<W>::KW] = tag;
    r[recio<W>::KW + 1] = pref_hi | id;
    if (s.n0 + s.n1 == 2) atomicMin(solved, (unsigned long long)tag);
    if ((uint32_t)(s.n0 + s.n1) < *(volatile uint32_t*)d.min_len) atomicMin(d.min_len, (uint32_t)(s.n0 + s.n1));
}

// Owner rank of a packed key: the arithmetic of ac_solver/search/sharded.py:owner_of on the key's int64 words.
ACX_HD uint64_t owner_mix(uint64_t h, uint64_t w) {
    h = (h ^ w) * 0x9E3779B97F4A7C15ull;
    return h ^ (h >> 29);
}
ACX_HD uint32_t owner_of_key(uint64_t k0, uint64_t k1, uint32_t world) {
    const uint64_t h = owner_mix(owner_mix(0, k0), k1);
    return (uint32_t)((h & 0x7FFFFFFFFFFFFFFFull) % world);
}
ACX_HD uint32_t owner_of_key(u128 k0, u128 k1, uint32_t world) {
    uint64_t h = owner_mix(owner_mix(0, (uint64_t)k0), (uint64_t)(k0 >> 64));
    h = owner_mix(owner_mix(h, (uint64_t)k1), (uint64_t)(k1 >> 64));
    return (uint32_t)((h & 0x7FFFFFFFFFFFFFFFull) % world);
}

// k_shard_expand + routing: the record of a child goes straight into the send region of the rank that owns the
// child's key (region o = rec[o * region_cap ...], filled through a wave-aggregated cursor counts[o]), so the
// all-to-all can leave without a sort by owner.  The order inside a region is arbitrary (the receiver orders by tag).
template <typename W>
__global__ void __launch_bounds__(1024) k_shard_expand_routed(SearchDev<W> d, const int64_t* __restrict__ ids, const int64_t* __restrict__ gpos, int64_t np,
                                                             int64_t pref_hi, uint32_t world, int64_t* __restrict__ rec, int64_t region_cap,
                                                             unsigned long long* __restrict__ counts, unsigned long long* __restrict__ solved) {
    ACX_VGPR_PAD_W(W, "v39", "v55");
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = t < 12 * np;
    const uint32_t lane = threadIdx.x & 63;
    W k0 = 0, k1 = 0;
    int64_t tag = 0, pref = 0;
    uint32_t owner = 0xFFFFFFFFu;
    if (active) {
        const int64_t p = t / 12;
        const int a = (int)(t - 12 * p);
        const int64_t id = ids[p];
        Pres<W> s;
        const W pk0 = d.k0[id], pk1 = d.k1[id];
        key_to_pres<W>(pk0, pk1, s);
        const int e = apply_move<W, kSearchSafe>(s, a, d.L, d.cyclical != 0);
        if (e) atomicMin(solved + 1, ((unsigned long long)(12 * gpos[p] + a) << 8) | (unsigned long long)e);  // first erroring move (global tag)
        k0 = keyops<W>::make(s.w0, s.n0);
        k1 = keyops<W>::make(s.w1, s.n1);
        tag = 12 * gpos[p] + a;
        pref = pref_hi | id;
        // a move that leaves the state unchanged (over-long product: ac_moves.py:64, :126) yields the parent itself, which
        // is in the visited set already: such a child can never be new, so it is not sent at all
        if (k0 != pk0 || k1 != pk1) owner = owner_of_key(k0, k1, world);
        if (s.n0 + s.n1 == 2) atomicMin(solved, (unsigned long long)tag);
        if ((uint32_t)(s.n0 + s.n1) < *(volatile uint32_t*)d.min_len) atomicMin(d.min_len, (uint32_t)(s.n0 + s.n1));
    }
    // position inside the destination region: wave-aggregated LDS counters per owner, then ONE global atomicAdd per
    // (workgroup, owner) -- per-wave global atomics on `world` addresses serialise (1.6 ms per 12 M children)
    __shared__ uint32_t s_cnt[64];
    __shared__ unsigned long long s_base[64];
    if (threadIdx.x < 64) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    uint32_t pos_in_block = 0;
    for (uint32_t o = 0; o < world; o++) {
        const unsigned long long m = __ballot(owner == o);
        if (!m) continue;
        const uint32_t lead = (uint32_t)__builtin_ctzll(m);
        uint32_t base = 0;
        if (lane == lead) base = atomicAdd(&s_cnt[o], (uint32_t)__popcll(m));
        base = (uint32_t)__shfl((int)base, (int)lead);
        if (owner == o) pos_in_block = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    }
    __syncthreads();
    if (threadIdx.x < world && s_cnt[threadIdx.x]) s_base[threadIdx.x] = atomicAdd(&counts[threadIdx.x], (unsigned long long)s_cnt[threadIdx.x]);
    __syncthreads();
    if (owner != 0xFFFFFFFFu) {
        const int64_t pos = (int64_t)s_base[owner] + pos_in_block;
        if (pos < region_cap) {
            int64_t* r = rec + ((int64_t)owner * region_cap + pos) * (recio<W>::KW + 2);
            recio<W>::put(r, k0, k1);
            r[recio<W>::KW] = tag;
            r[recio<W>::KW + 1] = pref;
        }  // an overflow shows in counts[o] > region_cap; the host reports it
    }
}

template <typename W> __global__ void __launch_bounds__(256) k_shard_tags(const int64_t* __restrict__ rec, int64_t n, uint64_t* __restrict__ tags, uint32_t* __restrict__ idx) {
    ACX_VGPR_PAD("v23");
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    tags[i] = (uint64_t)rec[i * (recio<W>::KW + 2) + recio<W>::KW];
    idx[i] = (uint32_t)i;
}

// candidate arena in tag order: j-th smallest tag -> slot j
template <typename W>
__global__ void __launch_bounds__(256) k_shard_gather(SearchDev<W> d, const int64_t* __restrict__ rec, const uint64_t* __restrict__ tags_sorted,
                                                      const uint32_t* __restrict__ idx_sorted, int64_t n, int64_t* __restrict__ ctag, int64_t* __restrict__ cpref) {
    ACX_VGPR_PAD_W(W, "v31", "v39");
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const int64_t* r = rec + (int64_t)idx_sorted[j] * (recio<W>::KW + 2);
    W k0, k1;
    recio<W>::get(r, k0, k1);
    d.ck0[j] = k0;
    d.ck1[j] = k1;
    d.clen[j] = (uint8_t)(keyops<W>::len(k0) + keyops<W>::len(k1));
    d.cslot[j] = 0;  // "replaced by a smaller tag" flag of k_insert_tab
    ctag[j] = (int64_t)tags_sorted[j];
    cpref[j] = r[recio<W>::KW + 1];
}

template <typename W>
__global__ void __launch_bounds__(256) k_shard_win_tags(SearchDev<W> d, const int64_t* __restrict__ ctag, int64_t n, int64_t* __restrict__ out) {
    ACX_VGPR_PAD("v23");
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n || !d.cflag[j]) return;
    out[d.cpos[j]] = ctag[j];
}

template <typename W>
__global__ void __launch_bounds__(256) k_shard_commit(SearchDev<W> d, const int64_t* __restrict__ ctag, const int64_t* __restrict__ cpref, int64_t n,
                                                      int64_t cutoff, uint32_t base, int64_t* __restrict__ node_pref) {
    ACX_VGPR_PAD("v31");
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n || !d.cflag[j] || ctag[j] >= cutoff) return;
    const uint32_t id = base + d.cpos[j];
    d.k0[id] = d.ck0[j];
    d.k1[id] = d.ck1[j];
    d.act[id] = (uint8_t)(ctag[j] % 12);
    d.tlen[id] = d.clen[j];
    node_pref[id] = cpref[j];
}

// number of winners with tag < cutoff: candidates are in tag order, so it is cpos at the first tag >= cutoff
template <typename W> __global__ void k_shard_count(SearchDev<W> d, const int64_t* __restrict__ ctag, int64_t n, int64_t cutoff, uint32_t* __restrict__ count) {
    ACX_VGPR_PAD("v15");
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (ctag[mid] < cutoff) lo = mid + 1;
        else hi = mid;
    }
    *count = lo >= n ? d.cpos[n - 1] + d.cflag[n - 1] : d.cpos[lo];
}

template <typename W> struct ShardEngine {
    SearchDev<W> d;
    DevBuf nodes_buf, cand_buf, tab_buf, scal_buf, tmp_buf, sort_buf;
    int64_t* node_pref = nullptr;  // [cap] parent_ref of every local node
    int64_t* ctag = nullptr;
    int64_t* cpref = nullptr;
    uint64_t* tags_in = nullptr;
    uint64_t* tags_sorted = nullptr;
    uint32_t* idx_in = nullptr;
    uint32_t* idx_sorted = nullptr;
    uint32_t* commit_count = nullptr;
    size_t scan_tmp = 0, sort_tmp = 0;
    uint64_t cap_nodes = 0, cap_cand = 0, n_slots = 0;
    uint64_t nodes = 0;    // committed local nodes
    uint32_t epoch = 0;    // insert calls so far (stamps of the inline-key table)
    int64_t pending = 0;   // candidates of the last insert (awaiting commit)
    int rank = 0, world = 1;

    int init(int L, int cyclical, int64_t node_cap, int64_t batch_cap, int rank_, int world_) {
        memset(&d, 0, sizeof(d));
        d.L = L;
        d.cyclical = cyclical;
        rank = rank_;
        world = world_;
        cap_nodes = (uint64_t)node_cap + 64;
        cap_cand = (uint64_t)std::max<int64_t>(batch_cap, 1024);
        n_slots = 1024;
        while (n_slots < 2 * (cap_nodes + cap_cand)) n_slots <<= 1;
        if (n_slots > (1ull << 31)) return fail(ACX_E_INVAL, "acx_shard: capacity too large for 32-bit node ids");
        size_t o = 0;
        auto take = [&](uint8_t* base, size_t bytes) {
            uint8_t* p = base ? base + o : nullptr;
            o += (bytes + 255) / 256 * 256;
            return p;
        };
        for (int pass = 0; pass < 2; pass++) {
            uint8_t* b = (uint8_t*)nodes_buf.p;
            o = 0;
            d.k0 = (W*)take(b, cap_nodes * sizeof(W));
            d.k1 = (W*)take(b, cap_nodes * sizeof(W));
            node_pref = (int64_t*)take(b, cap_nodes * 8);
            d.act = (uint8_t*)take(b, cap_nodes);
            d.tlen = (uint8_t*)take(b, cap_nodes);
            if (pass == 0 && nodes_buf.alloc(o)) return ACX_E_NOMEM;
        }
        for (int pass = 0; pass < 2; pass++) {
            uint8_t* b = (uint8_t*)cand_buf.p;
            o = 0;
            d.ck0 = (W*)take(b, cap_cand * sizeof(W));
            d.ck1 = (W*)take(b, cap_cand * sizeof(W));
            ctag = (int64_t*)take(b, cap_cand * 8);
            cpref = (int64_t*)take(b, cap_cand * 8);
            tags_in = (uint64_t*)take(b, cap_cand * 8);
            tags_sorted = (uint64_t*)take(b, cap_cand * 8);
            idx_in = (uint32_t*)take(b, cap_cand * 4);
            idx_sorted = (uint32_t*)take(b, cap_cand * 4);
            d.cslot = (uint32_t*)take(b, cap_cand * 4);
            d.cflag = (uint32_t*)take(b, cap_cand * 4);
            d.cpos = (uint32_t*)take(b, cap_cand * 4);
            d.clen = (uint8_t*)take(b, cap_cand);
            if (pass == 0 && cand_buf.alloc(o)) return ACX_E_NOMEM;
        }
        if (tab_buf.alloc(n_slots * sizeof(TabEntry<W>))) return ACX_E_NOMEM;
        d.tab = (TabEntry<W>*)tab_buf.p;
        d.tmask = (uint32_t)(n_slots - 1);
        if (scal_buf.alloc(256)) return ACX_E_NOMEM;
        uint8_t* sc = (uint8_t*)scal_buf.p;
        d.err = (uint32_t*)(sc + 24);
        d.min_len = (uint32_t*)(sc + 28);
        commit_count = (uint32_t*)(sc + 32);
        if (rocprim::exclusive_scan(nullptr, scan_tmp, d.cflag, d.cpos, 0u, cap_cand, rocprim::plus<uint32_t>(), (hipStream_t) nullptr) != hipSuccess)
            return fail(ACX_E_NODEVICE, "rocprim::exclusive_scan sizing failed");
        if (rocprim::radix_sort_pairs(nullptr, sort_tmp, tags_in, tags_sorted, idx_in, idx_sorted, cap_cand, 0, 64, (hipStream_t) nullptr) != hipSuccess)
            return fail(ACX_E_NODEVICE, "rocprim::radix_sort_pairs sizing failed");
        if (tmp_buf.alloc(std::max(scan_tmp, sort_tmp) + 256)) return ACX_E_NOMEM;
        ACX_HIP_TRY(hipMemset(d.tab, 0xff, n_slots * sizeof(TabEntry<W>)));
        ACX_HIP_TRY(hipMemset(scal_buf.p, 0xff, 256));
        ACX_HIP_TRY(hipMemset(d.err, 0, 4));
        ACX_HIP_TRY(hipDeviceSynchronize());  // the fills run on the null stream; the engine's calls arrive on the caller's (possibly non-blocking) stream
        return ACX_OK;
    }
};

struct ShardAny {
    bool wide;
    ShardEngine<uint64_t>* e64 = nullptr;
    ShardEngine<u128>* e128 = nullptr;
};

#define ACX_SHARD_DISPATCH(h, ...)                   \
    do {                                             \
        if ((h)->wide) {                             \
            typedef u128 W;                          \
            auto& E = *(h)->e128;                    \
            (void)sizeof(W);                         \
            __VA_ARGS__;                             \
        } else {                                     \
            typedef uint64_t W;                      \
            auto& E = *(h)->e64;                     \
            (void)sizeof(W);                         \
            __VA_ARGS__;                             \
        }                                            \
    } while (0)

template <typename W> static int shard_root(ShardEngine<W>& E, const int8_t* pres, int64_t* rec) {
    Pres<W> root;
    bool ok = pack_relator<W>(pres, E.d.L, root.w0, root.n0);
    ok = pack_relator<W>(pres + E.d.L, E.d.L, root.w1, root.n1) && ok;
    if (!ok) return fail(ACX_E_ROWERR, "acx_shard: the presentation is not a zero-padded word pair over {+-1,+-2}");
    recio<W>::put(rec, keyops<W>::make(root.w0, root.n0), keyops<W>::make(root.w1, root.n1));
    rec[recio<W>::KW] = 0;
    rec[recio<W>::KW + 1] = -1;
    return ACX_OK;
}

template <typename W> static int shard_expand(ShardEngine<W>& E, const int64_t* ids, const int64_t* gpos, int64_t np, int64_t* rec, int64_t* solved, hipStream_t st) {
    if (np <= 0) return ACX_OK;
    const int64_t m = 12 * np;
    hipLaunchKernelGGL(k_shard_expand<W>, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, E.d, ids, gpos, np, (int64_t)E.rank << 40, rec,
                       (unsigned long long*)solved);
    ACX_HIP_TRY(hipGetLastError());
    return ACX_OK;
}

template <typename W>
static int shard_expand_routed(ShardEngine<W>& E, const int64_t* ids, const int64_t* gpos, int64_t np, int64_t* rec, int64_t region_cap, int64_t* counts,
                               int64_t* solved, hipStream_t st) {
    ACX_HIP_TRY(hipMemsetAsync(counts, 0, (size_t)E.world * 8, st));
    if (np <= 0) return ACX_OK;
    const int64_t m = 12 * np;
    if (E.world > 64) return fail(ACX_E_INVAL, "acx_shard_expand_routed handles world <= 64");
    hipLaunchKernelGGL(k_shard_expand_routed<W>, dim3((unsigned)((m + 1023) / 1024)), dim3(1024), 0, st, E.d, ids, gpos, np, (int64_t)E.rank << 40, (uint32_t)E.world,
                       rec, region_cap, (unsigned long long*)counts, (unsigned long long*)solved);
    ACX_HIP_TRY(hipGetLastError());
    return ACX_OK;
}

template <typename W> static int shard_insert(ShardEngine<W>& E, const int64_t* rec, int64_t n, int tag_bits, int64_t* win_tags, int64_t* n_win, hipStream_t st) {
    *n_win = 0;
    E.pending = n;
    if (n <= 0) return ACX_OK;
    if ((uint64_t)n > E.cap_cand) return fail(ACX_E_CAPACITY, "acx_shard_insert: %lld records exceed the batch capacity %llu", (long long)n, (unsigned long long)E.cap_cand);
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    hipLaunchKernelGGL(k_shard_tags<W>, grid, block, 0, st, rec, n, E.tags_in, E.idx_in);
    size_t tb = E.sort_tmp;
    const unsigned end_bit = (unsigned)(tag_bits < 1 ? 64 : (tag_bits > 64 ? 64 : tag_bits));  // tags < 2^tag_bits: fewer radix passes
    if (rocprim::radix_sort_pairs(E.tmp_buf.p, tb, E.tags_in, E.tags_sorted, E.idx_in, E.idx_sorted, (size_t)n, 0, end_bit, st) != hipSuccess)
        return fail(ACX_E_NODEVICE, "rocprim::radix_sort_pairs failed");
    hipLaunchKernelGGL(k_shard_gather<W>, grid, block, 0, st, E.d, rec, E.tags_sorted, E.idx_sorted, n, E.ctag, E.cpref);
    E.epoch++;
    hipLaunchKernelGGL(k_insert_tab<W>, grid, block, 0, st, E.d, (uint32_t)n, E.epoch, 0);
    hipLaunchKernelGGL(k_mark_tab<W>, grid, block, 0, st, E.d, (uint32_t)n);
    tb = E.scan_tmp;
    if (rocprim::exclusive_scan(E.tmp_buf.p, tb, E.d.cflag, E.d.cpos, 0u, (size_t)n, rocprim::plus<uint32_t>(), st) != hipSuccess)
        return fail(ACX_E_NODEVICE, "rocprim::exclusive_scan failed");
    hipLaunchKernelGGL(k_shard_win_tags<W>, grid, block, 0, st, E.d, E.ctag, n, win_tags);
    ACX_HIP_TRY(hipGetLastError());
    uint32_t last[2];
    ACX_HIP_TRY(hipMemcpyAsync(&last[0], E.d.cpos + (n - 1), 4, hipMemcpyDeviceToHost, st));
    ACX_HIP_TRY(hipMemcpyAsync(&last[1], E.d.cflag + (n - 1), 4, hipMemcpyDeviceToHost, st));
    ACX_HIP_TRY(hipStreamSynchronize(st));
    *n_win = (int64_t)last[0] + last[1];
    return ACX_OK;
}

template <typename W> static int shard_commit(ShardEngine<W>& E, int64_t cutoff, int64_t* first_id, int64_t* n_committed, hipStream_t st) {
    *first_id = (int64_t)E.nodes;
    *n_committed = 0;
    const int64_t n = E.pending;
    E.pending = 0;
    if (n <= 0) return ACX_OK;
    hipLaunchKernelGGL(k_shard_count<W>, dim3(1), dim3(1), 0, st, E.d, E.ctag, n, cutoff, E.commit_count);
    hipLaunchKernelGGL(k_shard_commit<W>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, E.d, E.ctag, E.cpref, n, cutoff, (uint32_t)E.nodes,
                       E.node_pref);
    ACX_HIP_TRY(hipGetLastError());
    uint32_t c = 0;
    ACX_HIP_TRY(hipMemcpyAsync(&c, E.commit_count, 4, hipMemcpyDeviceToHost, st));
    ACX_HIP_TRY(hipStreamSynchronize(st));
    if (E.nodes + c > E.cap_nodes) return fail(ACX_E_CAPACITY, "acx_shard_commit: node capacity exceeded");
    E.nodes += c;
    *n_committed = c;
    return ACX_OK;
}

template <typename W> static int shard_node_info(ShardEngine<W>& E, int64_t id, int64_t* info) {
    if (id < 0 || (uint64_t)id >= E.nodes) return fail(ACX_E_INVAL, "acx_shard_node_info: id out of range");
    uint8_t a = 0, l = 0;
    int64_t pr = 0;
    ACX_HIP_TRY(hipDeviceSynchronize());
    ACX_HIP_TRY(hipMemcpy(&a, E.d.act + id, 1, hipMemcpyDeviceToHost));
    ACX_HIP_TRY(hipMemcpy(&l, E.d.tlen + id, 1, hipMemcpyDeviceToHost));
    ACX_HIP_TRY(hipMemcpy(&pr, E.node_pref + id, 8, hipMemcpyDeviceToHost));
    info[0] = pr < 0 ? -1 : (int64_t)a;
    info[1] = l;
    info[2] = pr;
    return ACX_OK;
}

}  // namespace acx

using namespace acx;

// ------------------------------------------------------------------ sharded frontier: C ABI ----
struct acx_shard {
    acx::ShardAny any;
};

extern "C" {

int acx_shard_key_words(int L) { return L <= 29 ? 2 : 4; }

acx_shard* acx_shard_create(int L, int cyclical, int64_t node_cap, int64_t batch_cap, int rank, int world) {
    if (!have_device()) return nullptr;
    if (L < 1 || L > 61 || node_cap < 1 || batch_cap < 1 || world < 1 || rank < 0 || rank >= world) {
        fail(ACX_E_INVAL, "acx_shard_create: bad argument (1 <= L <= 61)");
        return nullptr;
    }
    acx_shard* h = new (std::nothrow) acx_shard();
    if (!h) return nullptr;
    h->any.wide = L > 29;
    int rc;
    if (h->any.wide) {
        h->any.e128 = new ShardEngine<u128>();
        rc = h->any.e128->init(L, cyclical, node_cap, batch_cap, rank, world);
    } else {
        h->any.e64 = new ShardEngine<uint64_t>();
        rc = h->any.e64->init(L, cyclical, node_cap, batch_cap, rank, world);
    }
    if (rc != ACX_OK) {
        delete h->any.e64;
        delete h->any.e128;
        delete h;
        return nullptr;
    }
    return h;
}

void acx_shard_destroy(acx_shard* h) {
    if (!h) return;
    delete h->any.e64;
    delete h->any.e128;
    delete h;
}

int acx_shard_root_record(acx_shard* h, const int8_t* h_presentation, int64_t* h_record) {
    if (!h || !h_presentation || !h_record) return fail(ACX_E_INVAL, "acx_shard_root_record: bad argument");
    ACX_SHARD_DISPATCH(&h->any, return shard_root<W>(E, h_presentation, h_record));
}

int acx_shard_expand(acx_shard* h, const int64_t* d_ids, const int64_t* d_gpos, int64_t np, int64_t* d_records, int64_t* d_solved, void* stream) {
    if (!h || np < 0 || (np > 0 && (!d_ids || !d_gpos || !d_records || !d_solved))) return fail(ACX_E_INVAL, "acx_shard_expand: bad argument");
    ACX_SHARD_DISPATCH(&h->any, return shard_expand<W>(E, d_ids, d_gpos, np, d_records, d_solved, (hipStream_t)stream));
}

int acx_shard_expand_routed(acx_shard* h, const int64_t* d_ids, const int64_t* d_gpos, int64_t np, int64_t* d_records, int64_t region_cap,
                            int64_t* d_counts, int64_t* d_solved, void* stream) {
    if (!h || np < 0 || region_cap < 0 || !d_counts || (np > 0 && (!d_ids || !d_gpos || !d_records || !d_solved)))
        return fail(ACX_E_INVAL, "acx_shard_expand_routed: bad argument");
    ACX_SHARD_DISPATCH(&h->any, return shard_expand_routed<W>(E, d_ids, d_gpos, np, d_records, region_cap, d_counts, d_solved, (hipStream_t)stream));
}

int acx_shard_insert(acx_shard* h, const int64_t* d_records, int64_t n, int tag_bits, int64_t* d_win_tags, int64_t* n_win, void* stream) {
    if (!h || n < 0 || !n_win || (n > 0 && (!d_records || !d_win_tags))) return fail(ACX_E_INVAL, "acx_shard_insert: bad argument");
    ACX_SHARD_DISPATCH(&h->any, return shard_insert<W>(E, d_records, n, tag_bits, d_win_tags, n_win, (hipStream_t)stream));
}

int acx_shard_commit(acx_shard* h, int64_t cutoff_tag, int64_t* first_id, int64_t* n_committed, void* stream) {
    if (!h || !first_id || !n_committed) return fail(ACX_E_INVAL, "acx_shard_commit: bad argument");
    ACX_SHARD_DISPATCH(&h->any, return shard_commit<W>(E, cutoff_tag, first_id, n_committed, (hipStream_t)stream));
}

int acx_shard_node_info(acx_shard* h, int64_t id, int64_t* h_info3) {
    if (!h || !h_info3) return fail(ACX_E_INVAL, "acx_shard_node_info: bad argument");
    ACX_SHARD_DISPATCH(&h->any, return shard_node_info<W>(E, id, h_info3));
}

int64_t acx_shard_node_count(acx_shard* h) {
    if (!h) return 0;
    return h->any.wide ? (int64_t)h->any.e128->nodes : (int64_t)h->any.e64->nodes;
}

int acx_shard_status(acx_shard* h, int32_t* err, int32_t* min_len) {
    if (!h || !err || !min_len) return fail(ACX_E_INVAL, "acx_shard_status: bad argument");
    uint32_t v[2];
    ACX_HIP_TRY(hipDeviceSynchronize());
    ACX_SHARD_DISPATCH(&h->any, ACX_HIP_TRY(hipMemcpy(v, E.d.err, 8, hipMemcpyDeviceToHost)));
    *err = (int32_t)v[0];
    *min_len = (int32_t)v[1];
    return ACX_OK;
}

}  // extern "C"
