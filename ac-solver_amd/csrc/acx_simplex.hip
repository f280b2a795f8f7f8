// acx_simplex.hip -- the graph of presentations of total length <= n around <a, b>, as vertices and edges with their
// filtration values: the workload of the reference's C++ programs barcode_analysis/simplex_data_generation/
// {prime,classic}_moves/ac_bfs.cpp:12-98 (moves: AC_UTILS_as_sets.h:298-364), SURVEY section 8(f)-3.
//
// Breadth-first search over SORTED pairs of freely reduced relators (no cyclic reduction); a child of total length
// > n is ignored; vertices are named in the order the reference meets them (FIFO parent, move 0..M-1): the batch
// machinery of the BFS frontier (acx_frontier.h: id table with minimum-tag resolution, scan in tag order) gives exactly
// that numbering.  For every (vertex, move) whose child is kept and has a LARGER name the edge (vertex, child) is
// emitted, in (vertex, move) order, with filtration max(size(vertex), size(child)) -- repeated edges included, as the
// reference writes them.  Relators stay below 41 letters there (its Hash asserts it), so a key is two u128 words.
#include "acx_frontier.h"

namespace acx {
namespace simplex {

typedef u128 W;
constexpr int kLmax = 61;  // letters a key word can hold next to its 6-bit length

__device__ __forceinline__ bool rel_less(W a, int na, W b, int nb) {  // operator< on relators (AC_UTILS_as_sets.h:17-36)
    if (na != nb) return na < nb;
    const W t = a ^ b;
    if (!t) return false;
    const int p = wtraits<W>::ctz(t) >> 1;
    return get<W, true>(a, p) < get<W, true>(b, p);
}

__device__ __forceinline__ void sort_pair(W x, int nx, W y, int ny, Pres<W>& s) {  // Presentation(rel1, rel2), :282-290
    const bool keep = rel_less(x, nx, y, ny);
    s.w0 = keep ? x : y;
    s.n0 = keep ? nx : ny;
    s.w1 = keep ? y : x;
    s.n1 = keep ? ny : nx;
}

// Presentation::move (:298-364); letter codes a=2 b=3 A=1 B=0; conj0_(rel, x) = x^-1 rel x = conjugate_word with g = x^-1.
// false when a relator does not fit a key word (such a child is far above any length cap this program is used with).
__device__ __forceinline__ bool apply(const Pres<W>& p, int t, bool classic, Pres<W>& out) {
    const W r1 = p.w0, r2 = p.w1;
    const int n1 = p.n0, n2 = p.n1;
    W x = 0;
    int nx = 0;
    bool ok = true;
    if (classic) {
        const int cx[8] = {2, 3, 1, 0, 2, 3, 1, 0};  // conjugators of moves 4..11: a b A B (on rel2), a b A B (on rel1)
        if (t == 0) { ok = concat_words<W, true>(r1, n1, r2, n2, false, kLmax, x, nx); sort_pair(x, nx, r2, n2, out); }
        else if (t == 1) { ok = concat_words<W, true>(r2, n2, r1, n1, false, kLmax, x, nx); sort_pair(x, nx, r2, n2, out); }
        else if (t == 2) { ok = concat_words<W, true>(r1, n1, r2, n2, false, kLmax, x, nx); sort_pair(r1, n1, x, nx, out); }
        else if (t == 3) { ok = concat_words<W, true>(r2, n2, r1, n1, false, kLmax, x, nx); sort_pair(r1, n1, x, nx, out); }
        else if (t < 8) { ok = conjugate_word<W, true>(r2, n2, cx[t - 4] ^ 3, kLmax, x, nx); sort_pair(r1, n1, x, nx, out); }
        else if (t < 12) { ok = conjugate_word<W, true>(r1, n1, cx[t - 4] ^ 3, kLmax, x, nx); sort_pair(x, nx, r2, n2, out); }
        else if (t == 12) { sort_pair(inv<W, true>(r1, n1), n1, r2, n2, out); }
        else { sort_pair(r1, n1, inv<W, true>(r2, n2), n2, out); }
    } else {
        const int px[8] = {0, 1, 2, 3, 0, 1, 2, 3};  // conjugators of moves 4..11: B A a b (on rel1), B A a b (on rel2)
        if (t == 0) { ok = concat_words<W, true>(r1, n1, r2, n2, false, kLmax, x, nx); sort_pair(x, nx, r2, n2, out); }
        else if (t == 1) { ok = concat_words<W, true>(r2, n2, r1, n1, false, kLmax, x, nx); sort_pair(r1, n1, x, nx, out); }
        else if (t == 2) { ok = concat_words<W, true>(r1, n1, r2, n2, true, kLmax, x, nx); sort_pair(x, nx, r2, n2, out); }
        else if (t == 3) { ok = concat_words<W, true>(r2, n2, r1, n1, true, kLmax, x, nx); sort_pair(r1, n1, x, nx, out); }
        else if (t < 8) { ok = conjugate_word<W, true>(r1, n1, px[t - 4] ^ 3, kLmax, x, nx); sort_pair(x, nx, r2, n2, out); }
        else { ok = conjugate_word<W, true>(r2, n2, px[t - 4] ^ 3, kLmax, x, nx); sort_pair(r1, n1, x, nx, out); }
    }
    return ok;
}

// candidate t = M * parent + move; children above the length cap are flagged `known` (k_insert skips them)
__global__ void __launch_bounds__(256) k_expand_pairs(SearchDev<W> d, uint32_t pbegin, uint32_t np, int M, int ncap, int classic) {
    ACX_VGPR_PAD("v39");
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (uint32_t)M * np) return;
    const uint32_t p = t / (uint32_t)M, k = t - p * (uint32_t)M;
    Pres<W> s, c;
    key_to_pres<W>(d.k0[pbegin + p], d.k1[pbegin + p], s);
    const bool fits = apply(s, (int)k, classic != 0, c);
    if (fits && (c.n0 == 0 || c.n1 == 0)) atomicOr(d.err, 1u);  // the reference's Hash cannot hold an empty relator (it would crash)
    const int size = c.n0 + c.n1;
    const bool keep = fits && size <= ncap && c.n0 > 0 && c.n1 > 0;
    d.ck0[t] = keep ? keyops<W>::make(c.w0, c.n0) : (W)0;
    d.ck1[t] = keep ? keyops<W>::make(c.w1, c.n1) : (W)0;
    d.clen[t] = keep ? (uint8_t)size : (uint8_t)0xff;
    d.cknown[t] = keep ? 0 : 1;
}

__global__ void __launch_bounds__(256) k_commit_pairs(SearchDev<W> d, uint32_t m, uint32_t base, uint32_t cap_nodes) {
    ACX_VGPR_PAD("v23");
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= m || !d.cflag[t]) return;
    const uint32_t id = base + d.cpos[t];
    if (id < cap_nodes) {
        d.k0[id] = d.ck0[t];
        d.k1[id] = d.ck1[t];
        d.tlen[id] = d.clen[t];
    }
    d.slots[d.cslot[t]] = id;
}

// after the commit every kept candidate's table slot holds the final name of its state
__global__ void __launch_bounds__(256) k_edge_flags(SearchDev<W> d, uint32_t m, uint32_t pbegin, int M, uint32_t* __restrict__ eflag) {
    ACX_VGPR_PAD("v15");
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= m) return;
    eflag[t] = (!d.cknown[t] && pbegin + t / (uint32_t)M < d.slots[d.cslot[t]]) ? 1u : 0u;
}

__global__ void __launch_bounds__(256) k_edge_write(SearchDev<W> d, uint32_t m, uint32_t pbegin, int M, const uint32_t* __restrict__ eflag,
                                                     const uint32_t* __restrict__ epos, unsigned long long ebase, unsigned long long cap_edges,
                                                     uint32_t* __restrict__ edges, uint8_t* __restrict__ efilt) {
    ACX_VGPR_PAD("v23");
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= m || !eflag[t]) return;
    const unsigned long long e = ebase + epos[t];
    if (e >= cap_edges) return;
    const uint32_t cn = pbegin + t / (uint32_t)M;
    edges[2 * e] = cn;
    edges[2 * e + 1] = d.slots[d.cslot[t]];
    const uint8_t a = d.tlen[cn], b = d.clen[t];
    efilt[e] = a > b ? a : b;
}

__global__ void k_root_pairs(SearchDev<W> d) {
    ACX_VGPR_PAD("v23");  // <a, b>: relators [a] and [b], vertex 0
    const W k0 = keyops<W>::make((W)2, 1), k1 = keyops<W>::make((W)3, 1);
    d.k0[0] = k0;
    d.k1[0] = k1;
    d.tlen[0] = 2;
    d.slots[(uint32_t)hash_key<W>(k0, k1) & d.smask] = 0;
}

}  // namespace simplex
}  // namespace acx

using namespace acx;
using namespace acx::simplex;

extern "C" int acx_simplex_graph(int n, int classic, int64_t cap_nodes, int64_t cap_edges, int64_t* n_nodes, uint8_t* h_node_size, int64_t* n_edges,
                                 uint32_t* h_edges, uint8_t* h_edge_filt) {
    if (!have_device()) return ACX_E_NODEVICE;
    if (n < 2 || n > 41 || cap_nodes < 1 || cap_edges < 0 || !n_nodes || !n_edges || !h_node_size || (cap_edges > 0 && (!h_edges || !h_edge_filt)))
        return fail(ACX_E_INVAL, "acx_simplex_graph: bad argument (2 <= n <= 41)");
    if (cap_nodes > (1ll << 30)) return fail(ACX_E_INVAL, "acx_simplex_graph: cap_nodes above 2^30");
    const int M = classic ? 14 : 12;
    const uint32_t bmax = (uint32_t)std::min<int64_t>(std::max<int64_t>(cap_nodes / 4, 1024), 1 << 18);  // parents per batch
    const uint64_t cap_cand = (uint64_t)bmax * M;
    uint64_t n_slots = 1024;
    while (n_slots < 2 * ((uint64_t)cap_nodes + cap_cand)) n_slots <<= 1;
    SearchDev<W> d;
    memset((void*)&d, 0, sizeof(d));
    DevBuf b_nodes, b_cand, b_tab, b_scal, b_tmp, b_edges;
    size_t o = 0;
    auto take = [&](uint8_t* base, size_t bytes) {
        uint8_t* p = base ? base + o : nullptr;
        o += (bytes + 255) / 256 * 256;
        return p;
    };
    for (int pass = 0; pass < 2; pass++) {
        uint8_t* b = (uint8_t*)b_nodes.p;
        o = 0;
        d.k0 = (W*)take(b, (size_t)cap_nodes * sizeof(W));
        d.k1 = (W*)take(b, (size_t)cap_nodes * sizeof(W));
        d.tlen = take(b, (size_t)cap_nodes);
        if (pass == 0 && b_nodes.alloc(o)) return ACX_E_NOMEM;
    }
    uint32_t *eflag = nullptr, *epos = nullptr;
    for (int pass = 0; pass < 2; pass++) {
        uint8_t* b = (uint8_t*)b_cand.p;
        o = 0;
        d.ck0 = (W*)take(b, cap_cand * sizeof(W));
        d.ck1 = (W*)take(b, cap_cand * sizeof(W));
        d.cslot = (uint32_t*)take(b, cap_cand * 4);
        d.cflag = (uint32_t*)take(b, cap_cand * 4);
        d.cpos = (uint32_t*)take(b, cap_cand * 4);
        eflag = (uint32_t*)take(b, cap_cand * 4);
        epos = (uint32_t*)take(b, cap_cand * 4);
        d.clen = take(b, cap_cand);
        d.cknown = take(b, cap_cand);
        if (pass == 0 && b_cand.alloc(o)) return ACX_E_NOMEM;
    }
    if (b_tab.alloc(n_slots * 4) || b_scal.alloc(256)) return ACX_E_NOMEM;
    d.slots = (uint32_t*)b_tab.p;
    d.smask = (uint32_t)(n_slots - 1);
    d.err = (uint32_t*)b_scal.p;
    const size_t edge_bytes = (size_t)std::max<int64_t>(cap_edges, 1) * 9;
    if (b_edges.alloc(edge_bytes)) return ACX_E_NOMEM;
    uint32_t* d_edges = (uint32_t*)b_edges.p;
    uint8_t* d_efilt = (uint8_t*)b_edges.p + (size_t)std::max<int64_t>(cap_edges, 1) * 8;
    size_t tmp_bytes = 0;
    if (int src = scan_u32_exclusive(nullptr, &tmp_bytes, d.cflag, d.cpos, cap_cand, (hipStream_t) nullptr)) return src;
    if (b_tmp.alloc(tmp_bytes + 256)) return ACX_E_NOMEM;
    hipStream_t st = nullptr;
    ACX_HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    struct Guard {
        hipStream_t s;
        ~Guard() { (void)hipStreamDestroy(s); }
    } guard{st};
    ACX_HIP_TRY(hipMemsetAsync(d.slots, 0xff, n_slots * 4, st));
    ACX_HIP_TRY(hipMemsetAsync(b_scal.p, 0, 256, st));
    hipLaunchKernelGGL(k_root_pairs, dim3(1), dim3(1), 0, st, d);
    uint64_t nodes = 1, edges = 0, head = 0;
    int rc = ACX_OK;
    while (head < nodes && rc == ACX_OK) {
        const uint32_t np = (uint32_t)std::min<uint64_t>(nodes - head, bmax);
        const uint32_t m = np * (uint32_t)M;
        const dim3 grid((m + 255) / 256), block(256);
        hipLaunchKernelGGL(k_expand_pairs, grid, block, 0, st, d, (uint32_t)head, np, M, n, classic ? 1 : 0);
        hipLaunchKernelGGL(k_insert<W>, grid, block, 0, st, d, d.slots, d.smask, m, 1);
        hipLaunchKernelGGL(k_mark<W>, grid, block, 0, st, d, d.slots, m, -1);
        size_t tb = tmp_bytes;
        if (int src = scan_u32_exclusive(b_tmp.p, &tb, d.cflag, d.cpos, (size_t)m, st)) return src;
        hipLaunchKernelGGL(k_commit_pairs, grid, block, 0, st, d, m, (uint32_t)nodes, (uint32_t)cap_nodes);
        hipLaunchKernelGGL(k_edge_flags, grid, block, 0, st, d, m, (uint32_t)head, M, eflag);
        tb = tmp_bytes;
        if (int src = scan_u32_exclusive(b_tmp.p, &tb, eflag, epos, (size_t)m, st)) return src;
        hipLaunchKernelGGL(k_edge_write, grid, block, 0, st, d, m, (uint32_t)head, M, eflag, epos, (unsigned long long)edges, (unsigned long long)cap_edges, d_edges,
                           d_efilt);
        ACX_HIP_TRY(hipGetLastError());
        uint32_t last[4], err = 0;
        ACX_HIP_TRY(hipMemcpyAsync(&last[0], d.cpos + (m - 1), 4, hipMemcpyDeviceToHost, st));
        ACX_HIP_TRY(hipMemcpyAsync(&last[1], d.cflag + (m - 1), 4, hipMemcpyDeviceToHost, st));
        ACX_HIP_TRY(hipMemcpyAsync(&last[2], epos + (m - 1), 4, hipMemcpyDeviceToHost, st));
        ACX_HIP_TRY(hipMemcpyAsync(&last[3], eflag + (m - 1), 4, hipMemcpyDeviceToHost, st));
        ACX_HIP_TRY(hipMemcpyAsync(&err, d.err, 4, hipMemcpyDeviceToHost, st));
        ACX_HIP_TRY(hipStreamSynchronize(st));
        if (err) return fail(ACX_E_ROWERR, "acx_simplex_graph: a relator became empty (the reference's Hash cannot represent that)");
        nodes += (uint64_t)last[0] + last[1];
        edges += (uint64_t)last[2] + last[3];
        head += np;
        if (nodes > (uint64_t)cap_nodes || edges > (uint64_t)cap_edges) rc = ACX_E_CAPACITY;
    }
    *n_nodes = (int64_t)nodes;
    *n_edges = (int64_t)edges;
    if (rc == ACX_E_CAPACITY)
        return fail(ACX_E_CAPACITY, "acx_simplex_graph: the graph needs more than cap_nodes = %lld vertices or cap_edges = %lld edges", (long long)cap_nodes,
                    (long long)cap_edges);
    ACX_HIP_TRY(hipMemcpyAsync(h_node_size, d.tlen, (size_t)nodes, hipMemcpyDeviceToHost, st));
    if (edges) {
        ACX_HIP_TRY(hipMemcpyAsync(h_edges, d_edges, (size_t)edges * 8, hipMemcpyDeviceToHost, st));
        ACX_HIP_TRY(hipMemcpyAsync(h_edge_filt, d_efilt, (size_t)edges, hipMemcpyDeviceToHost, st));
    }
    ACX_HIP_TRY(hipStreamSynchronize(st));
    return ACX_OK;
}
