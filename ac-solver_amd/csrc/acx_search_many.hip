// acx_search_many.hip -- many independent searches per call (acx_search_many, acx_search_groups): what the reference's Miller-Schupp
// driver does one search after the other (search/miller_schupp/miller_schupp.py:140-158).
//   greedy_search  the searches are JOBS that a fixed set of persistent workgroups takes from a counter (acx_greedy.h: k_greedy_sched),
//                  one launch per key width
//   bfs            the searches of a batch share the launches of the fused single search, a tile of every search's frontier per
//                  workgroup (acx_bfs_many.h)
// A single search (n == 1), verbose searches and searches under the digest hook go through acx_search, one after the other on a few
// host threads.
#include "acx_searcher.h"
#include "acx_bfs.h"
#include "acx_bfs_many.h"
#include "acx_greedy.h"

namespace acx {

// Greedy searches as JOBS on a fixed set of workgroup slots (acx_greedy.h: k_greedy_sched).  `groups`: batches of presentations, each
// with its own max_relator_length (all of one key width W); out0 = index of a batch's first search in the output arrays.
struct SearchGroupIn {
    const int8_t* rows;
    int64_t n;
    int L;
    int64_t out0;
};
// Pinned host memory for the results of a launch, from a pool (never freed: no HIP calls at process exit).  The launches of a greedy sweep
// read 40 MB of path buffers back.  Into pageable vectors the runtime pins those pages for the copy and unpins them afterwards, and the
// device's NEXT operation -- the first fill of the next call, on any stream -- then completed 10-25 ms after it was enqueued (round 6:
// from the third sweep of a process on, rocprofv3 showing the device idle until a point on a 10 ms grid; the sweep's later repetitions
// took 0.15-0.16 s against the second's 0.135).
struct HostBuf {
    void* p = nullptr;
    size_t bytes = 0;
    struct Pooled {
        void* p;
        size_t bytes;
    };
    static std::mutex& mu() {
        static std::mutex m;
        return m;
    }
    static std::vector<Pooled>& pool() {
        static std::vector<Pooled>* v = new std::vector<Pooled>();
        return *v;
    }
    int alloc(size_t want) {
        want = want ? want : 1;
        {
            std::lock_guard<std::mutex> lock(mu());
            auto& v = pool();
            size_t best = v.size();
            for (size_t i = 0; i < v.size(); i++)
                if (v[i].bytes >= want && v[i].bytes <= 2 * want + 4096 && (best == v.size() || v[i].bytes < v[best].bytes)) best = i;
            if (best != v.size()) {
                p = v[best].p, bytes = v[best].bytes;
                v[best] = v.back();
                v.pop_back();
                return ACX_OK;
            }
        }
        if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) {
            p = nullptr;
            return fail(ACX_E_NOMEM, "hipHostMalloc(%zu) failed", want);
        }
        bytes = want;
        return ACX_OK;
    }
    HostBuf() = default;
    HostBuf(const HostBuf&) = delete;
    HostBuf& operator=(const HostBuf&) = delete;
    ~HostBuf() {
        if (!p) return;
        {
            std::lock_guard<std::mutex> lock(mu());
            size_t held = 0;
            for (const Pooled& q : pool()) held += q.bytes;
            if (held + bytes <= (512ull << 20)) {  // (a sweep's buffers are 40 MB; beyond half a gigabyte of idle pinned memory a buffer goes back to the driver)
                pool().push_back({p, bytes});
                return;
            }
        }
        (void)hipHostFree(p);
    }
};

void host_buffers_trim() {  // acx_release_cached_memory: the idle pinned buffers go back to the driver too
    std::vector<HostBuf::Pooled> old;
    {
        std::lock_guard<std::mutex> lock(HostBuf::mu());
        old.swap(HostBuf::pool());
    }
    for (const HostBuf::Pooled& q : old) (void)hipHostFree(q.p);
}

// The streams of a call come from a pool and go back to it (synchronised), as the stream pairs of acx_search do: a call used to create
// and destroy three -- one for the set-up, a high- and a low-priority one for the two launches.  Classes: 0 default priority, 1 highest,
// 2 lowest.
struct StreamLease {
    hipStream_t s = nullptr;
    int cls = 0, dev = 0;
    struct Pooled {
        hipStream_t s;
        int cls, dev;
    };
    static std::mutex& mu() {
        static std::mutex m;
        return m;
    }
    static std::vector<Pooled>& pool() {
        static std::vector<Pooled> p;
        return p;
    }
    int take(int cls_) {
        cls = cls_;
        (void)hipGetDevice(&dev);
        {
            std::lock_guard<std::mutex> lock(mu());
            auto& p = pool();
            for (size_t i = 0; i < p.size(); i++)
                if (p[i].cls == cls && p[i].dev == dev) {
                    s = p[i].s;
                    p[i] = p.back();
                    p.pop_back();
                    return ACX_OK;
                }
        }
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        if (cls == 0) ACX_HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        else ACX_HIP_TRY(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, cls == 1 ? hi : lo));
        return ACX_OK;
    }
    StreamLease() = default;
    StreamLease(const StreamLease&) = delete;
    StreamLease& operator=(const StreamLease&) = delete;
    ~StreamLease() {
        if (!s) return;
        (void)hipStreamSynchronize(s);
        std::lock_guard<std::mutex> lock(mu());
        pool().push_back({s, cls, dev});
    }
};

// workgroups of k_greedy_sched the device holds at once: kGreedyPerCu per compute unit (512 on an MI355X: 256 compute units)
static uint32_t greedy_resident() {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    return kGreedyPerCu * (uint32_t)std::max(cus, 1);
}
static uint32_t greedy_slots_wanted() {
    // a slot per resident workgroup; ACX_OPT_GREEDY_SLOTS: the tests run many jobs on a few slots
    return (uint32_t)std::min<int64_t>(std::max<int64_t>(option(ACX_OPT_GREEDY_SLOTS, greedy_resident()), 1), 4096);
}

// The SLOTS of a call -- the memory of one greedy search each (visited table, bucket table / bitmap / arena, node arrays, a small sort
// scratch) -- in ONE layout for every key width of the call (the arrays of 64-bit keys use the front of what 128-bit keys would), a
// free bitmap on the device (k_greedy_sched: a workgroup takes a slot with its first job and hands it back clean), and a few
// full-size sort scratch regions shared by all slots (acx_greedy.h: GreedyDev::big_lock).  Blocks of <= 6 GB: the block pool keeps them.
struct GreedySlots {
    uint32_t S = 0, per_group = 0, nlen_max = 0, big_n = 0, words = 0;
    uint64_t n_tab = 0, cap_nodes = 0, arena_entries = 0, scratch_cap = 0, sort_cap = 0, key_bytes = 0;
    uint64_t b_tab = 0, b_bk = 0, b_bm = 0, b_arena = 0, b_key = 0, b_u32 = 0, b_u8 = 0, b_gk = 0, b_gid = 0, per_rest = 0, per_slot = 0;
    uint64_t big_key_bytes = 0, big_stride = 0;
    std::vector<DevBuf> groups;
    std::vector<uint8_t> was_clean;  // per group: the block came from the pool with this layout's tag (no fills needed)
    DevBuf free_bits, big, big_lock;
    static uint64_t up(uint64_t b) { return (b + 255) / 256 * 256; }
    // names the layout of a group of m slots: a block that carries it holds m slots of exactly these arrays, all clean
    uint64_t layout_tag(uint64_t m) const {
        uint64_t h = 0x9E3779B97F4A7C15ull;
        for (uint64_t v : {m, b_tab, b_bk, b_bm, per_rest, (uint64_t)nlen_max}) h = (h ^ v) * 0xD6E8FEB86659FD93ull, h ^= h >> 32;
        return h | 1;
    }
    // every kernel of the call has ended well: every workgroup has handed its slot back clean, the blocks may say so in the pool
    void mark_clean() {
        for (uint32_t q = 0; q < groups.size(); q++) groups[q].clean_tag = layout_tag(std::min<uint32_t>(per_group, S - q * per_group));
    }

    // `n_jobs` searches of at most `max_nodes` nodes, max_relator_length <= L_max, `wide`: some of them with 128-bit keys.  Everything is
    // set up on `st` (tables free, bucket tables zero, every slot free); the caller synchronises `st` before it launches.
    int setup(int64_t n_jobs, int64_t max_nodes, int L_max, bool wide, uint32_t slots_wanted, hipStream_t st) {
        key_bytes = wide ? 16 : 8;
        cap_nodes = (uint64_t)max_nodes + 64 + 12 * 1024;
        n_tab = 1024;
        while (n_tab < 2 * (cap_nodes + 12288)) n_tab <<= 1;
        if (n_tab > (1ull << 31)) return fail(ACX_E_INVAL, "acx_search_many: budget too large for 32-bit node ids");
        nlen_max = (uint32_t)(2 * L_max + 1);
        arena_entries = std::min<uint64_t>(8ull * (uint64_t)std::max<int64_t>(max_nodes, 1) + (1ull << 20), 1ull << 31);
        sort_cap = 2 * ((uint64_t)std::max<int64_t>(max_nodes, 1) + 64) + 4096;
        scratch_cap = std::min<uint64_t>(sort_cap, (uint64_t)std::max<int64_t>(option(ACX_OPT_GREEDY_SCRATCH, 1 << 16), 2048));
        b_tab = up(n_tab * 8), b_bk = up((uint64_t)nlen_max * kDepthCap * sizeof(BucketRec)), b_bm = up((uint64_t)nlen_max * (kDepthCap / 32) * 4);
        b_arena = up(arena_entries * 4), b_key = up(cap_nodes * 2 * key_bytes), b_u32 = up(cap_nodes * 4), b_u8 = up(cap_nodes);
        b_gk = up(scratch_cap * key_bytes), b_gid = up(scratch_cap * 4);
        per_rest = b_arena + b_key + 2 * b_u32 + 2 * b_u8 + 2 * b_gk + b_gid;
        per_slot = b_tab + b_bk + b_bm + per_rest;
        size_t free_b = 0, total_b = 0;
        double avail = 64e9;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) avail = (double)free_b + (double)block_pool().cached_on(BlockPool::current_device());
        const double budget = std::max(2.0 * (double)per_slot, std::min(112e9, avail * 0.4));
        S = (uint32_t)std::max<double>(1.0, std::min<double>(std::min<double>((double)std::max<int64_t>(n_jobs, 1), (double)std::max<uint32_t>(slots_wanted, 1u)), budget / (double)per_slot));
        for (;;) {
            per_group = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(S, (6ull << 30) / per_slot));
            const uint32_t n_groups = (S + per_group - 1) / per_group;
            groups.clear();
            groups.resize(n_groups);
            bool ok = true;
            was_clean.assign(n_groups, 0);
            for (uint32_t q = 0; q < n_groups && ok; q++) {
                uint64_t tag = layout_tag(std::min<uint32_t>(per_group, S - q * per_group));
                ok = groups[q].alloc((uint64_t)std::min<uint32_t>(per_group, S - q * per_group) * per_slot, &tag) == ACX_OK;
                was_clean[q] = tag != 0;
            }
            if (ok) break;
            (void)hipGetLastError();  // (the estimate of the free memory was too good: fewer slots -- the jobs just take longer)
            groups.clear();
            if (S == 1) return ACX_E_NOMEM;
            S = (S + 1) / 2;
        }
        for (uint32_t q = 0; q * per_group < S; q++) {
            const uint64_t m = std::min<uint32_t>(per_group, S - q * per_group);
            uint8_t* base = (uint8_t*)groups[q].p;
            if (was_clean[q]) continue;  // the block comes back from the pool as the last call's workgroups left it: every slot clean
            ACX_HIP_TRY(hipMemsetAsync(base, 0xff, m * b_tab, st));
            ACX_HIP_TRY(hipMemsetAsync(base + m * b_tab, 0, m * (b_bk + b_bm), st));
        }
        // every slot free: all-ones words, the last one cut to the slots that exist (device fills: no host buffer, no synchronisation here)
        words = (S + 31) / 32;
        if (free_bits.alloc((size_t)words * 4)) return ACX_E_NOMEM;
        ACX_HIP_TRY(hipMemsetAsync(free_bits.p, 0xff, (size_t)words * 4, st));
        if (S & 31u) ACX_HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)((uint32_t*)free_bits.p + (words - 1)), (int)((1u << (S & 31u)) - 1u), 1, st));
        big_n = 0;
        if (sort_cap > scratch_cap) {  // buckets of more than half the slot's scratch: a few full-size regions for all slots
            big_key_bytes = up(sort_cap * key_bytes);
            big_stride = 2 * big_key_bytes + up(sort_cap * 4);
            big_n = std::min<uint32_t>(std::max<uint32_t>(S / 16, 2), 16);
            while (big_n > 1 && big.alloc((uint64_t)big_n * big_stride)) {
                (void)hipGetLastError();
                big_n /= 2;
            }
            if (!big.p && big.alloc((uint64_t)big_n * big_stride)) return ACX_E_NOMEM;
            if (big_lock.alloc(256)) return ACX_E_NOMEM;
            ACX_HIP_TRY(hipMemsetAsync(big_lock.p, 0, 256, st));
        }
        return ACX_OK;
    }
    template <typename W> void describe(std::vector<GreedyDev<W>>& out, int cyclical, int64_t max_nodes) const {
        out.resize(S);
        for (uint32_t r = 0; r < S; r++) {
            GreedyDev<W>& g = out[r];
            memset(&g, 0, sizeof(g));
            const uint32_t q = r / per_group, k = r % per_group;
            const uint64_t m = std::min<uint32_t>(per_group, S - q * per_group);
            uint8_t* base = (uint8_t*)groups[q].p;
            g.tab = (unsigned long long*)(base + k * b_tab);
            g.tmask = (uint32_t)(n_tab - 1);
            g.bk = (BucketRec*)(base + m * b_tab + k * b_bk);
            g.bitmap = (uint32_t*)(base + m * (b_tab + b_bk) + k * b_bm);
            uint8_t* x = base + m * (b_tab + b_bk + b_bm) + k * per_rest;
            auto take = [&](uint64_t bytes) {
                uint8_t* y = x;
                x += bytes;
                return y;
            };
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
            g.scratch_cap = (uint32_t)scratch_cap;
            if (big_n) {
                g.big_n = big_n;
                g.big_lock = (uint32_t*)big_lock.p;
                g.big_base = (uint8_t*)big.p;
                g.big_key_bytes = big_key_bytes;
                g.big_stride = big_stride;
            }
        }
    }
};

// The jobs of one key width on `n_wgs` workgroups over the call's slots.
template <typename W>
static int run_greedy_sched(const GreedySlots& pool, const std::vector<SearchGroupIn>& groups, int64_t max_nodes, int cyclical, int32_t* solved, int32_t* path_action,
                            int32_t* path_len, int64_t path_cap, int64_t* path_n, acx_search_stats* stats, int32_t* rc_out, uint8_t* need_rerun, uint32_t n_wgs,
                            std::atomic<int>* launched = nullptr, std::atomic<int>* wait_for = nullptr) {
    // `launched` is set once this width's first kernel is in its queue (or the function returns), `wait_for`: that flag of the other width
    struct Signal {
        std::atomic<int>* f;
        ~Signal() {
            if (f) f->store(1);
        }
    } signal{launched};
    int64_t n_all = 0;
    for (const auto& gr : groups) n_all += gr.n;
    if (n_all == 0) return ACX_OK;
    const auto t_begin = std::chrono::steady_clock::now();
    auto since = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count(); };
    // the jobs, by move code: a root in normal form keeps its search in normal form (the shorter move code); one launch per code
    std::vector<GreedyJob<W>> jobs[2];
    std::vector<int64_t> where[2];  // job -> index in the output arrays
    for (const auto& gr : groups)
        for (int64_t k = 0; k < gr.n; k++) need_rerun[gr.out0 + k] = 0, rc_out[gr.out0 + k] = ACX_OK, solved[gr.out0 + k] = 0, path_n[gr.out0 + k] = 0;
    for (const auto& gr : groups)
        for (int64_t k = 0; k < gr.n; k++) {
            const int64_t o = gr.out0 + k;
            Pres<W> root;
            bool ok = pack_relator<W>(gr.rows + k * 2 * gr.L, gr.L, root.w0, root.n0);
            ok = pack_relator<W>(gr.rows + k * 2 * gr.L + gr.L, gr.L, root.w1, root.n1) && ok;
            if (!ok) {
                rc_out[o] = ACX_E_ROWERR;
                return fail(ACX_E_ROWERR, "acx_search_many: presentation %lld is not a zero-padded word pair over {+-1,+-2}", (long long)o);
            }
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
    // which searches stay unsolved is not known beforehand, so: the smaller max_relator_length first, the longer relators first.
    // Round 5, on the slot pool: this order 0.147-0.153 s; longest roots first whatever the bound 0.168; the wider bounds first 0.167;
    // by 2 L - root length 0.145-0.150; the 128-bit launch held to 300 / 256 workgroups 0.148 / 0.153-0.163)
    for (int code = 0; code < 2 && !option(ACX_OPT_GREEDY_KEEP_ORDER, 0); code++) {
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
    const uint32_t R = pool.S;
    std::vector<GreedyDev<W>> hslots;
    pool.describe<W>(hslots, cyclical, max_nodes);
    DevBuf dslots, djobs, dcounter, douts, dpa, dpl;
    // the 128-bit searches are the longer ones: when both widths are in flight their workgroups get a free compute unit first
    StreamLease lease;  // (declared behind the device buffers: synchronised before they go back to the pool)
    if (lease.take(sizeof(W) > 8 ? 1 : 2) != ACX_OK) return ACX_E_NODEVICE;
    hipStream_t st = lease.s;
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
        // (the slots are as a search expects them -- table free, bucket records and bitmaps zero: GreedySlots::setup, and every workgroup
        // hands its slot back that way)
        ACX_HIP_TRY(hipMemcpyAsync(dslots.p, hslots.data(), (size_t)R * sizeof(GreedyDev<W>), hipMemcpyHostToDevice, st));
        ACX_HIP_TRY(hipMemcpyAsync(djobs.p, jobs[code].data(), nj * sizeof(GreedyJob<W>), hipMemcpyHostToDevice, st));
        ACX_HIP_TRY(hipMemsetAsync(dcounter.p, 0, 256, st));
        ACX_HIP_TRY(hipMemsetAsync(douts.p, 0, nj * sizeof(GreedyOut), st));
        ACX_HIP_TRY(hipEventRecord(evs.a, st));
        const unsigned grid = (unsigned)std::min<size_t>(std::max<uint32_t>(n_wgs, 1u), nj);
        if (wait_for) {
            while (!wait_for->load()) std::this_thread::yield();
            wait_for = nullptr;
        }
        if (code)
            hipLaunchKernelGGL((k_greedy_sched<W, true>), dim3(grid), dim3(kGreedyMultiThreads), 0, st, (const GreedyDev<W>*)dslots.p, (uint32_t*)pool.free_bits.p, pool.words,
                               (const GreedyJob<W>*)djobs.p, (uint32_t)nj, (uint32_t*)dcounter.p, (GreedyOut*)douts.p, (int32_t*)dpa.p, (int32_t*)dpl.p, (long long)pc);
        else
            hipLaunchKernelGGL((k_greedy_sched<W, false>), dim3(grid), dim3(kGreedyMultiThreads), 0, st, (const GreedyDev<W>*)dslots.p, (uint32_t*)pool.free_bits.p, pool.words,
                               (const GreedyJob<W>*)djobs.p, (uint32_t)nj, (uint32_t*)dcounter.p, (GreedyOut*)douts.p, (int32_t*)dpa.p, (int32_t*)dpl.p, (long long)pc);
        ACX_HIP_TRY(hipGetLastError());
        if (launched) launched->store(1);
        ACX_HIP_TRY(hipEventRecord(evs.b, st));
        const double t_launched = since();
        HostBuf ho, hpa, hpl;
        if (ho.alloc(nj * sizeof(GreedyOut)) || hpa.alloc(nj * (size_t)pc * 4) || hpl.alloc(nj * (size_t)pc * 4)) return ACX_E_NOMEM;
        struct View {  // (what the code below indexes)
            int32_t* q;
            int32_t* data() const { return q; }
        };
        const GreedyOut* o = (const GreedyOut*)ho.p;
        const View pa{(int32_t*)hpa.p}, pl{(int32_t*)hpl.p};
        ACX_HIP_TRY(hipMemcpyAsync(ho.p, douts.p, nj * sizeof(GreedyOut), hipMemcpyDeviceToHost, st));
        ACX_HIP_TRY(hipMemcpyAsync(hpa.p, dpa.p, nj * (size_t)pc * 4, hipMemcpyDeviceToHost, st));
        ACX_HIP_TRY(hipMemcpyAsync(hpl.p, dpl.p, nj * (size_t)pc * 4, hipMemcpyDeviceToHost, st));
        ACX_HIP_TRY(hipStreamSynchronize(st));
        float ms = 0;
        ACX_HIP_TRY(hipEventElapsedTime(&ms, evs.a, evs.b));
        if (g_debug) {  // -DACX_GREEDY_PROFILE=1 builds: where the workgroups' cycles go, summed over the launch's searches
            unsigned long long tp[12] = {}, tot = 0, batches = 0, sorts = 0, bigs = 0;
            for (size_t j = 0; j < nj; j++) {
                for (int q = 0; q < 12; q++) tp[q] += o[j].t_phase[q];
                batches += o[j].batches;
                sorts += o[j].sorts;
                bigs += o[j].big_sorts;
            }
            for (int q = 0; q < 8; q++) tot += tp[q];
            {
                unsigned long long hs[16] = {}, top = 0;
                for (size_t j = 0; j < nj; j++) {
                    for (int q = 0; q < 16; q++) hs[q] += o[j].hist_sort[q];
                    top = std::max<unsigned long long>(top, o[j].arena_top);
                }
                fprintf(stderr, "[acx_greedy_sched] sorts by log2(n):");
                for (int q = 0; q < 16; q++) fprintf(stderr, " %llu", hs[q]);
                fprintf(stderr, "; a slot's own scratch holds %llu entries, %u shared regions; largest bucket arena in use %llu of %llu entries\n", (unsigned long long)pool.scratch_cap, pool.big_n, top,
                        (unsigned long long)pool.arena_entries);
            }
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
            {  // the launch's timeline from the jobs' own stamps (100 MHz counter): how long the jobs are, when the long ones started, how busy the workgroups were
                unsigned long long t_first = ~0ull, t_last = 0;
                for (size_t j = 0; j < nj; j++) t_first = std::min(t_first, o[j].t_phase[22]), t_last = std::max(t_last, o[j].t_phase[23]);
                std::vector<size_t> idx(nj);
                for (size_t j = 0; j < nj; j++) idx[j] = j;
                std::sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return o[a].t_phase[23] - o[a].t_phase[22] > o[b].t_phase[23] - o[b].t_phase[22]; });
                double busy = 0;
                for (size_t j = 0; j < nj; j++) busy += (double)(o[j].t_phase[23] - o[j].t_phase[22]) * 1e-5;
                fprintf(stderr, "[acx_greedy_sched] timeline: first job starts at raw %.2f ms, last job ends %.2f ms later; jobs' own time %.1f ms in all = %.2f ms on each of %u workgroups\n",
                        (double)(t_first % 100000000ull) * 1e-5, (double)(t_last - t_first) * 1e-5, busy, busy / grid, grid);
                for (size_t q = 0; q < std::min<size_t>(nj, 10); q++) {
                    const GreedyOut& x = o[idx[q]];
                    fprintf(stderr, "[acx_greedy_sched]   job %zu (L %d, root length %u, %s, %llu batches): %.2f ms, started at %.2f, workgroup %llu\n", idx[q], jobs[code][idx[q]].L, jobs[code][idx[q]].root_len,
                            x.status == GREEDY_SOLVED ? "solved" : "unsolved", (unsigned long long)x.batches, (double)(x.t_phase[23] - x.t_phase[22]) * 1e-5, (double)(x.t_phase[22] - t_first) * 1e-5, x.t_phase[21]);
                }
                size_t late = 0;  // jobs of at least half the longest that started in the launch's second half
                const unsigned long long longest = o[idx[0]].t_phase[23] - o[idx[0]].t_phase[22];
                for (size_t j = 0; j < nj; j++)
                    if (2 * (o[j].t_phase[23] - o[j].t_phase[22]) >= longest && 2 * (o[j].t_phase[22] - t_first) >= t_last - t_first) late++;
                double dur[5];
                for (int q = 0; q < 5; q++) dur[q] = (double)(o[idx[std::min(nj - 1, nj * (size_t)q / 4)]].t_phase[23] - o[idx[std::min(nj - 1, nj * (size_t)q / 4)]].t_phase[22]) * 1e-5;
                fprintf(stderr, "[acx_greedy_sched]   job ms: max %.2f, upper quartile %.2f, median %.2f, lower quartile %.2f, min %.2f; %zu jobs of at least half the longest started in the second half\n", dur[0], dur[1],
                        dur[2], dur[3], dur[4], late);
            }
            fprintf(stderr, "[acx_greedy_sched] %zu searches on %u workgroups (%s move code), %llu batches, %llu sorts (%llu of buckets larger than the LDS), launch %.2f ms; host: launched at %.1f ms, results at %.1f ms\n", nj, grid,
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

// A group of independent breadth-first searches, level-synchronous on the kernels of the fused single search (acx_bfs_many.h): one
// round of launches advances every running search by one batch of at most `bmax` parents.  rc_out[k] = ACX_OK / ACX_E_CAPACITY (path
// buffer, or a probe sequence that ran through the whole table) / ACX_E_ROWERR (the reference raises).
static uint32_t bfs_many_bmax() {
    const int64_t v = option(ACX_OPT_BFS_MANY_BMAX, 1 << 15);  // (the tests shrink it: more rounds, every tile edge)
    return (uint32_t)std::min<int64_t>(std::max<int64_t>(v, 128), 1 << 22);
}
template <typename W>
static int run_bfs_group_fused(const int8_t* rows, int64_t n, int L, int64_t max_nodes, int cyclical, int32_t* solved, int32_t* path_action, int32_t* path_len,
                               int64_t path_cap, int64_t* path_n, acx_search_stats* stats, int32_t* rc_out) {
    if (n <= 0) return ACX_OK;
    const uint32_t bmax = (uint32_t)std::min<int64_t>(std::max<int64_t>(max_nodes / 4, 1024), bfs_many_bmax());
    const uint64_t cap_nodes = (uint64_t)std::max<int64_t>(max_nodes, 0) + 64, cap_cand = 12ull * bmax;
    uint64_t n_slots = 1024;
    while (n_slots < 2 * (cap_nodes + cap_cand)) n_slots <<= 1;  // two slots per stamp the table may ever hold
    if (n_slots > (1ull << 31)) return fail(ACX_E_INVAL, "acx_search_many: budget too large for 32-bit node ids");
    auto up = [](uint64_t b) { return (b + 255) / 256 * 256; };
    const uint64_t tiles = cap_cand / kCompactTile + 2;
    const uint64_t b_tab = up(n_slots * 8), b_key = up(cap_nodes * sizeof(W)), b_u32 = up(cap_nodes * 4), b_u8 = up(cap_nodes), b_flag = up(cap_cand + 8);
    const uint64_t b_counts = up(tiles * 4), b_masks = up(tiles * 1024), b_scal = 256;
    const uint64_t per_rest = 2 * b_key + 2 * b_u32 + 2 * b_u8 + b_flag + b_counts + b_masks + b_scal;
    // the stamp tables of all searches in ONE block of their own (StampBuf: the next group of the same size takes it over with a new
    // epoch instead of refilling it); [replaced-flags of all searches] (one memset), then the rest search by search
    // (every device block is declared before the stream guard below: an error return first waits for the streams, then frees)
    StampBuf tabs;
    DevBuf big, dcur, dq, droots, dstatus, dwant, dpa, dpl, dpn;
    if (big.alloc((uint64_t)n * (b_flag + per_rest))) return ACX_E_NOMEM;
    uint8_t* p_repl = (uint8_t*)big.p;
    uint8_t* p_rest = p_repl + (uint64_t)n * b_flag;
    SearchHandles H;
    if (int rc = search_handles_take(H)) return rc;
    struct Give {
        SearchHandles& h;
        ~Give() { search_handles_give(h); }
    } give{H};
    hipStream_t st = H.st;
    if (int rc = tabs.alloc((uint64_t)n * b_tab, st)) return rc;
    uint8_t* p_tab = (uint8_t*)tabs.p;
    ACX_HIP_TRY(hipMemsetAsync(p_repl, 0, (uint64_t)n * b_flag, st));  // once: k_bfs_count zeroes what a batch sets
    std::vector<int64_t> order[2];  // [0] general move code, [1] normal form (a root in normal form keeps its whole search there)
    std::vector<Pres<W>> roots((size_t)n);
    for (int64_t k = 0; k < n; k++) rc_out[k] = ACX_OK, solved[k] = 0, path_n[k] = 0;
    const bool general = option(ACX_OPT_GENERAL_MOVE, 0) != 0;
    for (int64_t k = 0; k < n; k++) {
        Pres<W>& root = roots[(size_t)k];
        bool ok = pack_relator<W>(rows + k * 2 * L, L, root.w0, root.n0);
        ok = pack_relator<W>(rows + k * 2 * L + L, L, root.w1, root.n1) && ok;
        if (!ok) {
            rc_out[k] = ACX_E_ROWERR;
            return fail(ACX_E_ROWERR, "acx_search_many: presentation %lld is not a zero-padded word pair over {+-1,+-2}", (long long)k);
        }
        order[is_normal_form<W>(root, cyclical != 0) && !general ? 1 : 0].push_back(k);
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
            d.epoch = tabs.epoch;
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
    // (pinned, pooled host buffers: the runtime pins and unpins the pages of a pageable target around every copy, and the device's next
    // operation then waits for the unmapping -- 10-25 ms behind the greedy sweep's 40 MB, round 6)
    HostBuf hpa, hpl, hpn;
    struct View {
        int32_t* q;
        int32_t* data() const { return q; }
    };
    View pa{nullptr}, pl{nullptr};
    if (hpn.alloc((size_t)n * 4)) return ACX_E_NOMEM;
    uint32_t* pn = (uint32_t*)hpn.p;
    memset(pn, 0, (size_t)n * 4);
    if (any_path) {
        if (hpa.alloc((size_t)n * pc * 4) || hpl.alloc((size_t)n * pc * 4)) return ACX_E_NOMEM;
        pa.q = (int32_t*)hpa.p, pl.q = (int32_t*)hpl.p;
        ACX_HIP_TRY(hipMemcpyAsync(dwant.p, want.data(), (size_t)n * 4, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_paths_many<W>, sgrid, sblock, 0, st, q, un, (const uint32_t*)dwant.p, (int32_t*)dpa.p, (int32_t*)dpl.p, (uint32_t*)dpn.p, (long long)pc);
        ACX_HIP_TRY(hipGetLastError());
        ACX_HIP_TRY(hipMemcpyAsync(pn, dpn.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
        ACX_HIP_TRY(hipMemcpyAsync(pa.data(), dpa.p, (size_t)n * pc * 4, hipMemcpyDeviceToHost, st));
        ACX_HIP_TRY(hipMemcpyAsync(pl.data(), dpl.p, (size_t)n * pc * 4, hipMemcpyDeviceToHost, st));
    }
    ACX_HIP_TRY(hipEventRecord(evs.b, st));
    ACX_HIP_TRY(hipStreamSynchronize(st));
    float ms = 0;
    ACX_HIP_TRY(hipEventElapsedTime(&ms, evs.a, evs.b));
    if (g_debug) fprintf(stderr, "[acx_bfs_many] %lld searches (%u general), %llu rounds of <= %u parents, group %.2f ms\n", (long long)n, n_gen, (unsigned long long)rounds, bmax, ms);
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

}  // namespace acx

using namespace acx;

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
    bool sched = kind == ACX_SEARCH_GREEDY && !option(ACX_OPT_GREEDY_HOST, 0) && !t_minima_on && !g_digest_on.load();
    for (int g = 0; g < n_groups; g++) sched = sched && L[g] <= 61;
    if (!sched) {
        // The batches through acx_search_many.  bfs: a batch fills the GPU by itself (acx_bfs_many.h) -- until its last searches are left: TWO
        // batches in flight (two host threads, each with its own stream) let the next one's rounds fill the chip while the running one's tail
        // drains.  Measured on the 1190-presentation sweep (tools/scratch/bfs_sweep_overlap.py): 0.137 s one after the other, 0.129 with two in
        // flight, 0.130 with three, 0.19-0.25 with all seven (round 4: their tables evict each other from the caches).
        const int in_flight = kind == ACX_SEARCH_BFS && n_groups > 1 && !t_minima_on && !g_digest_on.load() ? 2 : 1;
        int dev = 0;
        (void)hipGetDevice(&dev);
        std::atomic<int> next(0), first_rc(ACX_OK);
        std::mutex err_mu;
        std::string first_err;
        auto work = [&](bool own_thread) {
            if (own_thread) (void)hipSetDevice(dev);
            for (;;) {
                const int g = next.fetch_add(1);
                if (g >= n_groups || first_rc.load() != ACX_OK) return;
                const int64_t o = out0[(size_t)g];
                const int rc = acx_search_many(kind, h_presentations[g], n[g], L[g], max_nodes, cyclical, 16, solved + o, path_action ? path_action + o * path_cap : nullptr,
                                               path_len ? path_len + o * path_cap : nullptr, path_cap, path_n + o, stats ? stats + o : nullptr, rc_out + o);
                if (rc != ACX_OK) {
                    std::lock_guard<std::mutex> lock(err_mu);
                    if (first_rc.load() == ACX_OK) {
                        first_rc.store(rc);
                        first_err = acx_last_error();  // (the message is thread-local: carried to the caller's thread below)
                    }
                }
            }
        };
        {
            struct Joined {
                std::vector<std::thread> t;
                ~Joined() {
                    for (auto& x : t)
                        if (x.joinable()) x.join();
                }
            } others;
            for (int k = 1; k < in_flight; k++) others.t.emplace_back(work, true);
            work(false);
        }
        if (first_rc.load() != ACX_OK) return fail(first_rc.load(), "%s", first_err.c_str());
        return ACX_OK;
    }
    // greedy_search: ALL batches as jobs of one launch per key width (64-bit keys up to max_relator_length 29, 128-bit above), the two
    // launches side by side over ONE pool of slots
    std::vector<SearchGroupIn> narrow, wide;
    int L_max = 1;
    for (int g = 0; g < n_groups; g++)
        if (n[g]) (L[g] <= 29 ? narrow : wide).push_back(SearchGroupIn{h_presentations[g], n[g], L[g], out0[(size_t)g]}), L_max = std::max(L_max, (int)L[g]);
    const int64_t n_all = out0[(size_t)n_groups];
    if (n_all == 0) return ACX_OK;
    std::vector<uint8_t> rerun((size_t)n_all, 0);
    int dev = 0;
    (void)hipGetDevice(&dev);
    int64_t n_narrow = 0, n_wide = 0;
    for (const auto& gr : narrow) n_narrow += gr.n;
    for (const auto& gr : wide) n_wide += gr.n;
    const auto t_call = std::chrono::steady_clock::now();
    auto since_call = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_call).count(); };
    GreedySlots pool;
    {
        StreamLease l0;
        if (l0.take(0) != ACX_OK) return ACX_E_NODEVICE;
        hipStream_t st0 = l0.s;
        int rc0 = pool.setup(n_all, max_nodes, L_max, !wide.empty(), greedy_slots_wanted(), st0);
        if (rc0 == ACX_OK && hipStreamSynchronize(st0) != hipSuccess) rc0 = fail(ACX_E_NODEVICE, "acx_search_groups: setting up the slots failed");
        if (rc0 != ACX_OK) return rc0;
    }
    const double t_setup = since_call();
    // Workgroups per launch.  With a slot for every workgroup the chip can hold (two per compute unit) both launches get as many
    // workgroups as they have jobs: those beyond the chip's capacity wait in the dispatcher, and whichever launch runs out of jobs first
    // leaves its compute units (and slots) to the other -- no shares to guess.  With fewer slots (the tests' option, little free
    // memory) a waiting workgroup would hold a compute unit while it spins for a slot, so the slots are shared out in proportion to
    // the expected work: a 128-bit search costs ~1.7 x a 64-bit one (measured on the Miller-Schupp sweep).
    uint32_t wgs_wide = pool.S, wgs_narrow = pool.S;
    if (n_narrow && n_wide && pool.S < greedy_resident()) {
        const double share = 1.7 * (double)n_wide / (1.7 * (double)n_wide + (double)n_narrow);
        wgs_wide = (uint32_t)std::min<double>(std::max<double>(1.0, share * pool.S + 0.5), std::max<double>(1.0, (double)pool.S - 1.0));
        wgs_narrow = std::max<uint32_t>(1u, pool.S - wgs_wide);
    }
    // The 128-bit launch goes first (and its stream has the higher priority): its searches are the uniformly long ones (nine in ten of the
    // Miller-Schupp sweep's stay unsolved), so they should all be on the chip from the start; the 64-bit launch, with its many short
    // searches, fills in as compute units come free.  Measured: 0.149 s this way round, 0.158-0.161 s the other.
    std::atomic<int> wide_launched{0};
    int rc_wide = ACX_OK;
    std::string err_wide;
    struct Joined {  // (whatever leaves this function first -- an exception of a host allocation included -- the side thread is joined)
        std::thread t;
        ~Joined() {
            if (t.joinable()) t.join();
        }
    } side_guard;
    std::thread& side = side_guard.t;
    if (!wide.empty() && !narrow.empty())
        side = std::thread([&]() {
            (void)hipSetDevice(dev);
            rc_wide = run_greedy_sched<u128>(pool, wide, max_nodes, cyclical, solved, path_action, path_len, path_cap, path_n, stats, rc_out, rerun.data(), wgs_wide, &wide_launched);
            if (rc_wide != ACX_OK) err_wide = acx_last_error();
        });
    int rc = ACX_OK;
    if (!narrow.empty()) {
        rc = run_greedy_sched<uint64_t>(pool, narrow, max_nodes, cyclical, solved, path_action, path_len, path_cap, path_n, stats, rc_out, rerun.data(), wgs_narrow, nullptr,
                                        side.joinable() ? &wide_launched : nullptr);
    }
    if (side.joinable()) side.join();
    else if (!wide.empty()) rc_wide = run_greedy_sched<u128>(pool, wide, max_nodes, cyclical, solved, path_action, path_len, path_cap, path_n, stats, rc_out, rerun.data(), wgs_wide);
    if (g_debug) {
        size_t clean = 0;
        for (uint8_t c : pool.was_clean) clean += c;
        fprintf(stderr, "[acx_search_groups] %u slots of %.0f MB in %zu blocks (%zu of them came back clean from the pool) set up in %.1f ms; both launches done at %.1f ms\n", pool.S, pool.per_slot / 1e6,
                pool.groups.size(), clean, t_setup, since_call());
    }
    if (rc == ACX_OK && rc_wide == ACX_OK) pool.mark_clean();
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
    // max_relator_length 62 .. 64 (acx_keys.h): the reduced-word key type when every root is freely reduced (a presentation that is not goes through
    // acx_search below, which says so)
    bool long_ok = L >= 62 && L <= 64;
    for (int64_t k = 0; long_ok && k < n; k++)
        for (int h = 0; h < 2 && long_ok; h++) {
            const int8_t* w = h_presentations + (k * 2 + h) * L;
            for (int i = 0; i + 1 < L && w[i + 1] != 0; i++)
                if (w[i] == -w[i + 1]) long_ok = false;
        }
    const bool fits = L >= 1 && (L <= 61 || long_ok);
    if (kind == ACX_SEARCH_GREEDY && n > 1 && fits && !option(ACX_OPT_GREEDY_HOST, 0) && !t_minima_on && !g_digest_on.load()) {
        // greedy: the searches as jobs on a fixed set of workgroup slots (k_greedy_sched)
        if (max_nodes < 0) max_nodes = 0;
        std::vector<uint8_t> rerun((size_t)n, 0);
        const std::vector<SearchGroupIn> one{SearchGroupIn{h_presentations, n, L, 0}};
        GreedySlots pool;
        {
            StreamLease l0;
            if (l0.take(0) != ACX_OK) return ACX_E_NODEVICE;
            hipStream_t st0 = l0.s;
            int rc0 = pool.setup(n, max_nodes, L, L > 29, greedy_slots_wanted(), st0);
            if (rc0 == ACX_OK && hipStreamSynchronize(st0) != hipSuccess) rc0 = fail(ACX_E_NODEVICE, "acx_search_many: setting up the slots failed");
            if (rc0 != ACX_OK) return rc0;
        }
        const int rc = L <= 29   ? run_greedy_sched<uint64_t>(pool, one, max_nodes, cyclical, solved, path_action, path_len, path_cap, path_n, stats, rc_out, rerun.data(), pool.S)
                       : L <= 61 ? run_greedy_sched<u128>(pool, one, max_nodes, cyclical, solved, path_action, path_len, path_cap, path_n, stats, rc_out, rerun.data(), pool.S)
                                 : run_greedy_sched<u128x>(pool, one, max_nodes, cyclical, solved, path_action, path_len, path_cap, path_n, stats, rc_out, rerun.data(), pool.S);
        if (rc != ACX_OK) return rc;
        pool.mark_clean();
        for (int64_t k = 0; k < n; k++)
            if (rerun[k])
                rc_out[k] = acx_search(kind, h_presentations + k * 2 * L, L, max_nodes, cyclical, solved + k, path_action ? path_action + k * path_cap : nullptr,
                                       path_len ? path_len + k * path_cap : nullptr, path_cap, path_n + k, stats ? stats + k : nullptr);
        for (int64_t k = 0; k < n; k++)
            if (rc_out[k] != ACX_OK && rc_out[k] != ACX_E_CAPACITY) return fail(ACX_E_ROWERR, "acx_search_many: search %lld failed with code %d", (long long)k, rc_out[k]);
        return ACX_OK;
    }
    if (kind == ACX_SEARCH_BFS && n > 1 && fits && !t_minima_on && !g_digest_on.load()) {
        // bfs: groups of searches sharing the launches of the fused single search, a batch of every search per round (acx_bfs_many.h)
        if (max_nodes < 0) max_nodes = 0;
        const double nn = (double)std::max<int64_t>(max_nodes, 1);
        const double per_search = (L <= 29 ? 26.0 : 42.0) * nn + 32.0 * (nn + 12.0 * bfs_many_bmax()) + 64.0 * bfs_many_bmax() + 1e6;
        const int64_t group = (int64_t)std::max(1.0, std::min(4096.0, group_byte_budget(48e9) / per_search));
        for (int64_t k0 = 0; k0 < n; k0 += group) {
            const int64_t m = std::min<int64_t>(group, n - k0);
            int32_t* pa = path_action ? path_action + k0 * path_cap : nullptr;
            int32_t* pl = path_len ? path_len + k0 * path_cap : nullptr;
            acx_search_stats* ps = stats ? stats + k0 : nullptr;
            const int8_t* pr = h_presentations + k0 * 2 * L;
            const int rc = L <= 29   ? run_bfs_group_fused<uint64_t>(pr, m, L, max_nodes, cyclical, solved + k0, pa, pl, path_cap, path_n + k0, ps, rc_out + k0)
                           : L <= 61 ? run_bfs_group_fused<u128>(pr, m, L, max_nodes, cyclical, solved + k0, pa, pl, path_cap, path_n + k0, ps, rc_out + k0)
                                     : run_bfs_group_fused<u128x>(pr, m, L, max_nodes, cyclical, solved + k0, pa, pl, path_cap, path_n + k0, ps, rc_out + k0);
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
