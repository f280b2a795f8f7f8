// acx_shard_run.hip -- a whole sharded bfs as ONE C call (round 6): acx_bfs_sharded, and the RCCL communicator it runs on.
//
// Reference: bfs(presentation, max_nodes_to_explore, verbose, cyclically_reduce_after_moves), ac_solver/search/breadth_first.py:15-97.
// Rounds 3-5 drove the per-GPU engine (acx_shard.hip, acx_shard_*) chunk by chunk from Python (ac_solver/search/sharded.py: five or
// six ctypes calls and two torch.distributed calls per chunk).  This file is that orchestration in C++ on top of the SAME engine
// entry points -- the chunk loop, the two-stream pipeline, the lagged control block, the adaptive region capacity, the replicated small
// levels and the failure protocol are sharded.py's, statement for statement, and the Python orchestrator stays as the reference the
// tests compare this one with (tests/test_gpu_search.py: same result, same node numbering, same number of collectives).
//
// The collectives go through an acx_comm: two function pointers (equal-split all-to-all of int64 words, in-place all-reduce), so the
// library links no collective library.  acx_comm_rccl fills one in for a caller-provided ncclComm_t: librccl is resolved at run
// time (the copy the process has already loaded -- torch's --, else librccl.so), and torch.distributed hands the communicator of
// a process group out (ProcessGroupNCCL._comm_ptr()).  The GPU tests also run thread ranks through callbacks.
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <deque>
#include <mutex>
#include <string>
#include <vector>

#include "acx_frontier.h"  // DevBuf: device blocks from the library's pool

namespace acx {

namespace {

constexpr int64_t kInf = 1ll << 62;
constexpr int kLag = 2;  // chunks the host runs ahead of the control block it reads (sharded.py: LAG)
constexpr int kFillDefault = 320, kFillHard = 1 << 20;
constexpr int64_t kReplicateBelow = 1ll << 18;  // sharded.py: REPLICATE_BELOW
constexpr int kWalkCap = 256;
enum : int { ST_RUNNING = 0, ST_SOLVED = 1, ST_BUDGET = 2, ST_MOVE_ERROR = 3, ST_FAILED = 4 };
constexpr int RC_OVERFLOW = 1;  // internal: a region overflowed under a capacity tighter than the default -> rerun

__global__ void k_dead_headers(int64_t* __restrict__ send, uint32_t regions, int64_t region_words) {
    ACX_VGPR_PAD("v15");
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= regions) return;
    int64_t* h = send + (int64_t)r * region_words;
    h[0] = 0;
    h[1] = kInf;
    h[2] = kInf;
    h[3] = 4;
}

// ONE side stream per device for the life of the process (a new stream per search pays for a hardware queue each time)
hipStream_t side_stream_of(int dev) {
    static std::mutex m;
    static std::vector<std::pair<int, hipStream_t>> streams;
    std::lock_guard<std::mutex> lock(m);
    for (auto& s : streams)
        if (s.first == dev) return s.second;
    hipStream_t st = nullptr;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return nullptr;
    streams.emplace_back(dev, st);
    return st;
}

struct EventRing {
    std::vector<hipEvent_t> ev;
    size_t next = 0;
    int init(int n) {
        ev.assign((size_t)n, nullptr);
        for (auto& e : ev) ACX_HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        return ACX_OK;
    }
    hipEvent_t take() { return ev[next++ % ev.size()]; }
    ~EventRing() {
        for (auto e : ev)
            if (e) (void)hipEventDestroy(e);
    }
};

struct Produced {
    int64_t n_par = 0;
    hipEvent_t ev = nullptr;
    bool dead = false;
    bool valid = false;
};

struct ShardRun {
    // ---- inputs
    const int8_t* pres;
    int L, cyc;
    int64_t max_nodes;
    const acx_comm* comm;
    const acx_comm* mcomm;  // the mask all-reduce's communicator (the same one unless the caller gave a second)
    hipStream_t main_st, side_st;
    int world, rank;
    int64_t B, replicate_below;
    int region_fill;  // < 0: adaptive
    int fail_at_call, fail_rank;
    // ---- engine and buffers
    acx_shard* h = nullptr;
    DevBuf log, send, gmask, small, dead_send, dead_recv, dead_mask;
    int64_t log_words = 0, send_words = 0, cursor = 8;
    int KW = 2;
    bool eng_replicated = false;
    // ---- state (names as in sharded.py: _bfs_sharded_run)
    bool exchange = false, replicating = false, adaptive = false, tight_used = false, failed = false, phase_closed = true;
    std::string fail_msg;
    int fill = kFillDefault;
    int64_t fail_hdr_chunk = -1;
    int64_t ctl[ACX_SHARD_CTL_WORDS] = {};
    bool have_ctl = false;
    int64_t F = 1, F_prev = 0, nodes_seen = 1, levels = 0, chunks = 0, repl_levels = 0, min_len = kInf;
    int calls = 0;
    EventRing ev_ready, ev_done;
    std::vector<hipEvent_t> done;  // per consumed chunk of the running level: behind its commit on the main stream
    acx_shard_run_stats st = {};
    // results
    bool solved = false;
    std::vector<std::pair<int32_t, int32_t>> path;

    ~ShardRun() {
        if (h) {
            (void)hipStreamSynchronize(side_st);  // (a chunk that was expanded but never consumed may still be running)
            (void)hipStreamSynchronize(main_st);
            acx_shard_destroy(h);
        }
    }

    int world_eff() const { return eng_replicated ? 1 : world; }
    // an engine call goes through ECALL (test hook: opts.fail_at_call makes this rank's n-th one fail INSTEAD of being made)
    bool hook() {
        calls++;
        return fail_at_call > 0 && rank == fail_rank && calls == fail_at_call;
    }
#define ECALL(expr) (hook() ? fail(ACX_E_NODEVICE, "engine call failed (simulated): call %d, level %lld, %.40s", calls, (long long)levels, #expr) : (expr))
    void set_failed() {
        if (!failed) fail_msg = last_error_buf();
        failed = true;
        (void)acx_shard_fail(h, main_st);  // sticky on the device: the headers of every later chunk this engine expands say so
    }
    int layout(int64_t n_par, int fq, int64_t* S, int64_t* cap, int64_t* rw) { return acx_shard_layout(n_par, world_eff(), KW, fq, S, cap, rw); }
    int64_t layout_words(int64_t n_par, int fq) {
        int64_t S = 0, cap = 0, rw = 0;
        (void)layout(n_par, fq, &S, &cap, &rw);
        return S * world_eff() * rw;
    }
    int attach() {
        return acx_shard_attach(h, (int64_t*)log.p, log_words, world > 1 ? (int64_t*)send.p : nullptr, world > 1 ? send_words : 0, (int32_t*)gmask.p);
    }
    int all_to_all(const int64_t* s, int64_t* r, int64_t words, hipStream_t stream) {
        st.all_to_all_calls++;
        st.all_to_all_bytes += words * 8;
        if (comm->all_to_all(comm->ctx, s, r, words, stream) != 0) return fail(ACX_E_NODEVICE, "acx_bfs_sharded: the communicator's all_to_all failed");
        return ACX_OK;
    }
    int all_reduce(const acx_comm* c, void* buf, int64_t n, int dtype, int op, hipStream_t stream) {
        st.all_reduce_calls++;
        st.all_reduce_bytes += n * (dtype == ACX_I32 ? 4 : 8);
        if (c->all_reduce(c->ctx, buf, n, dtype, op, stream) != 0) return fail(ACX_E_NODEVICE, "acx_bfs_sharded: the communicator's all_reduce failed");
        return ACX_OK;
    }
    // small host vectors through the communicator: n <= 2 + 2 kWalkCap int64 words
    int reduce_host(int64_t* v, int n, int op) {
        ACX_HIP_TRY(hipMemcpyAsync(small.p, v, (size_t)n * 8, hipMemcpyHostToDevice, main_st));
        if (int rc = all_reduce(comm, small.p, n, ACX_I64, op, main_st)) return rc;
        ACX_HIP_TRY(hipMemcpyAsync(v, small.p, (size_t)n * 8, hipMemcpyDeviceToHost, main_st));
        ACX_HIP_TRY(hipStreamSynchronize(main_st));
        return ACX_OK;
    }

    int setup(int64_t node_cap, double est_parents) {
        h = acx_shard_create(L, cyc, node_cap, B, rank, world);
        if (!h) return ACX_E_NOMEM;
        KW = acx_shard_key_words(L);
        const int64_t full = layout_words(B, 0);
        log_words = 8 + (int64_t)((double)full * (est_parents / (double)B + 2.0)) + 64 * 40;
        if (log.alloc((size_t)log_words * 8)) return ACX_E_NOMEM;
        if (world > 1) {
            send_words = full;
            if (send.alloc((size_t)send_words * 8)) return ACX_E_NOMEM;
        }
        if (gmask.alloc((size_t)((B + 3) / 4 * 2) * 4)) return ACX_E_NOMEM;
        if (small.alloc((size_t)(2 + 2 * kWalkCap) * 8)) return ACX_E_NOMEM;
        if (int rc = ev_ready.init(16)) return rc;
        if (int rc = ev_done.init(16)) return rc;
        return attach();
    }

    // sharded.py: HipShardEngine.chunk_expand -- the log grows by doubling when the host-side cursor says the next chunk would not fit
    int chunk_expand(int64_t c0, int64_t c1, bool level_first, int fq, int64_t** send_out, int64_t** recv_out, int64_t* words_out, hipStream_t stream) {
        const int64_t need = layout_words(c1 - c0, fq);
        if (cursor + need > log_words) {
            ACX_HIP_TRY(hipDeviceSynchronize());
            DevBuf bigger;
            const int64_t nw = std::max<int64_t>(2 * log_words, cursor + 2 * need);
            if (bigger.alloc((size_t)nw * 8)) return ACX_E_NOMEM;
            ACX_HIP_TRY(hipMemcpy(bigger.p, log.p, (size_t)cursor * 8, hipMemcpyDeviceToDevice));
            std::swap(log, bigger);  // (the old block goes back to the pool when `bigger` leaves the scope)
            log_words = nw;
            if (int rc = attach()) return rc;
            ACX_HIP_TRY(hipDeviceSynchronize());
        }
        if (world > 1 && !eng_replicated && need > send_words) {  // (only under the hard bound of the last resort)
            ACX_HIP_TRY(hipDeviceSynchronize());
            DevBuf bigger;
            if (bigger.alloc((size_t)need * 8)) return ACX_E_NOMEM;
            std::swap(send, bigger);
            send_words = need;
            if (int rc = attach()) return rc;
        }
        int64_t off = 0, words = 0;
        if (int rc = ECALL(acx_shard_chunk_expand(h, c0, c1, level_first ? 1 : 0, fq, &off, &words, stream))) return rc;
        if (off != cursor || words != need) return fail(ACX_E_INVAL, "acx_bfs_sharded: the engine's log cursor (%lld, %lld words) is not the host's (%lld, %lld)", (long long)off, (long long)words, (long long)cursor, (long long)need);
        *recv_out = (int64_t*)log.p + off;
        *send_out = (world == 1 || eng_replicated) ? *recv_out : (int64_t*)send.p;
        *words_out = need;
        cursor += need;
        return ACX_OK;
    }

    // expansion + exchange of one chunk on the side stream (sharded.py: produce)
    int produce(int64_t c0, int64_t c1, int64_t idx, Produced* out) {
        const int64_t n_par = c1 - c0;
        tight_used = tight_used || (exchange && fill > 0 && fill < kFillHard);
        if (side_st != main_st && !done.empty()) ACX_HIP_TRY(hipStreamWaitEvent(side_st, done.back(), 0));  // at most ONE chunk ahead of the dedup
        bool dead = failed && exchange;
        int64_t *sp = nullptr, *rp = nullptr, words = 0;
        if (!dead) {
            if (int rc = chunk_expand(c0, c1, c0 == 0, fill, &sp, &rp, &words, side_st)) {
                if (!exchange) return rc;
                set_failed();
                dead = true;
            }
        }
        if (dead) {
            int64_t S = 0, cap = 0, rw = 0;
            (void)layout(n_par, fill, &S, &cap, &rw);
            words = S * world * rw;
            if ((size_t)words * 8 > dead_send.bytes || !dead_send.p) {
                ACX_HIP_TRY(hipDeviceSynchronize());
                dead_send.release();
                dead_recv.release();
                if (dead_send.alloc((size_t)words * 8) || dead_recv.alloc((size_t)words * 8)) return ACX_E_NOMEM;
            }
            sp = (int64_t*)dead_send.p;
            rp = (int64_t*)dead_recv.p;
            const uint32_t regions = (uint32_t)(S * world);
            hipLaunchKernelGGL(k_dead_headers, dim3((regions + 255) / 256), dim3(256), 0, side_st, sp, regions, rw);
            if (fail_hdr_chunk < 0) fail_hdr_chunk = idx;
        }
        if (exchange)
            if (int rc = all_to_all(sp, rp, words, side_st)) return rc;  // (a communicator that fails has no protocol to fall back on)
        hipEvent_t ev = ev_ready.take();
        ACX_HIP_TRY(hipEventRecord(ev, side_st));
        out->n_par = n_par;
        out->ev = ev;
        out->dead = dead;
        out->valid = true;
        return ACX_OK;
    }

    bool slot_ok[4] = {false, false, false, false};  // a snapshot that was never taken leaves WHATEVER an earlier chunk (or search: the pinned
                                                       // slots are pooled) wrote in its slot: such a slot is not read
    int ctl_snapshot(int slot) {
        slot_ok[slot] = true;
        if (int rc = ECALL(acx_shard_ctl_snapshot(h, slot, main_st))) {
            if (!exchange) return rc;
            set_failed();
            slot_ok[slot] = false;
        }
        return ACX_OK;
    }
    int ctl_wait(int slot) {
        int64_t got[ACX_SHARD_CTL_WORDS];
        if (int rc = slot_ok[slot] ? ECALL(acx_shard_ctl_wait(h, slot, got)) : fail(ACX_E_NODEVICE, "no snapshot in slot %d", slot)) {
            if (!exchange) return rc;
            set_failed();
            if (!have_ctl) {  // "running", nothing known: the loop goes on until the failure rule ends it
                std::fill(ctl, ctl + ACX_SHARD_CTL_WORDS, 0);
                ctl[ACX_SHARD_CTL_MIN_LEN] = kInf;
                ctl[ACX_SHARD_CTL_NODES_GLOBAL] = nodes_seen;
                have_ctl = true;
            }
            return ACX_OK;
        }
        std::copy(got, got + ACX_SHARD_CTL_WORDS, ctl);
        have_ctl = true;
        return ACX_OK;
    }

    // sharded.py: next_size -- near the end of the budget the chunks shrink to what the remaining budget is expected to need
    int64_t next_size(int64_t c0, const std::vector<int64_t>& sizes, size_t n_read, int64_t new_read) {
        const int64_t n = std::min<int64_t>(B, F - c0), min_chunk = std::min<int64_t>(B, 1 << 16);
        int64_t num, den;
        if (n_read > 0) {
            num = new_read;
            den = 0;
            for (size_t i = 0; i < n_read; i++) den += sizes[i];
        } else if (F_prev > 0) {
            num = F;
            den = F_prev;
        } else {
            return n;
        }
        if (num <= 0) return n;
        int64_t in_flight = 0;
        for (size_t i = n_read; i < sizes.size(); i++) in_flight += sizes[i];
        const int64_t remaining = max_nodes - nodes_seen - (in_flight * num + den - 1) / den;
        const __int128 top = (__int128)std::max<int64_t>(remaining, 0) * den * 9;
        const __int128 bot = (__int128)num * 8;
        int64_t want = (int64_t)((top + bot - 1) / bot);
        want = std::max<int64_t>(min_chunk, want);
        want = (want + 2047) / 2048 * 2048;
        return std::min<int64_t>(n, want);
    }

    int raise_failed(int64_t code) {
        if (code == 1 && !failed && tight_used) return RC_OVERFLOW;
        static const char* text[] = {"engine failure", "a send region or the record log overflowed", "node capacity exceeded", "visited table full", "engine call failed"};
        const bool mine = failed || (have_ctl && ctl[ACX_SHARD_CTL_FAIL_LOCAL]);
        if (mine) {
            const int64_t lc = have_ctl ? ctl[ACX_SHARD_CTL_FAIL_LOCAL] : 4;
            const std::string why = failed ? fail_msg : std::string(text[lc >= 1 && lc <= 4 ? lc : 0]);
            return fail(ACX_E_NODEVICE, "sharded bfs failed on rank %d: %s", rank, why.c_str());
        }
        return fail(ACX_E_NODEVICE, "sharded bfs failed on another rank: %s", text[code >= 1 && code <= 4 ? code : 0]);
    }

    // sharded.py: walk -- the owner of a node walks up while the parents are its own (one launch + one copy per segment) and shares the
    // segment with ONE all-reduce; `collective` false: the search ended in its replicated phase, every rank walks its own copy
    int walk(int64_t pref, std::pair<int32_t, int32_t> tail, bool collective) {
        std::vector<std::pair<int32_t, int32_t>> rev;
        std::vector<int64_t> buf(2 + 2 * kWalkCap);
        while (pref >= 0) {
            const int64_t r = pref >> 40, nid = pref & ((1ll << 40) - 1);
            std::fill(buf.begin(), buf.end(), 0);
            if (rank == r || !collective)
                if (int rc = acx_shard_walk(h, nid, kWalkCap, buf.data(), main_st)) return rc;
            if (collective)
                if (int rc = reduce_host(buf.data(), (int)buf.size(), ACX_RED_SUM)) return rc;
            const int64_t n = buf[1];
            if (n < 1 || n > kWalkCap) return fail(ACX_E_INVAL, "acx_bfs_sharded: a path segment of %lld nodes", (long long)n);
            for (int64_t k = 0; k < n; k++) rev.emplace_back((int32_t)buf[2 + 2 * k], (int32_t)buf[3 + 2 * k]);
            pref = buf[0];
        }
        path.assign(rev.rbegin(), rev.rend());
        path.push_back(tail);
        return ACX_OK;
    }

    void fill_stats(bool ok) {
        st.nodes = ctl[ACX_SHARD_CTL_NODES_GLOBAL];
        st.expanded = ctl[ACX_SHARD_CTL_EXPANDED];
        st.local_nodes = ctl[ACX_SHARD_CTL_NODES];
        st.levels = levels;
        st.chunks = chunks;
        st.replicated_levels = repl_levels;
        st.min_len = ok ? 2 : (int32_t)std::min<int64_t>(min_len, 1 << 30);
    }

    // the level loop of sharded.py:_bfs_sharded_run; returns ACX_OK (solved / budget / frontier empty), RC_OVERFLOW, or an error
    int run_levels() {
        const bool on_side = side_st != main_st;
        while (F > 0) {
            levels++;
            std::deque<int> pending;
            int64_t k = 0;
            have_ctl = false;
            if (on_side) {  // the level's parents are the nodes the main stream committed during the previous level
                hipEvent_t e = ev_done.take();
                ACX_HIP_TRY(hipEventRecord(e, main_st));
                ACX_HIP_TRY(hipStreamWaitEvent(side_st, e, 0));
            }
            std::vector<int64_t> sizes;
            size_t n_read = 0;
            int64_t new_read = 0, c_next = 0;
            done.clear();
            auto produce_next = [&](Produced* out) -> int {
                const int64_t n = next_size(c_next, sizes, n_read, new_read);
                sizes.push_back(n);
                c_next += n;
                return produce(c_next - n, c_next, (int64_t)sizes.size() - 1, out);
            };
            Produced ready;
            if (int rc = produce_next(&ready)) return rc;
            while (ready.valid && (!have_ctl || ctl[ACX_SHARD_CTL_STATUS] == ST_RUNNING) && (fail_hdr_chunk < 0 || k <= fail_hdr_chunk + kLag)) {
                const Produced cur = ready;
                ready = Produced();
                if (c_next < F)
                    if (int rc = produce_next(&ready)) return rc;  // runs beside this chunk's dedup and commit
                ACX_HIP_TRY(hipStreamWaitEvent(main_st, cur.ev, 0));
                int32_t* masks = (int32_t*)gmask.p;
                const int64_t mask_words = (cur.n_par + 1) / 2;
                if (cur.dead) {  // never went through the engine: nothing to dedup, nothing to commit; the collectives still pair up
                    if ((size_t)mask_words * 4 > dead_mask.bytes || !dead_mask.p) {
                        dead_mask.release();
                        if (dead_mask.alloc((size_t)((B + 3) / 4 * 2) * 4)) return ACX_E_NOMEM;
                    }
                    masks = (int32_t*)dead_mask.p;
                    ACX_HIP_TRY(hipMemsetAsync(masks, 0, (size_t)mask_words * 4, main_st));
                } else if (int rc = ECALL(acx_shard_chunk_insert(h, main_st))) {
                    if (!exchange) return rc;
                    set_failed();
                    if (acx_shard_chunk_insert_dead(h, main_st) != ACX_OK)  // masks only: the chunk stays in the engine's ring
                        ACX_HIP_TRY(hipMemsetAsync(masks, 0, (size_t)mask_words * 4, main_st));
                }
                if (exchange)
                    if (int rc = all_reduce(mcomm, masks, mask_words, ACX_I32, ACX_RED_SUM, main_st)) return rc;  // every child has one owner: SUM == OR
                if (!cur.dead)
                    if (int rc = ECALL(acx_shard_chunk_commit(h, max_nodes, main_st))) {
                        if (!exchange) return rc;
                        set_failed();
                    }
                if (on_side) {
                    hipEvent_t e = ev_done.take();
                    ACX_HIP_TRY(hipEventRecord(e, main_st));
                    done.push_back(e);
                }
                if (int rc = ctl_snapshot((int)(k % (kLag + 2)))) return rc;
                pending.push_back((int)(k % (kLag + 2)));
                chunks++;
                if ((int)pending.size() > kLag) {
                    if (int rc = ctl_wait(pending.front())) return rc;
                    pending.pop_front();
                    n_read++;
                    new_read = ctl[ACX_SHARD_CTL_NEXT_COUNT];
                    nodes_seen = ctl[ACX_SHARD_CTL_NODES_GLOBAL];
                }
                k++;
            }
            if (on_side) {  // (a chunk that was produced but never consumed: the search ended)
                hipEvent_t e = ev_ready.take();
                ACX_HIP_TRY(hipEventRecord(e, side_st));
                ACX_HIP_TRY(hipStreamWaitEvent(main_st, e, 0));
            }
            while (!pending.empty() && (!have_ctl || ctl[ACX_SHARD_CTL_STATUS] == ST_RUNNING)) {  // end of the level: the one synchronisation
                if (int rc = ctl_wait(pending.front())) return rc;
                pending.pop_front();
            }
            const int64_t status = ctl[ACX_SHARD_CTL_STATUS];
            int64_t code = ctl[ACX_SHARD_CTL_FAIL_LOCAL];
            if (status == ST_FAILED) code = std::max(code, ctl[ACX_SHARD_CTL_FAIL_SEEN]);
            if (failed) code = std::max<int64_t>(code, 4);
            int64_t closing[3] = {code, -ctl[ACX_SHARD_CTL_MIN_LEN], ctl[ACX_SHARD_CTL_LEVEL_FILL]};
            // the replicated phase: ONE closing all-reduce, when the phase ends (sharded.py)
            const bool phase_end = replicating && (status != ST_RUNNING || ctl[ACX_SHARD_CTL_NEXT_COUNT] >= replicate_below || ctl[ACX_SHARD_CTL_NEXT_COUNT] == 0);
            if (replicating) repl_levels++;
            if (phase_end) phase_closed = true;
            if (exchange || phase_end)
                if (int rc = reduce_host(closing, 3, ACX_RED_MAX)) return rc;
            min_len = std::min(min_len, -closing[1]);
            if (closing[0]) return raise_failed(closing[0]);
            if (status == ST_MOVE_ERROR) return fail(ACX_E_ROWERR, "a move emptied a relator during the search: the reference's ACMove raises here");
            if (status == ST_SOLVED) {
                const int64_t tag = ctl[ACX_SHARD_CTL_SOLVED_TAG];
                int64_t mine = -1;
                if (int rc = acx_shard_find(h, tag / 12, &mine, main_st)) return rc;
                solved = true;
                fill_stats(true);
                if (replicating) return walk(((int64_t)rank << 40) | mine, {(int32_t)(tag % 12), 2}, false);
                int64_t pref = mine >= 0 ? (((int64_t)rank << 40) | mine) : -1;
                if (world > 1)
                    if (int rc = reduce_host(&pref, 1, ACX_RED_MAX)) return rc;
                return walk(pref, {(int32_t)(tag % 12), 2}, world > 1);
            }
            if (status == ST_BUDGET) {
                fill_stats(false);
                return ACX_OK;
            }
            F_prev = F;
            F = ctl[ACX_SHARD_CTL_NEXT_COUNT];
            nodes_seen = ctl[ACX_SHARD_CTL_NODES_GLOBAL];
            if (replicating && phase_end && F > 0) {
                replicating = false;
                exchange = true;
                adaptive = region_fill < 0;
                const int rc = ECALL(acx_shard_partition(h, main_st));
                eng_replicated = false;
                if (rc) set_failed();  // goes on as a failed rank: the others learn it from the headers of its dead chunks
            }
            if (adaptive) {
                const int64_t lf = closing[2];
                fill = lf == 0 ? kFillDefault : (int)std::min<int64_t>(kFillDefault, std::max<int64_t>(24, lf * 5 / 4 + 12));
            }
        }
        fill_stats(false);
        return ACX_OK;
    }

    int run() {
        // root: every rank holds the replicated levels; else node 0 of its owner, global frontier position 0
        int64_t rec[8] = {};
        std::vector<int8_t> row(pres, pres + 2 * L);
        if (int rc = acx_shard_root_record(h, row.data(), rec)) return rc;
        replicating = world > 1 && replicate_below > 1;
        exchange = world > 1 && !replicating;
        phase_closed = !replicating;
        adaptive = region_fill < 0 && exchange;
        fill = region_fill < 0 ? kFillDefault : region_fill;
        if (replicating) {
            if (int rc = acx_shard_set_replicated(h, 1)) return rc;
            eng_replicated = true;
            if (int rc = acx_shard_seed(h, rec, main_st)) return rc;
        } else {
            const int owner = world == 1 ? 0 : acx_shard_owner(L, rec, world);
            if (owner < 0) return owner;
            if (int rc = acx_shard_seed(h, rank == owner ? rec : nullptr, main_st)) return rc;
        }
        int rc = run_levels();
        if (rc < 0 && !phase_closed && world > 1) {
            // an engine call of the replicated phase failed on THIS rank: the healthy ranks meet at the phase's closing all-reduce -- join it
            const std::string msg = last_error_buf();
            phase_closed = true;
            int64_t closing[3] = {4, 0, 0};
            (void)reduce_host(closing, 3, ACX_RED_MAX);
            return fail(rc, "sharded bfs failed on rank %d: %s", rank, msg.c_str());
        }
        return rc;
    }
};

// ---- RCCL, resolved at run time ------------------------------------------------------------------------------------------------
typedef int (*fn_allreduce)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*fn_alltoall)(const void*, void*, size_t, int, void*, hipStream_t);
typedef int (*fn_sendrecv)(void*, size_t, int, int, void*, hipStream_t);
typedef int (*fn_send)(const void*, size_t, int, int, void*, hipStream_t);
typedef int (*fn_void)(void);
typedef int (*fn_commint)(void*, int*);
typedef int (*fn_uid)(void*);
typedef const char* (*fn_errstr)(int);
struct UniqueId {
    char internal[128];
};
typedef int (*fn_init)(void**, int, UniqueId, int);
typedef int (*fn_destroy)(void*);

struct Rccl {
    void* lib = nullptr;
    fn_allreduce all_reduce = nullptr;
    fn_alltoall all_to_all = nullptr;
    fn_send send = nullptr;
    fn_sendrecv recv = nullptr;
    fn_void group_start = nullptr, group_end = nullptr;
    fn_commint count = nullptr, user_rank = nullptr;
    fn_uid unique_id = nullptr;
    fn_init init_rank = nullptr;
    fn_destroy destroy = nullptr;
    fn_errstr err = nullptr;
    bool ok = false;
};

Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, []() {
        // the copy the process already runs (torch.distributed's backend "nccl" loads torch/lib/librccl.so: a communicator made by it must
        // be driven by it), else the system's
        for (const char* name : {"librccl.so", "librccl.so.1"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
            if (r.lib) break;
        }
        if (!r.lib)
            for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
                r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
                if (r.lib) break;
            }
        if (!r.lib) return;
        r.all_reduce = (fn_allreduce)dlsym(r.lib, "ncclAllReduce");
        r.all_to_all = (fn_alltoall)dlsym(r.lib, "ncclAllToAll");
        r.send = (fn_send)dlsym(r.lib, "ncclSend");
        r.recv = (fn_sendrecv)dlsym(r.lib, "ncclRecv");
        r.group_start = (fn_void)dlsym(r.lib, "ncclGroupStart");
        r.group_end = (fn_void)dlsym(r.lib, "ncclGroupEnd");
        r.count = (fn_commint)dlsym(r.lib, "ncclCommCount");
        r.user_rank = (fn_commint)dlsym(r.lib, "ncclCommUserRank");
        r.unique_id = (fn_uid)dlsym(r.lib, "ncclGetUniqueId");
        r.init_rank = (fn_init)dlsym(r.lib, "ncclCommInitRank");
        r.destroy = (fn_destroy)dlsym(r.lib, "ncclCommDestroy");
        r.err = (fn_errstr)dlsym(r.lib, "ncclGetErrorString");
        r.ok = r.all_reduce && r.send && r.recv && r.group_start && r.group_end && r.count && r.user_rank && r.unique_id && r.init_rank && r.destroy;
    });
    return r;
}

int rccl_check(int rc, const char* what) {
    if (rc == 0) return ACX_OK;
    Rccl& r = rccl();
    return fail(ACX_E_NODEVICE, "%s failed: %s", what, r.err ? r.err(rc) : "RCCL error");
}

constexpr int kNcclInt32 = 2, kNcclInt64 = 4, kNcclSum = 0, kNcclMax = 2;

int rccl_all_to_all(void* ctx, const int64_t* d_send, int64_t* d_recv, int64_t words, void* stream) {
    Rccl& r = rccl();
    int world = 0;
    if (int rc = rccl_check(r.count(ctx, &world), "ncclCommCount")) return rc;
    if (world < 1 || words % world) return fail(ACX_E_INVAL, "acx_comm (RCCL): all_to_all of %lld words over %d ranks", (long long)words, world);
    const size_t k = (size_t)(words / world);
    if (d_send == d_recv && world == 1) return ACX_OK;
    if (r.all_to_all) return rccl_check(r.all_to_all(d_send, d_recv, k, kNcclInt64, ctx, (hipStream_t)stream), "ncclAllToAll");
    if (int rc = rccl_check(r.group_start(), "ncclGroupStart")) return rc;
    for (int p = 0; p < world; p++) {
        (void)r.send(d_send + (size_t)p * k, k, kNcclInt64, p, ctx, (hipStream_t)stream);
        (void)r.recv(d_recv + (size_t)p * k, k, kNcclInt64, p, ctx, (hipStream_t)stream);
    }
    return rccl_check(r.group_end(), "ncclGroupEnd");
}

int rccl_all_reduce(void* ctx, void* d_buf, int64_t n, int dtype, int op, void* stream) {
    Rccl& r = rccl();
    if ((dtype != ACX_I32 && dtype != ACX_I64) || (op != ACX_RED_SUM && op != ACX_RED_MAX)) return fail(ACX_E_INVAL, "acx_comm (RCCL): all_reduce dtype / op");
    return rccl_check(r.all_reduce(d_buf, d_buf, (size_t)n, dtype == ACX_I32 ? kNcclInt32 : kNcclInt64, op == ACX_RED_SUM ? kNcclSum : kNcclMax, ctx, (hipStream_t)stream),
                      "ncclAllReduce");
}

bool valid_presentation(const int8_t* p, int L) {  // envs/utils.py:13-54 (is_array_valid_presentation) on a 2L row
    for (int h = 0; h < 2; h++) {
        const int8_t* w = p + h * L;
        if (w[0] == 0) return false;
        bool pad = false;
        for (int i = 0; i < L; i++) {
            if (w[i] == 0) pad = true;
            else if (pad) return false;
        }
    }
    return true;
}

}  // namespace

}  // namespace acx

using namespace acx;

extern "C" {

int acx_rccl_available(void) { return rccl().ok ? 1 : 0; }

int acx_comm_rccl(void* nccl_comm, acx_comm* out) {
    if (!nccl_comm || !out) return fail(ACX_E_INVAL, "acx_comm_rccl: bad argument");
    Rccl& r = rccl();
    if (!r.ok) return fail(ACX_E_NODEVICE, "acx_comm_rccl: librccl.so could not be loaded");
    int rank = 0, world = 0;
    if (int rc = rccl_check(r.user_rank(nccl_comm, &rank), "ncclCommUserRank")) return rc;
    if (int rc = rccl_check(r.count(nccl_comm, &world), "ncclCommCount")) return rc;
    out->rank = rank;
    out->world = world;
    out->ctx = nccl_comm;
    out->all_to_all = rccl_all_to_all;
    out->all_reduce = rccl_all_reduce;
    return ACX_OK;
}

int acx_rccl_unique_id(void* id128) {
    if (!id128) return fail(ACX_E_INVAL, "acx_rccl_unique_id: bad argument");
    Rccl& r = rccl();
    if (!r.ok) return fail(ACX_E_NODEVICE, "acx_rccl_unique_id: librccl.so could not be loaded");
    return rccl_check(r.unique_id(id128), "ncclGetUniqueId");
}

int acx_rccl_comm_create(const void* id128, int rank, int world, void** nccl_comm) {
    if (!id128 || !nccl_comm || world < 1 || rank < 0 || rank >= world) return fail(ACX_E_INVAL, "acx_rccl_comm_create: bad argument");
    if (!have_device()) return ACX_E_NODEVICE;
    Rccl& r = rccl();
    if (!r.ok) return fail(ACX_E_NODEVICE, "acx_rccl_comm_create: librccl.so could not be loaded");
    UniqueId id;
    memcpy(id.internal, id128, sizeof(id.internal));
    return rccl_check(r.init_rank(nccl_comm, world, id, rank), "ncclCommInitRank");
}

int acx_rccl_comm_destroy(void* nccl_comm) {
    if (!nccl_comm) return ACX_OK;
    Rccl& r = rccl();
    if (!r.ok) return fail(ACX_E_NODEVICE, "acx_rccl_comm_destroy: librccl.so could not be loaded");
    return rccl_check(r.destroy(nccl_comm), "ncclCommDestroy");
}

int acx_bfs_sharded(const int8_t* h_presentation, int L, int64_t max_nodes, int cyclical, const acx_comm* comm, const acx_shard_opts* opts, int32_t* solved,
                    int32_t* path_action, int32_t* path_len, int64_t path_cap, int64_t* path_n, acx_shard_run_stats* stats, void* stream) {
    if (!h_presentation || !solved || !path_n || max_nodes < 0 || (path_cap > 0 && (!path_action || !path_len)))
        return fail(ACX_E_INVAL, "acx_bfs_sharded: bad argument");
    if (L < 1 || L > 64) return fail(ACX_E_INVAL, "acx_bfs_sharded: max_relator_length 1 .. 64");
    if (!have_device()) return ACX_E_NODEVICE;
    if (!valid_presentation(h_presentation, L)) return fail(ACX_E_ROWERR, "acx_bfs_sharded: not a valid presentation (breadth_first.py:36)");
    acx_comm one = {0, 1, nullptr, nullptr, nullptr};
    if (!comm) comm = &one;
    if (comm->world < 1 || comm->rank < 0 || comm->rank >= comm->world || (comm->world > 1 && (!comm->all_to_all || !comm->all_reduce)))
        return fail(ACX_E_INVAL, "acx_bfs_sharded: bad communicator");
    const int world = comm->world;
    acx_shard_opts o = {};
    if (opts) o = *opts;
    int64_t batch = o.batch_parents > 0 ? o.batch_parents : (1ll << (world >= 8 ? 23 : 21));
    int64_t repl = o.replicate_below == 0 ? kReplicateBelow : (o.replicate_below < 0 ? 0 : o.replicate_below);
    if (world == 1) repl = 0;
    int dev = 0;
    ACX_HIP_TRY(hipGetDevice(&dev));
    const auto t_begin = std::chrono::steady_clock::now();
    const int fills[3] = {o.region_fill > 0 ? o.region_fill : -1, kFillDefault, kFillHard};
    const int64_t bps[3] = {batch, batch, std::min<int64_t>(batch, 1 << 17)};
    int reruns = 0;
    for (int attempt = 0; attempt < 3; attempt++) {
        if (attempt > 0 && fills[attempt] == fills[attempt - 1] && bps[attempt] == bps[attempt - 1]) continue;  // (the caller asked for the default itself)
        ShardRun R;
        R.pres = h_presentation;
        R.L = L;
        R.cyc = cyclical ? 1 : 0;
        R.max_nodes = max_nodes;
        R.comm = comm;
        R.mcomm = o.mask_comm ? o.mask_comm : comm;
        R.world = world;
        R.rank = comm->rank;
        R.main_st = (hipStream_t)stream;
        const bool overlap = o.overlap == 0 ? world > 1 : o.overlap == 2;
        R.side_st = overlap ? side_stream_of(dev) : R.main_st;
        if (overlap && !R.side_st) return fail(ACX_E_NODEVICE, "acx_bfs_sharded: no side stream");
        R.B = std::max<int64_t>(1, std::min<int64_t>(bps[attempt], std::max<int64_t>(max_nodes, 64)));
        R.replicate_below = repl;
        R.region_fill = fills[attempt];
        R.fail_at_call = o.fail_at_call;
        R.fail_rank = o.fail_rank;
        const int64_t node_cap = (world == 1 ? max_nodes + 64 : (int64_t)(2.0 * (double)max_nodes / world) + std::min<int64_t>(max_nodes, 16 * std::max<int64_t>(repl, 0))) + 4096;
        if (int rc = R.setup(node_cap, std::max(1.0, (o.log_fraction_q8 > 0 ? o.log_fraction_q8 / 256.0 : 0.5) * (double)max_nodes))) return rc;
        const auto t_ready = std::chrono::steady_clock::now();
        const int rc = R.run();
        if (rc == RC_OVERFLOW) {
            reruns++;
            continue;
        }
        if (rc < 0) return rc;
        *solved = R.solved ? 1 : 0;
        *path_n = R.solved ? (int64_t)R.path.size() : 0;
        if (stats) {
            *stats = R.st;
            stats->reruns = reruns;
            stats->setup_seconds = std::chrono::duration<double>(t_ready - t_begin).count();
            stats->loop_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_ready).count();
        }
        if (R.solved) {
            if ((int64_t)R.path.size() > path_cap) return fail(ACX_E_CAPACITY, "acx_bfs_sharded: the path has %zu entries", R.path.size());
            for (size_t i = 0; i < R.path.size(); i++) {
                path_action[i] = R.path[i].first;
                path_len[i] = R.path[i].second;
            }
        }
        return ACX_OK;
    }
    return fail(ACX_E_CAPACITY, "acx_bfs_sharded: a region overflowed under the hard capacity bound");  // (cannot happen: acx_shard_layout)
}

}  // extern "C"
