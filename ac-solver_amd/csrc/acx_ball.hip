// acx_ball.hip -- sizes of radius-r neighbourhoods in the Andrews-Curtis graph of SORTED pairs of freely reduced
// relators of unbounded length: the workload of the reference's C++ side program
// barcode_analysis/5_steps_neibourhoods (neibourhoods.cpp:18-54 `neibourhood`, AC_UTILS_no_hash.cpp:83-211
// `reduce_`, `conj0_`, `inv0_`, `concat_`, `sort_`, `move`), SURVEY section 8(f)-3.
//
// One workgroup per presentation runs its whole breadth-first ball: frontier chunk -> 12 (prime) or 14 (classic)
// children per node -> exact dedup in an open-addressed table that holds the full key (two relators of up to 192
// letters, 2 bits per letter in 6 x u64 each, + lengths) -> next frontier.  Only the NUMBER of distinct pairs is
// wanted, so duplicates need no ordering: a candidate claims an empty entry with one CAS on its stamp; a candidate
// that meets an entry claimed in the same pass compares against the claimant's key in the candidate arena.
//
// Word arithmetic: inputs are freely reduced, so a product only cancels at the junction and a conjugation only at
// the two ends (same identities as acx_word.h, on 384-bit words).
#include <string.h>

#include <algorithm>
#include <vector>

#include "acx_common.h"

namespace acx {
namespace ball {

constexpr int BN = 6;              // u64 words per relator
constexpr int kMaxLetters = 32 * BN;  // 192
constexpr int KW = 2 * BN + 1;     // u64 words per key: relator a, relator b, na | nb << 16
constexpr int kThreads = 256;
constexpr uint32_t kCand = 4096;   // candidates per pass
constexpr unsigned long long kFree = ~0ull;

struct Rel {
    uint64_t w[BN];
    int n;
};

__device__ __forceinline__ uint64_t rev2_64(uint64_t w) {
    const uint64_t r = __builtin_bitreverse64(w);
    return ((r & 0x5555555555555555ull) << 1) | ((r >> 1) & 0x5555555555555555ull);
}

// out = in << (2 * letters), letters in [0, 192]
__device__ __forceinline__ void shl(const uint64_t* in, int letters, uint64_t* out) {
    const int bits = 2 * letters, ws = bits >> 6, bs = bits & 63;
    uint64_t t[BN];
#pragma unroll
    for (int i = 0; i < BN; i++) {  // word shift (select chain: runtime ws)
        uint64_t v = 0;
#pragma unroll
        for (int k = 0; k < BN; k++)
            if (i - k == ws) v = in[k];
        t[i] = v;
    }
#pragma unroll
    for (int i = 0; i < BN; i++) {
        const uint64_t lo = i > 0 ? t[i - 1] : 0;
        out[i] = bs ? (t[i] << bs) | (lo >> (64 - bs)) : t[i];
    }
}

__device__ __forceinline__ void shr(const uint64_t* in, int letters, uint64_t* out) {
    const int bits = 2 * letters, ws = bits >> 6, bs = bits & 63;
    uint64_t t[BN];
#pragma unroll
    for (int i = 0; i < BN; i++) {
        uint64_t v = 0;
#pragma unroll
        for (int k = 0; k < BN; k++)
            if (k - i == ws) v = in[k];
        t[i] = v;
    }
#pragma unroll
    for (int i = 0; i < BN; i++) {
        const uint64_t hi = i + 1 < BN ? t[i + 1] : 0;
        out[i] = bs ? (t[i] >> bs) | (hi << (64 - bs)) : t[i];
    }
}

// word i of the mask of the lowest `letters` letters
__device__ __forceinline__ uint64_t mask_word(int letters, int i) {
    const int bits = 2 * letters - 64 * i;
    return bits <= 0 ? 0ull : (bits >= 64 ? ~0ull : ((1ull << bits) - 1ull));
}

__device__ __forceinline__ int letter(const uint64_t* w, int k) {
    uint64_t v = 0;
#pragma unroll
    for (int i = 0; i < BN; i++)
        if ((k >> 5) == i) v = w[i];
    return (int)((v >> (2 * (k & 31))) & 3u);
}

// first letter position where a and b differ (192 if none)
__device__ __forceinline__ int first_diff(const uint64_t* a, const uint64_t* b) {
    int pos = kMaxLetters;
#pragma unroll
    for (int i = BN - 1; i >= 0; i--) {
        const uint64_t t = a[i] ^ b[i];
        if (t) pos = 32 * i + (__builtin_ctzll(t) >> 1);
    }
    return pos;
}

__device__ __forceinline__ void inverse(const Rel& r, Rel& out) {  // inv0_, AC_UTILS_no_hash.cpp:103-109
    uint64_t t[BN];
#pragma unroll
    for (int i = 0; i < BN; i++) t[i] = rev2_64(r.w[BN - 1 - i]);
    shr(t, kMaxLetters - r.n, out.w);
#pragma unroll
    for (int i = 0; i < BN; i++) out.w[i] ^= mask_word(r.n, i);
    out.n = r.n;
}

// concat_ (:111-119): a b freely reduced; both inputs reduced, so only the junction cancels.  false if too long.
__device__ __forceinline__ bool concat(const Rel& a, const Rel& b, Rel& out) {
    Rel ia;
    inverse(a, ia);
    int acc = first_diff(ia.w, b.w);
    const int m = a.n < b.n ? a.n : b.n;
    acc = acc < m ? acc : m;
    const int nn = a.n + b.n - 2 * acc;
    if (nn > kMaxLetters) return false;
    uint64_t t[BN], u[BN];
    shr(b.w, acc, t);
    shl(t, a.n - acc, u);
#pragma unroll
    for (int i = 0; i < BN; i++) out.w[i] = (a.w[i] & mask_word(a.n - acc, i)) | u[i];
    out.n = nn;
    return true;
}

// conj0_ (:93-101): x^-1 r x freely reduced, x given by its code; r reduced, so only the two ends can cancel
__device__ __forceinline__ bool conjugate(const Rel& r, int xcode, Rel& out) {
    const int g = xcode ^ 3;  // the letter put in front (x^-1); x goes behind
    if (r.n == 0) {
#pragma unroll
        for (int i = 0; i < BN; i++) out.w[i] = 0;
        out.n = 0;
        return true;
    }
    const int sc = letter(r.w, 0) == xcode, ec = letter(r.w, r.n - 1) == g;
    const int nb = r.n - sc - ec, nn = r.n + 2 - 2 * (sc + ec);
    if (nn > kMaxLetters) return false;
    uint64_t t[BN], u[BN];
    shr(r.w, sc, t);
#pragma unroll
    for (int i = 0; i < BN; i++) t[i] &= mask_word(nb, i);
    if (!sc) {
        shl(t, 1, u);
        u[0] |= (uint64_t)g;
    } else {
#pragma unroll
        for (int i = 0; i < BN; i++) u[i] = t[i];
    }
    if (!ec) {
        uint64_t one[BN] = {(uint64_t)xcode, 0, 0, 0, 0, 0}, sh[BN];
        shl(one, nn - 1, sh);
#pragma unroll
        for (int i = 0; i < BN; i++) u[i] |= sh[i];
    }
#pragma unroll
    for (int i = 0; i < BN; i++) out.w[i] = u[i];
    out.n = nn;
    return true;
}

__device__ __forceinline__ bool rel_less(const Rel& l, const Rel& r) {  // operator<, :19-38 (codes keep the letter order)
    if (l.n != r.n) return l.n < r.n;
    const int p = first_diff(l.w, r.w);
    return p < l.n && letter(l.w, p) < letter(r.w, p);
}

// key = sort_(x, y): (first, second) with first < second, else swapped (:121-137)
__device__ __forceinline__ void make_key(const Rel& x, const Rel& y, uint64_t* key) {
    const bool keep = rel_less(x, y);
    const Rel& a = keep ? x : y;
    const Rel& b = keep ? y : x;
#pragma unroll
    for (int i = 0; i < BN; i++) {
        key[i] = a.w[i];
        key[BN + i] = b.w[i];
    }
    key[2 * BN] = (uint64_t)a.n | ((uint64_t)b.n << 16);
}

__device__ __forceinline__ void key_rels(const uint64_t* key, Rel& a, Rel& b) {
#pragma unroll
    for (int i = 0; i < BN; i++) {
        a.w[i] = key[i];
        b.w[i] = key[BN + i];
    }
    a.n = (int)(key[2 * BN] & 0xffff);
    b.n = (int)(key[2 * BN] >> 16);
}

// move (:144-211) on the sorted pair (r1, r2); letter codes: a=2, b=3, A=1, B=0.  false when a relator outgrows 192 letters
__device__ __forceinline__ bool apply(const Rel& r1, const Rel& r2, int t, bool classic, uint64_t* key) {
    Rel x, y;
    bool ok = true;
    if (classic) {
        static const int8_t cx[8] = {2, 3, 1, 0, 2, 3, 1, 0};  // conjugators of moves 4..11: a b A B a b A B
        if (t == 0) { ok = concat(r1, r2, x); make_key(x, r2, key); }
        else if (t == 1) { ok = concat(r2, r1, x); make_key(x, r2, key); }
        else if (t == 2) { ok = concat(r1, r2, x); make_key(r1, x, key); }
        else if (t == 3) { ok = concat(r2, r1, x); make_key(r1, x, key); }
        else if (t < 8) { ok = conjugate(r2, cx[t - 4], x); make_key(r1, x, key); }
        else if (t < 12) { ok = conjugate(r1, cx[t - 4], x); make_key(x, r2, key); }
        else if (t == 12) { inverse(r1, x); make_key(x, r2, key); }
        else { inverse(r2, x); make_key(r1, x, key); }
    } else {
        static const int8_t px[8] = {0, 1, 2, 3, 0, 1, 2, 3};  // conjugators of moves 4..11: B A a b B A a b
        if (t == 0) { ok = concat(r1, r2, x); make_key(x, r2, key); }
        else if (t == 1) { ok = concat(r2, r1, x); make_key(r1, x, key); }
        else if (t == 2) { inverse(r2, y); ok = concat(r1, y, x); make_key(x, r2, key); }
        else if (t == 3) { inverse(r1, y); ok = concat(r2, y, x); make_key(r1, x, key); }
        else if (t < 8) { ok = conjugate(r1, px[t - 4], x); make_key(x, r2, key); }
        else { ok = conjugate(r2, px[t - 4], x); make_key(r1, x, key); }
    }
    return ok;
}

__device__ __forceinline__ uint64_t hash_key(const uint64_t* key) {
    uint64_t h = 0x9e3779b97f4a7c15ull;
#pragma unroll
    for (int i = 0; i < KW; i++) {
        h = (h ^ key[i]) * 0xd6e8feb86659fd93ull;
        h ^= h >> 32;
    }
    return h;
}

struct Job {
    uint64_t* tab;       // [tcap][KW + 1]: key words then the stamp (kFree: empty)
    uint64_t* front[2];  // frontier keys, [fcap][KW] each
    uint64_t* cand;      // [kCand][KW] children of the running pass
    uint64_t root[KW];
    uint32_t tmask, fcap;
    int32_t radius, classic;
    // results
    unsigned long long size;
    uint32_t status;  // 0 ok, 1 table full, 2 frontier overflow, 3 relator longer than 192 letters
    uint32_t max_len;
};

__global__ void __launch_bounds__(kThreads) k_ball(Job* jobs) {
    ACX_VGPR_PAD("v87");
    __shared__ Job J;  // this workgroup's job (pointers, capacities, root)
    __shared__ uint32_t s_next, s_status, s_maxlen, s_count;
    const uint32_t tid = threadIdx.x;
    if (tid == 0) J = jobs[blockIdx.x];
    __syncthreads();
    const int M = J.classic ? 14 : 12;
    const uint32_t ESZ = KW + 1;
    if (tid == 0) {
        s_status = 0;
        s_maxlen = 0;
        s_count = 1;
        // the root: frontier 0 and the table
        uint64_t* e = J.tab + (size_t)((uint32_t)hash_key(J.root) & J.tmask) * ESZ;
        for (int i = 0; i < KW; i++) {
            J.front[0][i] = J.root[i];
            e[i] = J.root[i];
        }
        e[KW] = 0;  // stamp of pass 0: committed
    }
    __syncthreads();
    uint32_t nf = 1, pass = 0;
    for (int dist = 0; dist < J.radius && nf > 0; dist++) {
        const uint64_t* cur = J.front[dist & 1];
        uint64_t* nxt = J.front[(dist + 1) & 1];
        const bool keep_next = dist + 1 < J.radius;  // the last level is counted, never expanded
        if (tid == 0) s_next = 0;
        __syncthreads();
        const uint32_t per_pass = kCand / (uint32_t)M;
        for (uint32_t p0 = 0; p0 < nf; p0 += per_pass) {
            pass++;
            const uint32_t np = nf - p0 < per_pass ? nf - p0 : per_pass;
            const uint32_t m = np * (uint32_t)M;
            // ---- expand: candidate c = (parent c / M, move c % M)
            for (uint32_t c = tid; c < m; c += kThreads) {
                const uint32_t p = c / (uint32_t)M, t = c - p * (uint32_t)M;
                uint64_t pk[KW], ck[KW];
#pragma unroll
                for (int i = 0; i < KW; i++) pk[i] = cur[(size_t)(p0 + p) * KW + i];
                Rel r1, r2;
                key_rels(pk, r1, r2);
                if (!apply(r1, r2, (int)t, J.classic != 0, ck)) atomicMax(&s_status, 3u);
#pragma unroll
                for (int i = 0; i < KW; i++) J.cand[(size_t)c * KW + i] = ck[i];
                const uint32_t la = (uint32_t)(ck[2 * BN] & 0xffff), lb = (uint32_t)(ck[2 * BN] >> 16);
                if ((la > lb ? la : lb) > s_maxlen) atomicMax(&s_maxlen, la > lb ? la : lb);
            }
            __syncthreads();
            // ---- insert: exact dedup; a new key is counted and (unless this is the last level) joins the next frontier
            for (uint32_t c = tid; c < m && s_status == 0; c += kThreads) {
                uint64_t ck[KW];
#pragma unroll
                for (int i = 0; i < KW; i++) ck[i] = J.cand[(size_t)c * KW + i];
                const unsigned long long me = ((unsigned long long)pass << 32) | c;
                uint32_t h = (uint32_t)hash_key(ck) & J.tmask;
                bool fresh = false;
                for (uint32_t probes = 0;; probes++) {
                    if (probes > J.tmask) {
                        atomicMax(&s_status, 1u);
                        break;
                    }
                    uint64_t* e = J.tab + (size_t)h * ESZ;
                    unsigned long long st = e[KW];
                    if (st == kFree) {
                        st = atomicCAS((unsigned long long*)&e[KW], kFree, me);
                        if (st == kFree) {  // claimed: the key moves in for the later passes
#pragma unroll
                            for (int i = 0; i < KW; i++) e[i] = ck[i];
                            fresh = true;
                            break;
                        }
                    }
                    const uint64_t* other = (uint32_t)(st >> 32) == pass ? J.cand + (size_t)(uint32_t)st * KW : e;
                    bool same = true;
#pragma unroll
                    for (int i = 0; i < KW; i++) same = same && other[i] == ck[i];
                    if (same) break;
                    h = (h + 1) & J.tmask;
                }
                if (fresh) {
                    atomicAdd(&s_count, 1u);
                    if (keep_next) {
                        const uint32_t pos = atomicAdd(&s_next, 1u);
                        if (pos < J.fcap) {
#pragma unroll
                            for (int i = 0; i < KW; i++) nxt[(size_t)pos * KW + i] = ck[i];
                        } else {
                            atomicMax(&s_status, 2u);
                        }
                    }
                }
            }
            __syncthreads();
            if (s_status) break;
        }
        if (s_status) break;
        nf = s_next;
        __syncthreads();
    }
    if (tid == 0) {
        jobs[blockIdx.x].size = s_count;
        jobs[blockIdx.x].status = s_status;
        jobs[blockIdx.x].max_len = s_maxlen;
    }
}

// code of a reference letter (+-1, +-2): -2 -> 0, -1 -> 1, +1 -> 2, +2 -> 3
static inline int code_of(int a) { return a < 0 ? a + 2 : a + 1; }

static bool pack_rel(const int8_t* r, int L, uint64_t* w, int* n) {
    int len = 0;
    for (int i = 0; i < BN; i++) w[i] = 0;
    for (int k = 0; k < L; k++) {
        const int a = r[k];
        if (a == 0) continue;  // the reference program drops zeros wherever they are (neibourhoods.cpp:76-85)
        if (a < -2 || a > 2 || len >= kMaxLetters) return false;
        w[len >> 5] |= (uint64_t)code_of(a) << (2 * (len & 31));
        len++;
    }
    *n = len;
    return true;
}

static bool host_less(const uint64_t* a, int na, const uint64_t* b, int nb) {
    if (na != nb) return na < nb;
    for (int k = 0; k < na; k++) {
        const int ca = (int)((a[k >> 5] >> (2 * (k & 31))) & 3), cb = (int)((b[k >> 5] >> (2 * (k & 31))) & 3);
        if (ca != cb) return ca < cb;
    }
    return false;
}

}  // namespace ball
}  // namespace acx

using namespace acx;
using namespace acx::ball;

extern "C" int acx_ball_sizes(const int8_t* h_presentations, int64_t n, int L, int radius, int classic, int64_t* h_sizes, int32_t* h_max_len) {
    if (!have_device()) return ACX_E_NODEVICE;
    if (n < 0 || L < 1 || radius < 0 || !h_presentations || !h_sizes) return fail(ACX_E_INVAL, "acx_ball_sizes: bad argument");
    if (n == 0) return ACX_OK;
    std::vector<Job> jobs((size_t)n);
    for (int64_t k = 0; k < n; k++) {
        uint64_t a[BN], b[BN];
        int na = 0, nb = 0;
        if (!pack_rel(h_presentations + k * 2 * L, L, a, &na) || !pack_rel(h_presentations + k * 2 * L + L, L, b, &nb))
            return fail(ACX_E_ROWERR, "acx_ball_sizes: presentation %lld is not a word pair over {+-1,+-2} of at most %d letters", (long long)k, kMaxLetters);
        Job& J = jobs[k];
        ::memset((void*)&J, 0, sizeof(J));
        const bool keep = host_less(a, na, b, nb);
        for (int i = 0; i < BN; i++) {
            J.root[i] = keep ? a[i] : b[i];
            J.root[BN + i] = keep ? b[i] : a[i];
        }
        J.root[2 * BN] = (uint64_t)(keep ? na : nb) | ((uint64_t)(keep ? nb : na) << 16);
        J.radius = radius;
        J.classic = classic ? 1 : 0;
    }
    // capacities grow for the jobs that overflow; groups bound the device memory in use
    std::vector<uint32_t> tcap((size_t)n, 1u << 19), fcap((size_t)n, 1u << 16);
    std::vector<int64_t> todo((size_t)n);
    for (int64_t k = 0; k < n; k++) todo[k] = k;
    hipStream_t st = nullptr;
    ACX_HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    int rc = ACX_OK;
    while (!todo.empty() && rc == ACX_OK) {
        std::vector<int64_t> again;
        size_t pos = 0;
        while (pos < todo.size() && rc == ACX_OK) {
            // a group: as many jobs as fit 12 GB
            size_t end = pos;
            uint64_t bytes = 0;
            auto job_bytes = [&](int64_t k) {
                return (uint64_t)tcap[k] * (KW + 1) * 8 + 2ull * fcap[k] * KW * 8 + (uint64_t)kCand * KW * 8 + 1024;
            };
            while (end < todo.size() && end - pos < 1024 && (end == pos || bytes + job_bytes(todo[end]) <= (12ull << 30))) bytes += job_bytes(todo[end++]);
            void* base = nullptr;
            void* djobs = nullptr;
            if (hipMalloc(&base, bytes) != hipSuccess || hipMalloc(&djobs, (end - pos) * sizeof(Job)) != hipSuccess) {
                if (base) (void)hipFree(base);
                rc = fail(ACX_E_NOMEM, "acx_ball_sizes: hipMalloc(%llu) failed", (unsigned long long)bytes);
                break;
            }
            std::vector<Job> grp;
            uint8_t* q = (uint8_t*)base;
            for (size_t i = pos; i < end; i++) {
                const int64_t k = todo[i];
                Job J = jobs[k];
                J.tab = (uint64_t*)q;
                const uint64_t tb = (uint64_t)tcap[k] * (KW + 1) * 8;
                (void)hipMemsetAsync(q, 0xff, tb, st);  // stamps (and keys) all ones: every entry free
                q += tb;
                J.front[0] = (uint64_t*)q;
                q += (uint64_t)fcap[k] * KW * 8;
                J.front[1] = (uint64_t*)q;
                q += (uint64_t)fcap[k] * KW * 8;
                J.cand = (uint64_t*)q;
                q += (uint64_t)kCand * KW * 8 + 1024;
                J.tmask = tcap[k] - 1;
                J.fcap = fcap[k];
                grp.push_back(J);
            }
            hipError_t e = hipMemcpyAsync(djobs, grp.data(), grp.size() * sizeof(Job), hipMemcpyHostToDevice, st);
            if (e == hipSuccess) {
                hipLaunchKernelGGL(k_ball, dim3((unsigned)grp.size()), dim3(kThreads), 0, st, (Job*)djobs);
                e = hipGetLastError();
            }
            if (e == hipSuccess) e = hipMemcpyAsync(grp.data(), djobs, grp.size() * sizeof(Job), hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
            (void)hipFree(base);
            (void)hipFree(djobs);
            if (e != hipSuccess) {
                rc = fail(ACX_E_NODEVICE, "acx_ball_sizes: %s", hipGetErrorString(e));
                break;
            }
            for (size_t i = pos; i < end; i++) {
                const int64_t k = todo[i];
                const Job& J = grp[i - pos];
                if (J.status == 3) {
                    rc = fail(ACX_E_CAPACITY, "acx_ball_sizes: a relator outgrew %d letters (presentation %lld)", kMaxLetters, (long long)k);
                    break;
                }
                if (J.status == 1 || J.status == 2) {
                    if (J.status == 1) tcap[k] *= 4;
                    else fcap[k] *= 4;
                    if (tcap[k] > (1u << 27) || fcap[k] > (1u << 26)) {
                        rc = fail(ACX_E_CAPACITY, "acx_ball_sizes: neighbourhood of presentation %lld does not fit", (long long)k);
                        break;
                    }
                    again.push_back(k);
                    continue;
                }
                h_sizes[k] = (int64_t)J.size;
                if (h_max_len) h_max_len[k] = (int32_t)J.max_len;
            }
            pos = end;
        }
        todo.swap(again);
    }
    (void)hipStreamDestroy(st);
    return rc;
}
