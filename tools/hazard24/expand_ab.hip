#include "acx_frontier.h"
#include <set>
#include <tuple>
#include <stdio.h>
using namespace acx;
namespace acx { int fail(int code, const char* fmt, ...) { return code; } }
typedef uint64_t W;
static uint64_t fnv(const void* p, size_t n) { const uint8_t* b = (const uint8_t*)p; uint64_t h = 1469598103934665603ull; for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; } return h; }
int main(int argc, char** argv) {
    int n = 1 << 20, L = 25, cyc = argc > 1 ? atoi(argv[1]) : 0;
    std::vector<W> k0, k1;
    {
        std::set<std::pair<W, W>> seen;
        Pres<W> r0; int a0[7] = {1,1,1,-2,-2,-2,-2}, a1[6] = {1,2,1,-2,-1,-2};
        r0.w0 = r0.w1 = 0; r0.n0 = 7; r0.n1 = 6;
        for (int k = 0; k < 7; k++) r0.w0 |= (W)code_of_letter(a0[k]) << (2 * k);
        for (int k = 0; k < 6; k++) r0.w1 |= (W)code_of_letter(a1[k]) << (2 * k);
        k0.push_back(keyops<W>::make(r0.w0, r0.n0)); k1.push_back(keyops<W>::make(r0.w1, r0.n1)); seen.insert({k0[0], k1[0]});
        for (size_t hd = 0; hd < k0.size() && (int)k0.size() < n; hd++)
            for (int a = 0; a < 12 && (int)k0.size() < n; a++) {
                Pres<W> x; x.w0 = keyops<W>::word(k0[hd]); x.n0 = keyops<W>::len(k0[hd]); x.w1 = keyops<W>::word(k1[hd]); x.n1 = keyops<W>::len(k1[hd]);
                apply_move<W, true>(x, a, L, cyc != 0);
                W c0 = keyops<W>::make(x.w0, x.n0), c1 = keyops<W>::make(x.w1, x.n1);
                if (seen.insert({c0, c1}).second) { k0.push_back(c0); k1.push_back(c1); }
            }
        n = (int)k0.size();
    }
    SearchDev<W> d; memset(&d, 0, sizeof(d));
    size_t m = 12 * (size_t)n;
    hipMalloc(&d.k0, n * 8); hipMalloc(&d.k1, n * 8); hipMalloc(&d.ck0, m * 8); hipMalloc(&d.ck1, m * 8); hipMalloc(&d.clen, m); hipMalloc(&d.cknown, m); hipMalloc(&d.cslot, m * 4);
    unsigned long long* sc; hipMalloc(&sc, 64); hipMemset(sc, 0xff, 64);
    d.solved_tag = sc; d.shorter_tag = sc + 1; d.err_tag = sc + 2; d.min_len = (uint32_t*)(sc + 3); d.err = (uint32_t*)(sc + 4);
    d.L = L; d.cyclical = cyc;
    hipMemcpy(d.k0, k0.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(d.k1, k1.data(), n * 8, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 3; rep++) {
        hipMemset(d.ck0, 0, m * 8); hipMemset(d.ck1, 0, m * 8);
        hipLaunchKernelGGL(k_expand<W>, dim3((m + 255) / 256), dim3(256), 0, 0, d, (const uint32_t*)nullptr, 0u, (uint32_t)n);
        std::vector<W> h0(m), h1(m); std::vector<uint8_t> hl(m), hk(m); unsigned long long hs[8];
        hipMemcpy(h0.data(), d.ck0, m * 8, hipMemcpyDeviceToHost); hipMemcpy(h1.data(), d.ck1, m * 8, hipMemcpyDeviceToHost);
        hipMemcpy(hl.data(), d.clen, m, hipMemcpyDeviceToHost); hipMemcpy(hk.data(), d.cknown, m, hipMemcpyDeviceToHost); hipMemcpy(hs, sc, 64, hipMemcpyDeviceToHost);
        int shown = 0; size_t nbad = 0;
        for (size_t t = 0; t < m; t++) {
            size_t p = t / 12; int a = (int)(t % 12);
            Pres<W> x; x.w0 = keyops<W>::word(k0[p]); x.n0 = keyops<W>::len(k0[p]); x.w1 = keyops<W>::word(k1[p]); x.n1 = keyops<W>::len(k1[p]);
            apply_move<W, true>(x, a, L, cyc != 0);
            W c0 = keyops<W>::make(x.w0, x.n0), c1 = keyops<W>::make(x.w1, x.n1);
            if (c0 != h0[t] || c1 != h1[t]) { nbad++; if (shown++ < 6) printf("  t=%zu lane=%zu a=%d parent (%016llx %016llx) want (%016llx %016llx) got (%016llx %016llx)\n", t, t & 63, a, (unsigned long long)k0[p], (unsigned long long)k1[p], (unsigned long long)c0, (unsigned long long)c1, (unsigned long long)h0[t], (unsigned long long)h1[t]); }
        }
        printf("  mismatches=%zu\n", nbad);
        printf("safe=%d cyc=%d rep=%d n=%d ck0=%016llx ck1=%016llx clen=%016llx known=%016llx err_tag=%llx solved=%llx\n", ACX_SEARCH_SAFE, cyc, rep, n, (unsigned long long)fnv(h0.data(), m * 8),
               (unsigned long long)fnv(h1.data(), m * 8), (unsigned long long)fnv(hl.data(), m), (unsigned long long)fnv(hk.data(), m), hs[2], hs[0]);
    }
    return 0;
}
