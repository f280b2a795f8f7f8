#include "acx_frontier.h"
#include <set>
#include <stdio.h>
using namespace acx;
namespace acx { int fail(int code, const char* fmt, ...) { return code; } }
typedef uint64_t W;
static uint64_t fnv(const void* p, size_t n) { const uint8_t* b = (const uint8_t*)p; uint64_t h = 1469598103934665603ull; for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; } return h; }
struct Args { SearchDev<W> d; const uint32_t* plist; uint32_t pbegin, np; };
int main(int argc, char** argv) {
    int n = 1 << 20, L = 25, cyc = 0;
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
    Args A; memset(&A, 0, sizeof(A));
    SearchDev<W>& d = A.d;
    size_t m = 12 * (size_t)n;
    hipMalloc(&d.k0, n * 8); hipMalloc(&d.k1, n * 8); hipMalloc(&d.ck0, m * 8); hipMalloc(&d.ck1, m * 8); hipMalloc(&d.clen, m); hipMalloc(&d.cknown, m); hipMalloc(&d.cslot, m * 4);
    unsigned long long* sc; hipMalloc(&sc, 64); hipMemset(sc, 0xff, 64);
    d.solved_tag = sc; d.shorter_tag = sc + 1; d.err_tag = sc + 2; d.min_len = (uint32_t*)(sc + 3); d.err = (uint32_t*)(sc + 4);
    d.L = L; d.cyclical = cyc; A.np = n;
    hipMemcpy(d.k0, k0.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(d.k1, k1.data(), n * 8, hipMemcpyHostToDevice);
    for (int i = 1; i < argc; i++) {
        hipModule_t mod; hipFunction_t fn;
        if (hipModuleLoad(&mod, argv[i]) != hipSuccess || hipModuleGetFunction(&fn, mod, "_ZN3acx8k_expandImEEvNS_9SearchDevIT_EEPKjjj") != hipSuccess) { printf("load failed %s\n", argv[i]); continue; }
        for (int rep = 0; rep < 3; rep++) {
            const unsigned lds = rep == 0 ? 0 : rep == 1 ? 65536 : 150000;
            hipMemset(d.ck0, 0, m * 8); hipMemset(d.ck1, 0, m * 8); hipMemset(sc, 0xff, 64);
            size_t sz = sizeof(A);
            void* cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &A, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
            hipError_t e = hipModuleLaunchKernel(fn, (unsigned)((m + 255) / 256), 1, 1, 256, 1, 1, lds, 0, nullptr, cfg);
            hipDeviceSynchronize();
            std::vector<W> h0(m), h1(m); unsigned long long hs[8];
            hipMemcpy(h0.data(), d.ck0, m * 8, hipMemcpyDeviceToHost); hipMemcpy(h1.data(), d.ck1, m * 8, hipMemcpyDeviceToHost); hipMemcpy(hs, sc, 64, hipMemcpyDeviceToHost);
            if (strstr(argv[i], "dbg")) {
                int shown = 0; size_t bad = 0, tot = 0;
                for (size_t t = 0; t < m; t++) {
                    int a = (int)(t % 12); size_t p = t / 12;
                    if (a >= 4) continue;
                    const bool i1 = ((a + 1) & 1) != 0;
                    W wj = i1 ? keyops<W>::word(k0[p]) : keyops<W>::word(k1[p]);
                    int nj = i1 ? keyops<W>::len(k0[p]) : keyops<W>::len(k1[p]);
                    int ni = i1 ? keyops<W>::len(k1[p]) : keyops<W>::len(k0[p]);
                    W want = inv<W, true>(wj, nj);
                    tot++;
                    uint32_t meta = (uint32_t)h1[t], wjlo = (uint32_t)(h1[t] >> 32);
                    if (h0[t] != want || (meta & 255) != (uint32_t)nj || ((meta >> 8) & 255) != (uint32_t)ni || wjlo != (uint32_t)wj) {
                        bad++;
                        if (shown++ < 8) printf("  t=%zu a=%d wj=%016llx nj=%d ni=%d: inv want %016llx got %016llx, seen nj=%u ni=%u wjlo=%08x\n", t, a, (unsigned long long)wj, nj, ni, (unsigned long long)want, (unsigned long long)h0[t], meta & 255, (meta >> 8) & 255, wjlo);
                    }
                }
                printf("  dbg: %zu of %zu concat lanes differ\n", bad, tot);
            }
            printf("lds=%u %s rep=%d launch=%d ck0=%016llx ck1=%016llx err_tag=%llx (want bee47a8ea79b9f05 fd6a521c177424db)\n", lds, argv[i], rep, (int)e, (unsigned long long)fnv(h0.data(), m * 8), (unsigned long long)fnv(h1.data(), m * 8), hs[2]);
        }
    }
    return 0;
}
