// node_write_bench.hip -- what the node arrays of a search cost to WRITE: n consecutive ids, a lane per id, into
//   A: six arrays of 8, 8, 4, 4, 1, 1 bytes (k0, k1, parent, depth, act, tlen -- today's layout)
//   B: five arrays of 8, 8, 4, 4, 2 bytes (act and tlen in one halfword)
//   C: three arrays of 8, 8, 8 bytes (parent | depth-or-act/tlen packed in one word)
//   D: one array of 32-byte records
// hipcc --offload-arch=gfx950 -O3 -o /tmp/nwb tools/scratch/node_write_bench.hip && /tmp/nwb
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
template <int V> __global__ void __launch_bounds__(256) k(uint64_t* a, uint64_t* b, uint32_t* c, uint32_t* d, uint8_t* e, uint8_t* f, uint64_t n, uint64_t base) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint64_t id = base + i, x = id * 0x9E3779B97F4A7C15ull;
    if (V == 0) { a[id] = x; b[id] = x >> 7; c[id] = (uint32_t)x; d[id] = (uint32_t)(x >> 9); e[id] = (uint8_t)x; f[id] = (uint8_t)(x >> 3); }
    if (V == 1) { a[id] = x; b[id] = x >> 7; c[id] = (uint32_t)x; d[id] = (uint32_t)(x >> 9); ((uint16_t*)e)[id] = (uint16_t)x; }
    if (V == 2) { a[id] = x; b[id] = x >> 7; ((uint64_t*)c)[id] = x >> 5; }
    if (V == 3) { ulonglong4 r; r.x = x; r.y = x >> 7; r.z = x >> 5; r.w = x >> 3; ((ulonglong4*)a)[id] = r; }
}
int main() {
    const uint64_t n = 1ull << 27;  // 1.3e8 nodes
    uint64_t *a, *b; uint32_t *c, *d; uint8_t *e, *f;
    CK(hipMalloc(&a, n * 32 + 4096)); CK(hipMalloc(&b, n * 8 + 4096)); CK(hipMalloc(&c, n * 8 + 4096)); CK(hipMalloc(&d, n * 4 + 4096)); CK(hipMalloc(&e, n * 2 + 4096)); CK(hipMalloc(&f, n + 4096));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char* names[4] = {"A six arrays 8/8/4/4/1/1", "B five arrays 8/8/4/4/2", "C three arrays 8/8/8", "D one array of 32-byte records"};
    const double bytes[4] = {26, 26, 24, 32};
    for (int rep = 0; rep < 2; rep++)
        for (int v = 0; v < 4; v++) {
            CK(hipEventRecord(e0));
            const dim3 grid((unsigned)(n / 256));
            if (v == 0) hipLaunchKernelGGL(k<0>, grid, dim3(256), 0, 0, a, b, c, d, e, f, n, 3ull);
            if (v == 1) hipLaunchKernelGGL(k<1>, grid, dim3(256), 0, 0, a, b, c, d, e, f, n, 3ull);
            if (v == 2) hipLaunchKernelGGL(k<2>, grid, dim3(256), 0, 0, a, b, c, d, e, f, n, 3ull);
            if (v == 3) hipLaunchKernelGGL(k<3>, grid, dim3(256), 0, 0, a, b, c, d, e, f, n, 3ull);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("%-34s %7.3f ms  %6.2f TB/s  (%.2f ns per 1e3 nodes)\n", names[v], ms, bytes[v] * n / ms * 1e-9, ms * 1e6 / (n / 1e3));
        }
    return 0;
}
