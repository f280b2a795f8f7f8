// microbench_table.hip -- what random table traffic costs on MI355X (the floor under k_insert_tab).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mbt tools/microbench_table.hip && /tmp/mbt [log2_accesses]
// One lane = one access at a pseudo-random entry of a table of `bytes`; rates in G accesses/s.
//   read16/32/64   plain loads of 16 / 32 / 64 B per lane
//   cas_agent      one 64-bit device-scope CAS per lane (what a claim of an empty entry costs today)
//   cas_wg_xcd     64-bit workgroup-scope CAS (executes in the XCD's L2), lanes of an XCD confined to that XCD's eighth of the table
//   st16/st32      plain stores
//   probe          today's insert pattern: 32-B read, then for 30 % of the lanes CAS + 16-B store into the same entry
//   probe16        16-B entries: 16-B read, 30 %: CAS on the first word + 8-B store of the second
//   probe_xcd      as probe16, XCD-partitioned with L2-scope atomics and sc1 loads
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x)                                                                     \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                              \
        }                                                                         \
    } while (0)

__device__ __forceinline__ uint64_t mix64(uint64_t x) {
    x ^= x >> 32;
    x *= 0xd6e8feb86659fd93ull;
    x ^= x >> 32;
    x *= 0xd6e8feb86659fd93ull;
    x ^= x >> 32;
    return x;
}
__device__ __forceinline__ uint32_t xcc_id() {
    uint32_t v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 7u;
}

struct u4 {
    uint64_t a, b;
};
struct u8x {
    uint64_t a, b, c, d;
};

template <int BYTES> __global__ void __launch_bounds__(256) k_read(const uint8_t* __restrict__ tab, uint64_t mask, uint64_t n, uint64_t seed, uint64_t* sink) {
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const uint64_t h = mix64(t + seed) & mask;
    uint64_t acc = 0;
    const uint4* p = (const uint4*)(tab + h * BYTES);
#pragma unroll
    for (int i = 0; i < BYTES / 16; i++) {
        const uint4 v = p[i];
        acc ^= ((uint64_t)v.x << 32 | v.y) + ((uint64_t)v.z << 32 | v.w);
    }
    if (acc == 0x1234567887654321ull) *sink = acc;
}

template <int BYTES> __global__ void __launch_bounds__(256) k_store(uint8_t* __restrict__ tab, uint64_t mask, uint64_t n, uint64_t seed) {
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const uint64_t h = mix64(t + seed) & mask;
    uint4* p = (uint4*)(tab + h * BYTES);
    const uint4 v = make_uint4((uint32_t)t, 1, 2, 3);
#pragma unroll
    for (int i = 0; i < BYTES / 16; i++) p[i] = v;
}

__global__ void __launch_bounds__(256) k_cas_agent(unsigned long long* __restrict__ tab, uint64_t mask, uint64_t n, uint64_t seed, int stride8) {
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const uint64_t h = mix64(t + seed) & mask;
    atomicCAS(&tab[h * stride8], 0ull, t + 1);
}

// persistent blocks: each takes the lanes of ITS XCD's share (lane ids are dealt to XCDs by a per-XCD ticket)
__global__ void __launch_bounds__(256) k_cas_wg_xcd(unsigned long long* __restrict__ tab, uint64_t mask, uint64_t n, uint64_t seed, int stride8, uint32_t* tickets) {
    __shared__ uint32_t s_t;
    const uint32_t x = xcc_id();
    const uint64_t per = n / 8, region = (mask + 1) / 8;
    for (;;) {
        if (threadIdx.x == 0) s_t = atomicAdd(&tickets[x * 32], 1u);
        __syncthreads();
        const uint64_t base = (uint64_t)s_t * 256;
        __syncthreads();
        if (base >= per) break;
        const uint64_t t = base + threadIdx.x;
        const uint64_t h = (mix64(t + seed + x * per) & (region - 1)) + x * region;
        unsigned long long exp = 0ull;
        __hip_atomic_compare_exchange_strong(&tab[h * stride8], &exp, (unsigned long long)(t + 1), __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

// today's pattern on 32-B entries {k0, k1, stamp, pad}
__global__ void __launch_bounds__(256) k_probe32(uint8_t* __restrict__ tab, uint64_t mask, uint64_t n, uint64_t seed, uint64_t* sink) {
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const uint64_t r = mix64(t + seed);
    const uint64_t h = r & mask;
    u8x* e = (u8x*)(tab + h * 32);
    const uint64_t k0 = e->a, k1 = e->b;
    const unsigned long long st = *(volatile unsigned long long*)&e->c;
    if ((r >> 40) % 10 < 3) {
        const unsigned long long old = atomicCAS((unsigned long long*)&e->c, st, (unsigned long long)t);
        if (old == st) {
            e->a = r;
            e->b = t;
        }
    } else if (k0 + k1 == 0x1234567887654321ull) {
        *sink = k0;
    }
}

__global__ void __launch_bounds__(256) k_probe16(uint8_t* __restrict__ tab, uint64_t mask, uint64_t n, uint64_t seed, uint64_t* sink) {
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const uint64_t r = mix64(t + seed);
    const uint64_t h = r & mask;
    u4* e = (u4*)(tab + h * 16);
    const uint4 v = *(const uint4*)e;
    const uint64_t k0 = (uint64_t)v.x | ((uint64_t)v.y << 32), k1 = (uint64_t)v.z | ((uint64_t)v.w << 32);
    if ((r >> 40) % 10 < 3) {
        const unsigned long long old = atomicCAS((unsigned long long*)&e->a, k0, (unsigned long long)r | 1ull);
        if (old == k0) e->b = t;
    } else if (k0 + k1 == 0x1234567887654321ull) {
        *sink = k0;
    }
}

// the stamp table: 8-B slots, `pct` % of the lanes claim with a CAS, nothing else is stored
__global__ void __launch_bounds__(256) k_probe8(unsigned long long* __restrict__ tab, uint64_t mask, uint64_t n, uint64_t seed, uint64_t* sink, uint32_t pct) {
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const uint64_t r = mix64(t + seed);
    const uint64_t h = r & mask;
    const unsigned long long st = tab[h];
    if ((r >> 40) % 100 < pct) {
        atomicCAS(&tab[h], st, (unsigned long long)r);
    } else if (st == 0x1234567887654321ull) {
        *sink = st;
    }
}

__global__ void __launch_bounds__(256) k_read8(const unsigned long long* __restrict__ tab, uint64_t mask, uint64_t n, uint64_t seed, uint64_t* sink) {
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const unsigned long long st = tab[mix64(t + seed) & mask];
    if (st == 0x1234567887654321ull) *sink = st;
}

__global__ void __launch_bounds__(256) k_probe16_xcd(uint8_t* __restrict__ tab, uint64_t mask, uint64_t n, uint64_t seed, uint64_t* sink, uint32_t* tickets) {
    __shared__ uint32_t s_t;
    const uint32_t x = xcc_id();
    const uint64_t per = n / 8, region = (mask + 1) / 8;
    for (;;) {
        if (threadIdx.x == 0) s_t = atomicAdd(&tickets[x * 32], 1u);
        __syncthreads();
        const uint64_t base = (uint64_t)s_t * 256;
        __syncthreads();
        if (base >= per) break;
        const uint64_t t = base + threadIdx.x;
        const uint64_t r = mix64(t + seed + x * per);
        const uint64_t h = (r & (region - 1)) + x * region;
        u4* e = (u4*)(tab + h * 16);
        const uint64_t k0 = __hip_atomic_load(&e->a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint64_t k1 = __hip_atomic_load(&e->b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((r >> 40) % 10 < 3) {
            unsigned long long exp = k0;
            if (__hip_atomic_compare_exchange_strong((unsigned long long*)&e->a, &exp, (unsigned long long)r | 1ull, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_WORKGROUP))
                e->b = t;
        } else if (k0 + k1 == 0x1234567887654321ull) {
            *sink = k0;
        }
    }
}

int main(int argc, char** argv) {
    const int lg = argc > 1 ? atoi(argv[1]) : 27;
    const uint64_t n = 1ull << lg;
    uint64_t* sink;
    uint32_t* tickets;
    CK(hipMalloc(&sink, 8));
    CK(hipMalloc(&tickets, 8 * 32 * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const size_t sizes[] = {64ull << 20, 256ull << 20, 2ull << 30, 8ull << 30};
    uint8_t* tab;
    CK(hipMalloc(&tab, sizes[3]));
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    for (size_t bytes : sizes) {
        CK(hipMemset(tab, 0, bytes));
        auto run = [&](const char* name, auto&& launch, double bytes_per) {
            float best = 1e30f;
            for (int rep = 0; rep < 3; rep++) {
                CK(hipMemset(tickets, 0, 8 * 32 * 4));
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0));
                launch((uint64_t)rep * n + 12345);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                CK(hipGetLastError());
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            printf("table %6.0f MB  %-14s %8.3f ms  %7.2f G/s  %7.2f TB/s(payload)\n", bytes / 1048576.0, name, best, n / best / 1e6, n * bytes_per / best / 1e9);
            fflush(stdout);
        };
        run("read8", [&](uint64_t s) { hipLaunchKernelGGL(k_read8, grid, block, 0, 0, (const unsigned long long*)tab, bytes / 8 - 1, n, s, sink); }, 8);
        run("probe8_30", [&](uint64_t s) { hipLaunchKernelGGL(k_probe8, grid, block, 0, 0, (unsigned long long*)tab, bytes / 8 - 1, n, s, sink, 30u); }, 8);
        run("probe8_55", [&](uint64_t s) { hipLaunchKernelGGL(k_probe8, grid, block, 0, 0, (unsigned long long*)tab, bytes / 8 - 1, n, s, sink, 55u); }, 8);
        run("read16", [&](uint64_t s) { hipLaunchKernelGGL(k_read<16>, grid, block, 0, 0, tab, bytes / 16 - 1, n, s, sink); }, 16);
        run("read32", [&](uint64_t s) { hipLaunchKernelGGL(k_read<32>, grid, block, 0, 0, tab, bytes / 32 - 1, n, s, sink); }, 32);
        run("read64", [&](uint64_t s) { hipLaunchKernelGGL(k_read<64>, grid, block, 0, 0, tab, bytes / 64 - 1, n, s, sink); }, 64);
        run("read128", [&](uint64_t s) { hipLaunchKernelGGL(k_read<128>, grid, block, 0, 0, tab, bytes / 128 - 1, n, s, sink); }, 128);
        run("st16", [&](uint64_t s) { hipLaunchKernelGGL(k_store<16>, grid, block, 0, 0, tab, bytes / 16 - 1, n, s); }, 16);
        run("st32", [&](uint64_t s) { hipLaunchKernelGGL(k_store<32>, grid, block, 0, 0, tab, bytes / 32 - 1, n, s); }, 32);
        CK(hipMemset(tab, 0, bytes));
        run("cas_agent/32", [&](uint64_t s) { hipLaunchKernelGGL(k_cas_agent, grid, block, 0, 0, (unsigned long long*)tab, bytes / 32 - 1, n, s, 4); }, 8);
        CK(hipMemset(tab, 0, bytes));
        run("cas_wg_xcd/32", [&](uint64_t s) { hipLaunchKernelGGL(k_cas_wg_xcd, dim3(2048), block, 0, 0, (unsigned long long*)tab, bytes / 32 - 1, n, s, 4, tickets); }, 8);
        CK(hipMemset(tab, 0, bytes));
        run("probe32", [&](uint64_t s) { hipLaunchKernelGGL(k_probe32, grid, block, 0, 0, tab, bytes / 32 - 1, n, s, sink); }, 32);
        CK(hipMemset(tab, 0, bytes));
        run("probe16", [&](uint64_t s) { hipLaunchKernelGGL(k_probe16, grid, block, 0, 0, tab, bytes / 16 - 1, n, s, sink); }, 16);
        CK(hipMemset(tab, 0, bytes));
        run("probe16_xcd", [&](uint64_t s) { hipLaunchKernelGGL(k_probe16_xcd, dim3(2048), block, 0, 0, tab, bytes / 16 - 1, n, s, sink, tickets); }, 16);
    }
    return 0;
}
