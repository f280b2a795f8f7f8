// microbench_afterlong.hip -- how long does a few-byte fill take to complete right after a kernel that ran for ~130 ms?
// (round 6: from the third greedy sweep of a process on, the first device operation of the next call completes 10-25 ms late.)
//   hipcc --offload-arch=gfx950 -O3 -o tools/_build/mbafter tools/microbench_afterlong.hip && tools/_build/mbafter [kernel ms] [host pause ms] [workgroups]
#include <hip/hip_runtime.h>
#include <chrono>
#include <stdio.h>
#include <stdlib.h>
#include <thread>

__global__ void k_spin(unsigned long long ticks, unsigned* sink) {  // ticks of the 100 MHz counter
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned x = threadIdx.x;
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {
        x = x * 1664525u + 1013904223u;
        __builtin_amdgcn_s_sleep(8);
    }
    if (x == 12345u) sink[0] = x;
}

int main(int argc, char** argv) {
    const double kernel_ms = argc > 1 ? atof(argv[1]) : 130.0, pause_ms = argc > 2 ? atof(argv[2]) : 5.0;
    const int wgs = argc > 3 ? atoi(argv[3]) : 512;
    hipStream_t a, b;
    hipStreamCreateWithFlags(&a, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
    unsigned* d;
    hipMalloc(&d, 4096);
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](auto t0, auto t1) { return std::chrono::duration<double, std::milli>(t1 - t0).count(); };
    for (int rep = 0; rep < 8; rep++) {
        const auto t0 = now();
        hipLaunchKernelGGL(k_spin, dim3(wgs), dim3(512), 0, a, (unsigned long long)(kernel_ms * 1e5), d);
        hipStreamSynchronize(a);
        const auto t1 = now();
        std::this_thread::sleep_for(std::chrono::microseconds((long)(pause_ms * 1e3)));
        const auto t2 = now();
        hipMemsetAsync(d, 0xff, 64, b);
        hipStreamSynchronize(b);
        const auto t3 = now();
        printf("rep %d: kernel + wait %.2f ms (asked %.0f), then a 64-byte fill on another stream after a %.1f ms pause: %.3f ms\n", rep, ms(t0, t1), kernel_ms, pause_ms, ms(t2, t3));
    }
    return 0;
}
