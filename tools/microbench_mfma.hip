// microbench_mfma.hip -- how many shader cycles does one v_mfma_f32_32x32x16_bf16 take per SIMD?  (round 6: k_policy_sample's floor,
// tanh and barrier removed, is ~62 cycles per MFMA and SIMD with one wave of two chains or two waves of one chain.)
//   hipcc --offload-arch=gfx950 -O3 -o tools/_build/mbmfma tools/microbench_mfma.hip && tools/_build/mbmfma
// Each wave (one per SIMD) issues N MFMAs over CH independent accumulator chains, operands in registers, nothing else.
// Measured (MI355X): 52 ticks per MFMA with ONE dependent chain, 36 with two, 34 with four (= 1.9 PFLOP/s on the chip's own clock).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) short frag_ab;
typedef __attribute__((ext_vector_type(16))) float frag_cd;

template <int CH> __global__ void __launch_bounds__(256) k_mfma(int n, unsigned long long* out, float* sink) {
    frag_ab a, b;
    for (int j = 0; j < 8; j++) {
        a[j] = (short)(threadIdx.x + j);
        b[j] = (short)(threadIdx.x * 3 + j);
    }
    frag_cd acc[CH];
    for (int c = 0; c < CH; c++)
        for (int j = 0; j < 16; j++) acc[c][j] = 0.0f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int c = 0; c < CH; c++) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[c], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int c = 0; c < CH; c++)
        for (int j = 0; j < 16; j++) s += acc[c][j];
    if (s == 12345.0f) sink[0] = s;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int CH> void run(int waves_per_simd, int n) {
    unsigned long long* d;
    float* sink;
    const int blocks = 256, threads = 256 * waves_per_simd;  // one workgroup per CU, 4 x waves_per_simd waves
    hipMalloc(&d, blocks * 16 * 8);
    hipMalloc(&sink, 4);
    hipLaunchKernelGGL(k_mfma<CH>, dim3(blocks), dim3(threads), 0, 0, n, d, sink);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_mfma<CH>, dim3(blocks), dim3(threads), 0, 0, n, d, sink);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[16];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const double mfmas_per_simd = (double)n * CH * waves_per_simd;
    const double flops = 2.0 * 32 * 32 * 16 * (double)n * CH * (threads / 64) * blocks;
    printf("chains %d, waves/SIMD %d: %.1f ticks per MFMA and SIMD (wave 0: %llu ticks for %d MFMAs), kernel %.1f us -> %.0f TFLOP/s\n", CH, waves_per_simd,
           (double)h[0] / mfmas_per_simd, h[0], n * CH, ms * 1e3, flops / (ms * 1e-3) / 1e12);
    hipFree(d);
    hipFree(sink);
}

int main() {
    const int n = 2000;
    run<1>(1, n);
    run<2>(1, n);
    run<4>(1, n);
    return 0;
}
