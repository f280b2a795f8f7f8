// microbench_mix.hip -- does vector work hide behind the matrix pipe?  Per wave: one dependent chain of v_mfma_f32_32x32x16_bf16 (as a
// block of k_policy_sample has), and between two MFMAs K independent vector instructions (plain f32 FMAs, or the exp / add / rcp / fma
// mix of the kernel's tanh).  One or two waves per SIMD.  Ticks of s_memtime per MFMA and SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o tools/_build/mbmix tools/microbench_mix.hip && tools/_build/mbmix
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) short frag_ab;
typedef __attribute__((ext_vector_type(16))) float frag_cd;

#define FMA(r) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(r) : "v"(k))
#define EXP(r) asm volatile("v_exp_f32 %0, %0" : "+v"(r))
#define RCP(r) asm volatile("v_rcp_f32 %0, %0" : "+v"(r))

template <int K, int KIND, int CH, int AG> __global__ void __launch_bounds__(512) k_mix(int n, unsigned long long* out, float* sink) {
    frag_ab a, b;
    for (int j = 0; j < 8; j++) {
        a[j] = (short)(threadIdx.x + j);
        b[j] = (short)(threadIdx.x * 3 + j);
    }
    frag_cd acc[CH];
    for (int c = 0; c < CH; c++)
        for (int j = 0; j < 16; j++) acc[c][j] = 0.0f;
    float v[16];
    const float k = 0.999f + 1e-9f * threadIdx.x;
    for (int j = 0; j < 16; j++) v[j] = 0.001f * (threadIdx.x + j);
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int c = 0; c < CH; c++) {
            if (AG) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[c]) : "v"(a), "v"(b));  // accumulator in the AccVGPR half
            else acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[c], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < K; j++) {
                if (KIND == 0) FMA(v[j % 16]);
                if (KIND == 1) {  // the tanh's mix: exp, add, rcp, fma per value (here as 4 independent instructions on rotating registers)
                    if (j % 4 == 0) EXP(v[j % 16]);
                    if (j % 4 == 1) FMA(v[j % 16]);
                    if (j % 4 == 2) RCP(v[j % 16]);
                    if (j % 4 == 3) FMA(v[j % 16]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int c = 0; c < CH; c++)
        for (int j = 0; j < 16; j++) s += acc[c][j] + v[j];
    if (s == 12345.0f) sink[0] = s;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int K, int KIND, int CH, int AG = 0> void run(int waves_per_simd) {
    const int n = 1000;
    unsigned long long* d;
    float* sink;
    hipMalloc(&d, 256 * 16 * 8);
    hipMalloc(&sink, 4);
    for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL((k_mix<K, KIND, CH, AG>), dim3(256), dim3(256 * waves_per_simd), 0, 0, n, d, sink);
    hipDeviceSynchronize();
    unsigned long long h[8];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("%schains %d, waves/SIMD %d, %2d %s per MFMA: %6.1f ticks per MFMA and SIMD\n", AG ? "AccVGPR accumulators, " : "", CH, waves_per_simd, K, KIND ? "tanh-mix instructions" : "v_fma_f32            ",
           (double)h[0] / ((double)n * CH * waves_per_simd));
    hipFree(d);
    hipFree(sink);
}

template <int KIND, int CH> void sweep(int w) {
    run<0, KIND, CH>(w);
    run<2, KIND, CH>(w);
    run<4, KIND, CH>(w);
    run<6, KIND, CH>(w);
    run<8, KIND, CH>(w);
    run<12, KIND, CH>(w);
    run<16, KIND, CH>(w);
}

int main() {
    for (int w = 1; w <= 2; w++) {
        sweep<0, 1>(w);
        sweep<1, 1>(w);
    }
    sweep<0, 2>(1);
    sweep<1, 2>(1);
    run<0, 1, 1, 1>(2);
    run<4, 1, 1, 1>(2);
    run<8, 1, 1, 1>(2);
    run<12, 1, 1, 1>(2);
    run<8, 0, 1, 1>(2);
    run<16, 0, 1, 1>(2);
    return 0;
}
