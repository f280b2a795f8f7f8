// microbench_valu.hip -- what the vector pipe charges for the instructions of k_policy_sample's tanh (acx_policy.hip): the
// quarter-rate transcendentals, full-rate f32 and packed-f32 arithmetic, alone and mixed (do they overlap?), one or two waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o tools/_build/mbvalu tools/microbench_valu.hip && tools/_build/mbvalu
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(2))) float f32x2;

#define EXP(r) asm volatile("v_exp_f32 %0, %0" : "+v"(r))
#define RCP(r) asm volatile("v_rcp_f32 %0, %0" : "+v"(r))
#define FMA(r) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(r) : "v"(k))
#define PKF(r) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(r) : "v"(kk))
#define X8(M, a) M(a[0]); M(a[1]); M(a[2]); M(a[3]); M(a[4]); M(a[5]); M(a[6]); M(a[7])
#define I8(M, N, a, b) M(a[0]); N(b[0]); M(a[1]); N(b[1]); M(a[2]); N(b[2]); M(a[3]); N(b[3]); M(a[4]); N(b[4]); M(a[5]); N(b[5]); M(a[6]); N(b[6]); M(a[7]); N(b[7])

template <int MODE> __global__ void __launch_bounds__(512) k_valu(int n, unsigned long long* out, float* sink) {
    float e[8], f[8], g[8];
    f32x2 p[8], q[8];
    const float k = 0.999f + 1e-9f * threadIdx.x;
    const f32x2 kk = {k, k};
    for (int j = 0; j < 8; j++) {
        e[j] = 0.001f * (threadIdx.x + j);
        f[j] = 0.002f * (threadIdx.x + j);
        g[j] = 0.003f * (threadIdx.x + j);
        p[j] = f32x2{e[j], f[j]};
        q[j] = f32x2{g[j], f[j]};
    }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; i++) {
        if (MODE == 0) { X8(EXP, e); }                                  // 8 transcendentals
        if (MODE == 1) { X8(EXP, e); X8(RCP, f); }                      // 16 transcendentals
        if (MODE == 2) { X8(FMA, f); X8(FMA, g); }                      // 16 full-rate f32
        if (MODE == 3) { I8(EXP, FMA, e, f); X8(FMA, g); }              // 8 transcendentals interleaved with 16 full-rate
        if (MODE == 4) { X8(PKF, p); }                                  // 8 packed f32 (16 values)
        if (MODE == 5) { I8(EXP, PKF, e, p); }                          // 8 transcendentals interleaved with 8 packed
        if (MODE == 6) { I8(EXP, PKF, e, p); X8(PKF, q); }              // 8 transcendentals, 16 packed
        if (MODE == 7) { I8(EXP, FMA, e, f); }                          // 8 transcendentals, 8 full-rate
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int j = 0; j < 8; j++) s += e[j] + f[j] + g[j] + p[j][0] + p[j][1] + q[j][0] + q[j][1];
    if (s == 12345.0f) sink[0] = s;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int MODE> void run(const char* what, int instr, int waves_per_simd) {
    const int n = 2000;
    unsigned long long* d;
    float* sink;
    hipMalloc(&d, 256 * 16 * 8);
    hipMalloc(&sink, 4);
    for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(k_valu<MODE>, dim3(256), dim3(256 * waves_per_simd), 0, 0, n, d, sink);
    hipDeviceSynchronize();
    unsigned long long h[8];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-52s waves/SIMD %d: %6.1f ticks per iteration and SIMD (%d instructions per wave)\n", what, waves_per_simd, (double)h[0] / n, instr);
    hipFree(d);
    hipFree(sink);
}

int main() {
    for (int w = 1; w <= 2; w++) {
        run<0>("8 v_exp_f32", 8, w);
        run<1>("8 v_exp_f32 + 8 v_rcp_f32", 16, w);
        run<2>("16 v_fma_f32", 16, w);
        run<7>("8 v_exp_f32 interleaved with 8 v_fma_f32", 16, w);
        run<3>("8 v_exp_f32 interleaved with 16 v_fma_f32", 24, w);
        run<4>("8 v_pk_fma_f32", 8, w);
        run<5>("8 v_exp_f32 interleaved with 8 v_pk_fma_f32", 16, w);
        run<6>("8 v_exp_f32 interleaved with 16 v_pk_fma_f32", 24, w);
    }
    return 0;
}
