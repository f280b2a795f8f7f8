// shift64_probe.cpp -- loads shift64 probe code objects (tools/hazard24/shift64_probe.s.in) and counts lanes whose 64-bit shift
// differed from the reference shift.  usage: shift64_probe <hsaco>... ; see run_shift64_probe.sh
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
int main(int argc, char** argv) {
    unsigned int* d = nullptr;
    CK(hipMalloc(&d, 4));
    for (int k = 1; k < argc; k++) {
        hipModule_t mod;
        hipFunction_t fn;
        CK(hipModuleLoad(&mod, argv[k]));
        CK(hipModuleGetFunction(&fn, mod, "shift64_probe"));
        for (int blocks : {256, 8192}) {  // one workgroup per CU (4 waves: one per SIMD) / 32 per CU (queued, 8 waves per SIMD resident)
            unsigned int iters = 20000, h = 0;
            CK(hipMemset(d, 0, 4));
            struct { unsigned int* p; unsigned int n; } args = {d, iters};
            size_t sz = 12;
            void* cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
            CK(hipModuleLaunchKernel(fn, blocks, 1, 1, 256, 1, 1, 0, nullptr, nullptr, cfg));
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost));
            printf("%-60s blocks %5d: %u mismatching lane-results of %.3g\n", argv[k], blocks, h, (double)blocks * 256 * iters);
        }
        CK(hipModuleUnload(mod));
    }
    return 0;
}
