#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
int main(int argc, char** argv) {
    hipModule_t mod; if (hipModuleLoad(&mod, argv[1]) != hipSuccess) { printf("load failed\n"); return 1; }
    const unsigned blocks = 1 << 15, bs = 256; size_t n = (size_t)blocks * bs;
    unsigned* out; hipMalloc(&out, n * 4);
    std::vector<unsigned> h(n);
    for (int i = 2; i < argc; i++) {
        char name[64]; snprintf(name, sizeof name, "regtest%s", argv[i]);
        hipFunction_t fn; if (hipModuleGetFunction(&fn, mod, name) != hipSuccess) { printf("no %s\n", name); continue; }
        for (int rep = 0; rep < 2; rep++) {
            hipMemset(out, 0xff, n * 4);
            struct { unsigned* out; unsigned bx, by, bz; unsigned short gx, gy, gz; } A = {out, blocks, 1, 1, (unsigned short)bs, 1, 1};
            size_t sz = sizeof(A);
            void* cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &A, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
            hipError_t e = hipModuleLaunchKernel(fn, blocks, 1, 1, bs, 1, 1, 0, 0, nullptr, cfg);
            hipError_t e2 = hipDeviceSynchronize();
            hipMemcpy(h.data(), out, n * 4, hipMemcpyDeviceToHost);
            size_t bad = 0; unsigned m = 0; for (size_t k = 0; k < n; k++) if (h[k]) { bad++; m |= h[k]; }
            printf("%s rep=%d launch=%d sync=%d lanes with a corrupted register: %zu of %zu, register mask %08x\n", name, rep, (int)e, (int)e2, bad, n, m);
        }
    }
    return 0;
}
