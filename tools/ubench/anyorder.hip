// Does hipExtAnyOrderLaunch let two INDEPENDENT kernels of ONE stream overlap on gfx950 / ROCm 7.2?
// (hip_ext.h says the flag is "not supported on GFX9xx"; measured instead of believed.)
// Each kernel: G workgroups spin for T microseconds (s_memrealtime, 100 MHz) and record first-start / last-end stamps.
//   build:  hipcc --offload-arch=gfx950 -O2 tools/ubench/anyorder.hip -o tools/ubench/anyorder
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdlib.h>

__global__ void spin_kernel(unsigned long long* stamps, int slot, int us) {
    const unsigned long long t0 = __builtin_readcyclecounter() * 0 + wall_clock64();
    if (threadIdx.x == 0) atomicMin(&stamps[2 * slot], t0);
    while (wall_clock64() - t0 < (unsigned long long)us * 100ull) { __builtin_amdgcn_s_sleep(8); }
    if (threadIdx.x == 0) atomicMax(&stamps[2 * slot + 1], wall_clock64());
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main(int argc, char** argv) {
    const int G = argc > 1 ? atoi(argv[1]) : 64, us = argc > 2 ? atoi(argv[2]) : 100, NK = 4;
    unsigned long long* st;
    CK(hipMalloc(&st, 64 * sizeof(unsigned long long)));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 3; ++mode) {           // 0: plain launches, 1: any-order from the 2nd on, 2: all any-order
        unsigned long long init[64];
        for (int i = 0; i < 32; ++i) { init[2 * i] = ~0ull; init[2 * i + 1] = 0; }
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemcpy(st, init, sizeof init, hipMemcpyHostToDevice));
            CK(hipEventRecord(e0, s));
            for (int k = 0; k < NK; ++k) {
                const unsigned flags = (mode == 2 || (mode == 1 && k > 0)) ? hipExtAnyOrderLaunch : 0;
                hipExtLaunchKernelGGL(spin_kernel, dim3(G), dim3(256), 0, s, nullptr, nullptr, flags, st, k, us);
            }
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            unsigned long long h[64]; CK(hipMemcpy(h, st, sizeof h, hipMemcpyDeviceToHost));
            printf("mode %d rep %d: %d kernels x %d wgs x %d us: total %.1f us;", mode, rep, NK, G, us, ms * 1e3);
            for (int k = 0; k < NK; ++k) printf("  k%d [%.1f, %.1f]", k, (h[2 * k] - h[0]) * 0.01, (h[2 * k + 1] - h[0]) * 0.01);
            printf("\n");
        }
    }
    return 0;
}
