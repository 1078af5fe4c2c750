// Micro-benchmark for the fused low-resolution sub-hourglass: G persistent workgroups (one image each) exchange per-channel
// BatchNorm partial sums through global memory and meet at a counter barrier, K times in one launch.
//   publish : C x float2 per workgroup with device-scope (sc1, write-through) 8-byte stores, s_waitcnt vmcnt(0), one device-scope
//             atomic add on a monotonic counter
//   wait    : lane 0 polls the counter (device-scope relaxed loads + s_sleep), __syncthreads
//   collect : every workgroup sums the G rows in a fixed order (device-scope loads) -> identical, reproducible totals
// hipcc --offload-arch=gfx950 -O3 grid_barrier.hip -o grid_barrier && ./grid_barrier
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(512) void chain(float2* rows, unsigned* counter, int C, int K, float* out, int work) {
    const int G = gridDim.x, g = blockIdx.x, tid = threadIdx.x;
    __shared__ float tot[512];
    float carry = 1.f + g;
    for (int k = 0; k < K; ++k) {
        // some per-workgroup work between the barriers (FMA chain ~ `work` x 4 cycles)
        float v = carry;
        for (int i = 0; i < work; ++i) v = fmaf(v, 1.0001f, 0.5f);
        float2* mine = rows + ((size_t)(k & 1) * G + g) * C;          // two row sets: a fast workgroup's next publish cannot race a slow reader
        for (int c = tid; c < C; c += blockDim.x) {
            float2 p = make_float2(v + c, v * 0.5f + c);
            __hip_atomic_store(reinterpret_cast<unsigned long long*>(mine + c), *reinterpret_cast<unsigned long long*>(&p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (unsigned)(k + 1) * G;
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
        const float2* set = rows + (size_t)(k & 1) * G * C;
        float s = 0.f;
        for (int c = tid; c < C; c += blockDim.x) {
            float a = 0.f, b = 0.f;
            for (int r = 0; r < G; ++r) {
                unsigned long long u = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(set + (size_t)r * C + c), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                float2 p = *reinterpret_cast<float2*>(&u);
                a += p.x; b += p.y;
            }
            s += a * 1e-6f + b * 1e-7f;
        }
        tot[tid] = s;
        __syncthreads();
        carry = tot[(tid * 7) & 511] * 1e-3f + 1.f;
        __syncthreads();
    }
    if (tid == 0) out[g] = carry;
}

int main() {
    float2* rows; unsigned* counter; float* out;
    CK(hipMalloc(&rows, 2 * 256 * 256 * sizeof(float2))); CK(hipMalloc(&counter, 64)); CK(hipMalloc(&out, 1024));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int Gs[4] = {2, 24, 48, 96};
    for (int gi = 0; gi < 4; ++gi)
        for (int C = 128; C <= 256; C += 128)
            for (int work = 0; work <= 2000; work += 2000) {
                const int G = Gs[gi], K = 200;
                float ms = 0.f;
                for (int pass = 0; pass < 2; ++pass) {
                    CK(hipMemsetAsync(counter, 0, 64, st));
                    CK(hipEventRecord(e0, st));
                    hipLaunchKernelGGL(chain, dim3(G), dim3(512), 0, st, rows, counter, C, K, out, work);
                    CK(hipEventRecord(e1, st));
                    CK(hipEventSynchronize(e1));
                    CK(hipEventElapsedTime(&ms, e0, e1));
                }
                printf("G=%3d C=%3d work=%4d: %.2f us per publish+barrier+collect (launch of %d steps %.1f us)\n", G, C, work, 1e3 * ms / K, K, 1e3 * ms);
            }
    return 0;
}
