// Micro-benchmark behind DESIGN.md "BatchNorm finalize in the consumer's prologue": what do the fixed-point integer
// atomics of the producers cost, and what does a consumer pay for reducing 8 rows itself, against the separate
// finalize launch.   hipcc --offload-arch=gfx950 -O3 bn_atomics.hip -o bn_atomics && ./bn_atomics
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ int xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7; }

// producer: reads `bytes_per_wg` of input (stand-in for the conv), then publishes C x 2 per-channel sums
// mode 0: partial row per workgroup (float2 stores)   1: int64 atomics, agent scope, row = blockIdx % 8
// mode 2: int64 atomics, workgroup scope (L2-local), row = XCC id        3: nothing
template <int MODE>
__global__ __launch_bounds__(256) void producer(const float4* in, int n4_per_wg, float* rows_f, long long* rows_i, int C, float* sink) {
    float acc = 0.f;
    const float4* p = in + (size_t)blockIdx.x * n4_per_wg;
    for (int i = threadIdx.x; i < n4_per_wg; i += 256) { float4 v = p[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 12345.678f) sink[0] = acc;
    for (int c = threadIdx.x; c < C; c += 256) {
        const float s1 = acc + c, s2 = acc * acc + c;
        if (MODE == 0) {
            reinterpret_cast<float2*>(rows_f)[(size_t)blockIdx.x * C + c] = make_float2(s1, s2);
        } else if (MODE == 1) {
            long long* r = rows_i + ((size_t)(blockIdx.x & 7) * C + c) * 2;
            __hip_atomic_fetch_add(r, (long long)(s1 * 1048576.f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(r + 1, (long long)(s2 * 1048576.f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (MODE == 2) {
            long long* r = rows_i + ((size_t)xcc_id() * C + c) * 2;
            __hip_atomic_fetch_add(r, (long long)(s1 * 1048576.f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(r + 1, (long long)(s2 * 1048576.f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
}

// separate finalize: rows_f[rows][C][2] -> scale/shift   (8 channels x 32 row slices per workgroup of 256)
__global__ __launch_bounds__(256) void finalize_f(const float* rows_f, int rows, int C, float* scale, float* shift, float cnt) {
    __shared__ float red[256][2];
    const int c = blockIdx.x * 8 + (threadIdx.x & 7), sl = threadIdx.x >> 3;
    float s1 = 0.f, s2 = 0.f;
    for (int r = sl; r < rows; r += 32) { float2 v = reinterpret_cast<const float2*>(rows_f)[(size_t)r * C + c]; s1 += v.x; s2 += v.y; }
    red[threadIdx.x][0] = s1; red[threadIdx.x][1] = s2;
    __syncthreads();
    if (sl == 0) {
        for (int k = 1; k < 32; ++k) { s1 += red[k * 8 + (threadIdx.x & 7)][0]; s2 += red[k * 8 + (threadIdx.x & 7)][1]; }
        const float mean = s1 / cnt, var = s2 / cnt - mean * mean, is = rsqrtf(fabsf(var) + 1e-5f);
        scale[c] = is; shift[c] = -mean * is;
    }
}
__global__ __launch_bounds__(256) void finalize_i(long long* rows_i, int C, float* scale, float* shift, float cnt) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    long long s1 = 0, s2 = 0;
    for (int r = 0; r < 8; ++r) { s1 += rows_i[((size_t)r * C + c) * 2]; s2 += rows_i[((size_t)r * C + c) * 2 + 1]; }
    const float mean = (float)s1 * (1.f / 1048576.f) / cnt, var = (float)s2 * (1.f / 1048576.f) / cnt - mean * mean, is = rsqrtf(fabsf(var) + 1e-5f);
    scale[c] = is; shift[c] = -mean * is;
}

// consumer: PRO = 1 reduces the 8 integer rows itself (every workgroup), writes scale/shift, barrier; then reads its
// constants from global and streams `n4_per_wg` of input
template <int PRO>
__global__ __launch_bounds__(256) void consumer(const float4* in, int n4_per_wg, const long long* rows_i, int C, float* scale, float* shift, float cnt,
                                                float4* out) {
    if (PRO) {
        for (int c = threadIdx.x; c < C; c += 256) {
            long long s1 = 0, s2 = 0;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const longlong2 v = *reinterpret_cast<const longlong2*>(rows_i + ((size_t)r * C + c) * 2);
                s1 += v.x; s2 += v.y;
            }
            const float mean = (float)s1 * (1.f / 1048576.f) / cnt, var = (float)s2 * (1.f / 1048576.f) / cnt - mean * mean, is = rsqrtf(fabsf(var) + 1e-5f);
            scale[c] = is; shift[c] = -mean * is;
        }
        __syncthreads();
    }
    const int c0 = (threadIdx.x * 4) % C;
    const float k0 = scale[c0], k1 = shift[c0];
    const float4* p = in + (size_t)blockIdx.x * n4_per_wg;
    float4* q = out + (size_t)blockIdx.x * n4_per_wg;
    for (int i = threadIdx.x; i < n4_per_wg; i += 256) { float4 v = p[i]; v.x = v.x * k0 + k1; v.y = v.y * k0 + k1; v.z = v.z * k0 + k1; v.w = v.w * k0 + k1; q[i] = v; }
}

int main() {
    const int C = 256;
    const size_t BUF = 256u << 20;
    float4 *in, *out; float *rows_f, *scale, *shift, *sink; long long* rows_i;
    CK(hipMalloc(&in, BUF)); CK(hipMalloc(&out, BUF)); CK(hipMalloc(&rows_f, 8192 * C * 2 * 4)); CK(hipMalloc(&rows_i, 8 * C * 2 * 8));
    CK(hipMalloc(&scale, C * 4)); CK(hipMalloc(&shift, C * 4)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(in, 0, BUF)); CK(hipMemset(rows_i, 0, 8 * C * 2 * 8));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grids[4] = {6, 96, 384, 1536};
    const int kb_per_wg[4] = {64, 64, 64, 48};
    const int REP = 200;
    for (int gi = 0; gi < 4; ++gi) {
        const int G = grids[gi], n4 = kb_per_wg[gi] * 1024 / 16;
        float ms[8];
        for (int variant = 0; variant < 7; ++variant) {
            for (int pass = 0; pass < 2; ++pass) {
                CK(hipEventRecord(e0, st));
                for (int it = 0; it < REP; ++it) {
                    switch (variant) {
                        case 0:   // producer (no stats) -> consumer: the floor
                            hipLaunchKernelGGL(producer<3>, dim3(G), dim3(256), 0, st, in, n4, rows_f, rows_i, C, sink);
                            hipLaunchKernelGGL(consumer<0>, dim3(G), dim3(256), 0, st, in, n4, rows_i, C, scale, shift, 1000.f, out); break;
                        case 1:   // today: partial rows + finalize launch + consumer
                            hipLaunchKernelGGL(producer<0>, dim3(G), dim3(256), 0, st, in, n4, rows_f, rows_i, C, sink);
                            hipLaunchKernelGGL(finalize_f, dim3(C / 8), dim3(256), 0, st, rows_f, G, C, scale, shift, 1000.f);
                            hipLaunchKernelGGL(consumer<0>, dim3(G), dim3(256), 0, st, in, n4, rows_i, C, scale, shift, 1000.f, out); break;
                        case 2:   // agent-scope atomics + finalize launch + consumer
                            hipLaunchKernelGGL(producer<1>, dim3(G), dim3(256), 0, st, in, n4, rows_f, rows_i, C, sink);
                            hipLaunchKernelGGL(finalize_i, dim3(1), dim3(256), 0, st, rows_i, C, scale, shift, 1000.f);
                            hipLaunchKernelGGL(consumer<0>, dim3(G), dim3(256), 0, st, in, n4, rows_i, C, scale, shift, 1000.f, out); break;
                        case 3:   // agent-scope atomics + consumer prologue
                            hipLaunchKernelGGL(producer<1>, dim3(G), dim3(256), 0, st, in, n4, rows_f, rows_i, C, sink);
                            hipLaunchKernelGGL(consumer<1>, dim3(G), dim3(256), 0, st, in, n4, rows_i, C, scale, shift, 1000.f, out); break;
                        case 4:   // L2-local atomics (row = XCC id) + consumer prologue
                            hipLaunchKernelGGL(producer<2>, dim3(G), dim3(256), 0, st, in, n4, rows_f, rows_i, C, sink);
                            hipLaunchKernelGGL(consumer<1>, dim3(G), dim3(256), 0, st, in, n4, rows_i, C, scale, shift, 1000.f, out); break;
                        case 5:   // producers alone: rows
                            hipLaunchKernelGGL(producer<0>, dim3(G), dim3(256), 0, st, in, n4, rows_f, rows_i, C, sink); break;
                        case 6:   // producers alone: agent atomics
                            hipLaunchKernelGGL(producer<1>, dim3(G), dim3(256), 0, st, in, n4, rows_f, rows_i, C, sink); break;
                    }
                }
                CK(hipEventRecord(e1, st));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms[variant], e0, e1));
            }
        }
        printf("G=%4d (%d KB/WG): floor %.2f | rows+finalize %.2f | agent-atomics+finalize %.2f | agent-atomics+prologue %.2f | xcc-atomics+prologue %.2f | producer rows %.2f | producer agent-atomics %.2f  us per producer->consumer pair\n",
               G, kb_per_wg[gi], 1e3 * ms[0] / REP, 1e3 * ms[1] / REP, 1e3 * ms[2] / REP, 1e3 * ms[3] / REP, 1e3 * ms[4] / REP, 1e3 * ms[5] / REP, 1e3 * ms[6] / REP);
    }
    // correctness of the L2-local rows: total over the 8 rows must equal the agent-scope total
    std::vector<long long> h(8 * C * 2);
    CK(hipMemset(rows_i, 0, 8 * C * 2 * 8));
    hipLaunchKernelGGL(producer<2>, dim3(1536), dim3(256), 0, st, in, 64, rows_f, rows_i, C, sink);
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(h.data(), rows_i, h.size() * 8, hipMemcpyDeviceToHost));
    long long bad = 0; long long per_row[8] = {0};
    for (int c = 0; c < C; ++c) {
        long long t = 0; for (int r = 0; r < 8; ++r) { t += h[(r * C + c) * 2]; per_row[r] += h[(r * C + c) * 2] != 0; }
        if (t != (long long)1536 * (long long)(c * 1048576.0)) ++bad;
    }
    printf("xcc rows: %lld channels with a wrong total; rows used:", bad);
    for (int r = 0; r < 8; ++r) printf(" %lld", per_row[r]);
    printf("\n");
    return 0;
}
