// Streaming (HBM-bound) kernels around the convolutions: BatchNorm bookkeeping, 2x2 max pool,
// nearest-upsample + add, their backward passes with the ReLU mask / BatchNorm-backward reductions
// fused in, RMSprop, weight re-packing and layout conversion.  NHWC bf16, 16-byte accesses
// (8 channels per thread), per-channel reductions staged through LDS atomics, one global atomic per
// channel per workgroup.
#include "common.h"
#include "kernels.h"
// wave priority of the main chain's finalize / pooling / upsampling kernels beside the weight-gradient group kernels (priority 0): 5.863 -> 5.842 ms (four interleaved pairs, round 5)
#ifndef PA_ELT_PRIO
#define PA_ELT_PRIO 2
#endif
#define PA_SET_ELT_PRIO() do { if (PA_ELT_PRIO > 0) __builtin_amdgcn_s_setprio(PA_ELT_PRIO); } while (0)
#include "bn_fin.h"
#include <type_traits>
#include <hip/hip_ext.h>
// TIMING EXPERIMENT (tuning builds, PA_FIN_ANYORDER=1; results may be WRONG): the finalize launches without the completion / cache round trip
// behind their producer (hipExtAnyOrderLaunch) -- what a tagged-row hand-off from the producers to the finalize kernel could return at most
static unsigned fin_launch_flags() { static int v = -1; if (v < 0) { const char* e = pa_getenv("PA_FIN_ANYORDER"); v = e ? atoi(e) : 0; } return v ? hipExtAnyOrderLaunch : 0u; }

// ------------------------------------------------------------------------------------------------
// BatchNorm bookkeeping (reference semantics: torch.nn.BatchNorm2d, eps 1e-5, momentum 0.1,
// biased variance for normalisation, unbiased for the running estimate).
// Sum `rows` partial rows [rows][C][2] for the FC channels of this workgroup: FT threads =
// FC channels x (FT / FC) row slices, float2 loads, LDS tree over the slices.  Result in LDS sums[FC][2].
// FC = 8: 16-32 workgroups per BatchNorm (the kernel is pure latency: launch + row loads in batches + tree; 32 channels per workgroup
// took 6 us; FC = 4 -- every common row count in ONE batch of loads, twice the workgroups -- measured 0.03 ms per step slower, round 4).
// workgroup shape of the finalize launches: FT threads = FC channels x FT / FC row slices (tuning builds: PA_FIN_THREADS, PA_FIN_FC)
template <class F>
static void fin_dispatch(F f) {
    // 512 threads (8 channels x 64 row slices): measured 6.44 - 6.49 ms per step against 6.50 - 6.54 with 1024-thread workgroups (three
    // interleaved pairs; 256 threads, or 4 channels per workgroup: the same as 512 / 8) -- a 1024-thread workgroup has to find a whole CU's
    // worth of wave slots next to the other queues' kernels, and the launch is all these kernels are
    static int ft = -1;
    if (ft < 0) { const char* e = pa_getenv("PA_FIN_THREADS"); ft = e ? atoi(e) : 512; }
    if (ft == 256) f(std::integral_constant<int, 256>{}, std::integral_constant<int, 8>{});
    else if (ft == 1024) f(std::integral_constant<int, 1024>{}, std::integral_constant<int, 8>{});
    else f(std::integral_constant<int, 512>{}, std::integral_constant<int, 8>{});
}
template <int FT, int FC>
__device__ __forceinline__ void reduce_partial_rows(const float* part, int rows, int C, int c0, float (*sums)[2]) {
    constexpr int SL = FT / FC;
    __shared__ __attribute__((aligned(16))) float red[SL][FC + 1][2];
    const int cl = threadIdx.x % FC, sl = threadIdx.x / FC;
    if (rows <= PA_FIN_SMALL_ROWS) {
        // the order the consumer-prologue form of the finalize uses too (bn_fin.h): sixteen interleaved chains, stride-halving tree;
        // here one thread per chain and channel pair (4 pairs x 16 chains), ONE round of <= 8 loads each
        constexpr int NP = FC / 2;                                       // channel pairs of this workgroup
        f32x4* part4 = reinterpret_cast<f32x4*>(&red[0][0][0]);          // [16][NP]
        if (threadIdx.x < 16 * NP) {
            const int pr = threadIdx.x % NP, j = threadIdx.x / NP;
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            part4[j * NP + pr] = c0 + 2 * pr < C ? pa_fin_chain16(part, rows, C, c0 + 2 * pr, j) : z;
        }
        __syncthreads();
        if (threadIdx.x < FC) {
            const int c = threadIdx.x, half = c & 1;
            float a[16], b[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) { const f32x4 v = part4[i * NP + (c >> 1)]; a[i] = half ? v[2] : v[0]; b[i] = half ? v[3] : v[1]; }
#pragma unroll
            for (int n = 16; n > 1; n >>= 1)
#pragma unroll
                for (int i = 0; i < n / 2; ++i) { a[i] += a[i + n / 2]; b[i] += b[i + n / 2]; }
            sums[c][0] = a[0]; sums[c][1] = b[0];
        }
        __syncthreads();
        return;
    }
    float a = 0.f, b = 0.f;
    if (c0 + cl < C) {
        // the thread's rows sl, sl + SL, ... in batches of NB loads in flight (clamped, unconditional; the sums stay in increasing order):
        // hipcc compiled the plain loop `#pragma unroll 8` into a one-load-one-wait remainder loop in FRONT of the unrolled body, i.e. 3 / 6
        // serial memory round trips at 384 / 768 rows in a kernel that is nothing but latency.  NB = 4 up to 512 rows (one round trip,
        // <= 3 loads wasted), 8 above (768 rows: one round trip; 1536: two)
        auto batches = [&](auto nbc) {
            constexpr int NB = decltype(nbc)::value;
            for (int r0 = sl; r0 < rows; r0 += NB * SL) {
                f32x2 v[NB];
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    const int r = r0 + u * SL;
                    v[u] = *reinterpret_cast<const f32x2*>(part + ((size_t)(r < rows ? r : r0) * C + c0 + cl) * 2);
                }
#pragma unroll
                for (int u = 0; u < NB; ++u)
                    if (r0 + u * SL < rows) { a += v[u][0]; b += v[u][1]; }
            }
        };
        if (rows <= 4 * SL) batches(std::integral_constant<int, 4>{});
        else if (rows <= 8 * SL) batches(std::integral_constant<int, 8>{});
        else batches(std::integral_constant<int, 12>{});      // (1536 rows at 128 slices: ONE round trip instead of 8 + 4)
    }
    red[sl][cl][0] = a; red[sl][cl][1] = b;
    __syncthreads();
    // two-level tree over the SL slices: 16 groups of SL/16 slices, then the 16 group sums
    const bool act = threadIdx.x < FC * 16;
    const int tc = threadIdx.x % FC, tg = threadIdx.x / FC;
    float x2 = 0.f, y2 = 0.f;
    if (act) {
#pragma unroll
        for (int s = tg * (SL / 16); s < (tg + 1) * (SL / 16); ++s) { x2 += red[s][tc][0]; y2 += red[s][tc][1]; }
    }
    __syncthreads();
    if (act) { red[tg][tc][0] = x2; red[tg][tc][1] = y2; }
    __syncthreads();
    if (threadIdx.x < FC) {
        float x = 0.f, y = 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s) { x += red[s][threadIdx.x][0]; y += red[s][threadIdx.x][1]; }
        sums[threadIdx.x][0] = x; sums[threadIdx.x][1] = y;
    }
    __syncthreads();
}

template <int FT, int FC>
__global__ __launch_bounds__(FT) void bn_finalize_kernel(const float* stats, int rows, const float* gamma, const float* beta,
                                                           float* rmean, float* rvar, float* scale, float* shift, float* mean,
                                                           float* invstd, int C, float count, float momentum, float eps,
                                                           int update_running) {
    PA_SET_ELT_PRIO();
    __shared__ float sums[FC][2];
    const int c0 = blockIdx.x * FC;
    // the channel's parameters are requested BEFORE the reduction (after it they were one more dependent memory round trip of a kernel that
    // is nothing but round trips)
    const bool mine = threadIdx.x < FC && c0 + (int)threadIdx.x < C;
    float g = 0.f, bt = 0.f, rm = 0.f, rv = 0.f;
    if (mine) {
        g = gamma[c0 + threadIdx.x]; bt = beta[c0 + threadIdx.x];
        if (update_running) { rm = rmean[c0 + threadIdx.x]; rv = rvar[c0 + threadIdx.x]; }
    }
    reduce_partial_rows<FT, FC>(stats, rows, C, c0, sums);
    if (!mine) return;
    const int c = c0 + threadIdx.x;
    float s, sh, mu, is, var;
    pa_bn_fwd_consts(sums[threadIdx.x][0], sums[threadIdx.x][1], count, eps, g, bt, s, sh, mu, is, var);
    scale[c] = s;
    shift[c] = sh;
    mean[c] = mu;
    invstd[c] = is;
    if (update_running) {
        pa_bn_running(mu, var, count, momentum, rm, rv);
        rmean[c] = rm; rvar[c] = rv;
    }
}

int pa_launch_bn_finalize(const float* stats, int rows, const float* gamma, const float* beta, float* rmean, float* rvar,
                          float* scale, float* shift, float* mean, float* invstd, int C, float count,
                          float momentum, float eps, int update_running, hipStream_t st) {
    if (C & 1) { pa_set_error_msg("BatchNorm finalize: the statistics rows are read as 16-byte channel pairs -- C must be even (the networks pad to 64)"); return 1; }
    fin_dispatch([&](auto ft, auto fc) {
        constexpr int FT = decltype(ft)::value, FC = decltype(fc)::value;
        if (fin_launch_flags())
            hipExtLaunchKernelGGL((bn_finalize_kernel<FT, FC>), dim3((C + FC - 1) / FC), dim3(FT), 0, st, nullptr, nullptr, fin_launch_flags(), stats, rows, gamma, beta, rmean, rvar, scale,
                                  shift, mean, invstd, C, count, momentum, eps, update_running);
        else
            hipLaunchKernelGGL((bn_finalize_kernel<FT, FC>), dim3((C + FC - 1) / FC), dim3(FT), 0, st, stats, rows, gamma, beta, rmean, rvar, scale,
                               shift, mean, invstd, C, count, momentum, eps, update_running);
    });
    return (int)hipGetLastError();
}

__global__ void bn_eval_kernel(const PaBnEvalJob* jobs, float eps) {
    const PaBnEvalJob j = jobs[blockIdx.x];
    for (int c = threadIdx.x; c < j.C; c += blockDim.x) {
        float s = j.gamma[c] * rsqrtf(j.rvar[c] + eps);
        j.scale[c] = s;
        j.shift[c] = j.beta[c] - j.rmean[c] * s;
    }
}

int pa_launch_bn_eval(const PaBnEvalJob* jobs_dev, int njobs, float eps, hipStream_t st) {
    if (njobs <= 0) return 0;
    hipLaunchKernelGGL(bn_eval_kernel, dim3(njobs), dim3(256), 0, st, jobs_dev, eps);
    return (int)hipGetLastError();
}

// dx = s*(dz - S1/M - xhat*S2/M) = kA*dz + kB*x + kC    (per channel)
template <int FT, int FC>
__global__ __launch_bounds__(FT) void bn_bwd_finalize_kernel(const float* bstats, int rows, const float* scale, const float* mean,
                                                               const float* invstd, float* kA, float* kB, float* kC,
                                                               float* dgamma, float* dbeta, int C, float count) {
    PA_SET_ELT_PRIO();
    __shared__ float sums[FC][2];
    const int c0 = blockIdx.x * FC;
    const bool mine = threadIdx.x < FC && c0 + (int)threadIdx.x < C;
    float sc = 0.f, is = 0.f, mu = 0.f;              // (requested before the reduction, see bn_finalize_kernel)
    if (mine) { sc = scale[c0 + threadIdx.x]; is = invstd[c0 + threadIdx.x]; mu = mean[c0 + threadIdx.x]; }
    reduce_partial_rows<FT, FC>(bstats, rows, C, c0, sums);
    if (!mine) return;
    const int c = c0 + threadIdx.x;
    float S1 = sums[threadIdx.x][0], S2 = sums[threadIdx.x][1];
    float ka, kb, kc;
    pa_bn_bwd_consts(S1, S2, count, sc, is, mu, ka, kb, kc);
    kA[c] = ka; kB[c] = kb; kC[c] = kc;
    if (dgamma) dgamma[c] = S2;
    if (dbeta) dbeta[c] = S1;
}

// two BatchNorm-backward finalizes in one launch (blockIdx.y picks the layer): the upsample-add backward feeds two layers at once
struct BwdFinArgs { const float* bstats; int rows; const float* scale; const float* mean; const float* invstd; float *kA, *kB, *kC, *dgamma, *dbeta; int C; float count; };
template <int FT, int FC>
__global__ __launch_bounds__(FT) void bn_bwd_finalize2_kernel(BwdFinArgs a0, BwdFinArgs a1) {
    PA_SET_ELT_PRIO();
    const BwdFinArgs& a = blockIdx.y == 0 ? a0 : a1;
    __shared__ float sums[FC][2];
    const int c0 = blockIdx.x * FC;
    if (c0 >= a.C) return;
    const bool mine = threadIdx.x < FC && c0 + (int)threadIdx.x < a.C;
    float sc = 0.f, is = 0.f, mu = 0.f;
    if (mine) { sc = a.scale[c0 + threadIdx.x]; is = a.invstd[c0 + threadIdx.x]; mu = a.mean[c0 + threadIdx.x]; }
    reduce_partial_rows<FT, FC>(a.bstats, a.rows, a.C, c0, sums);
    if (!mine) return;
    const int c = c0 + threadIdx.x;
    float S1 = sums[threadIdx.x][0], S2 = sums[threadIdx.x][1];
    float ka, kb, kc;
    pa_bn_bwd_consts(S1, S2, a.count, sc, is, mu, ka, kb, kc);
    a.kA[c] = ka; a.kB[c] = kb; a.kC[c] = kc;
    if (a.dgamma) a.dgamma[c] = S2;
    if (a.dbeta) a.dbeta[c] = S1;
}

int pa_launch_bn_bwd_finalize2(const float* bs0, int rows0, const float* sc0, const float* mu0, const float* is0, float* kA0, float* kB0, float* kC0,
                               float* dg0, float* db0, int C0, float cnt0,
                               const float* bs1, int rows1, const float* sc1, const float* mu1, const float* is1, float* kA1, float* kB1, float* kC1,
                               float* dg1, float* db1, int C1, float cnt1, hipStream_t st) {
    BwdFinArgs a0 = {bs0, rows0, sc0, mu0, is0, kA0, kB0, kC0, dg0, db0, C0, cnt0};
    BwdFinArgs a1 = {bs1, rows1, sc1, mu1, is1, kA1, kB1, kC1, dg1, db1, C1, cnt1};
    if ((C0 | C1) & 1) { pa_set_error_msg("BatchNorm finalize: C must be even (16-byte channel pairs)"); return 1; }
    const int cm = C0 > C1 ? C0 : C1;
    fin_dispatch([&](auto ft, auto fc) {
        constexpr int FT = decltype(ft)::value, FC = decltype(fc)::value;
        if (fin_launch_flags()) hipExtLaunchKernelGGL((bn_bwd_finalize2_kernel<FT, FC>), dim3((cm + FC - 1) / FC, 2), dim3(FT), 0, st, nullptr, nullptr, fin_launch_flags(), a0, a1);
        else hipLaunchKernelGGL((bn_bwd_finalize2_kernel<FT, FC>), dim3((cm + FC - 1) / FC, 2), dim3(FT), 0, st, a0, a1);
    });
    return (int)hipGetLastError();
}

int pa_launch_bn_bwd_finalize(const float* bstats, int rows, const float* scale, const float* mean, const float* invstd,
                              float* kA, float* kB, float* kC, float* dgamma, float* dbeta, int C, float count,
                              hipStream_t st) {
    if (C & 1) { pa_set_error_msg("BatchNorm finalize: the statistics rows are read as 16-byte channel pairs -- C must be even (the networks pad to 64)"); return 1; }
    fin_dispatch([&](auto ft, auto fc) {
        constexpr int FT = decltype(ft)::value, FC = decltype(fc)::value;
        if (fin_launch_flags()) {
            hipExtLaunchKernelGGL((bn_bwd_finalize_kernel<FT, FC>), dim3((C + FC - 1) / FC), dim3(FT), 0, st, nullptr, nullptr, fin_launch_flags(), bstats, rows, scale, mean, invstd, kA, kB,
                           kC, dgamma, dbeta, C, count);
        } else {
            hipLaunchKernelGGL((bn_bwd_finalize_kernel<FT, FC>), dim3((C + FC - 1) / FC), dim3(FT), 0, st, bstats, rows, scale, mean, invstd, kA, kB,
                           kC, dgamma, dbeta, C, count);
        }
    });
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void load8_rt(const PaOperand& op, size_t idx, int c, float (&v)[8]) {
    if (op.mode == PA_LD_PLAIN) pa_load8<PA_LD_PLAIN>(op, idx, c, v);
    else if (op.mode == PA_LD_BNRELU) pa_load8<PA_LD_BNRELU>(op, idx, c, v);
    else if (op.mode == PA_LD_LIN2) pa_load8<PA_LD_LIN2>(op, idx, c, v);
    else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.f;
    }
}

// apply an epilogue to 8 channel values at element offset idx (channel c..c+7); s1/s2 accumulate the
// per-channel reductions of the STATS / BWD modes
__device__ __forceinline__ bf16x8 epilogue8(const PaEpilogue& ep, size_t idx, int c, const float (&v)[8],
                                            float (&s1)[8], float (&s2)[8]) {
    bf16x8 o;
    if (ep.mode == PA_OUT_BWD) {
        bf16x8 xr = *reinterpret_cast<const bf16x8*>(ep.xref + idx);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float x = (float)xr[j];
            float dz = (fmaf(ep.scale[c + j], x, ep.shift[c + j]) > 0.f) ? v[j] : 0.f;
            o[j] = (bf16)dz;
            float dzr = (float)o[j];
            s1[j] += dzr;
            s2[j] += dzr * (x - ep.mean[c + j]) * ep.invstd[c + j];
        }
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            o[j] = (bf16)v[j];
            float r = (float)o[j];
            s1[j] += r;
            s2[j] += r * r;
        }
    }
    return o;
}

// block-level flush of per-thread channel partials: LDS float atomics, then this workgroup's own
// partial row gstats[blockIdx.x][C][2] (plain stores; the BatchNorm finalize kernels sum the rows)
// Per-workgroup partial row of the two per-channel reductions of a streaming kernel whose threads own the
// 8-channel group  c/8 = threadIdx.x % (C/8): shuffle-reduce the lanes of a wave that share a group, one LDS
// row per wave, then a tree over the waves.  (LDS float atomics here cost 8-32 k serialized atomics per
// workgroup: ~10 us of tail.)  red: [nwaves][2*C] floats of LDS.
__device__ __forceinline__ void flush_stats(float* red, float* gstats, int C, int c, const float (&s1)[8], const float (&s2)[8]) {
    const int CG = C / 8, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    float a[8], b[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        a[j] = s1[j]; b[j] = s2[j];
        for (int o = 32; o >= CG; o >>= 1) { a[j] += __shfl_xor(a[j], o, 64); b[j] += __shfl_xor(b[j], o, 64); }
    }
    __syncthreads();                       // (callers may have used `red` before)
    if (lane < CG) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { red[wave * 2 * C + c + j] = a[j]; red[wave * 2 * C + C + c + j] = b[j]; }
    }
    __syncthreads();
    float* row = gstats + (size_t)blockIdx.x * C * 2;
    for (int i = threadIdx.x; i < C; i += blockDim.x) {
        float x = 0.f, y = 0.f;
        for (int w = 0; w < nw; ++w) { x += red[w * 2 * C + i]; y += red[w * 2 * C + C + i]; }
        f32x2 v = {x, y};
        *reinterpret_cast<f32x2*>(row + i * 2) = v;
    }
}

struct Bwd8 {            // one BatchNorm-backward epilogue on 8 channels: dz = g * [s*x+t > 0], sums of dz and dz*xhat
    __device__ __forceinline__ static bf16x8 apply(const float4* cst, int c, const bf16x8& xr, const float (&g)[8], float (&s1)[8], float (&s2)[8]) {
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float4 k = cst[c + j];
            const float x = (float)xr[j];
            const float dz = (fmaf(k.x, x, k.y) > 0.f) ? g[j] : 0.f;
            o[j] = (bf16)dz;
            const float dzr = (float)o[j];
            s1[j] += dzr;
            s2[j] += dzr * (x - k.z) * k.w;
        }
        return o;
    }
};

__device__ __forceinline__ void stage_bwd_consts(float4* dst, const PaEpilogue& ep, int C) {
    for (int i = threadIdx.x; i < C; i += blockDim.x) dst[i] = make_float4(ep.scale[i], ep.shift[i], ep.mean[i], ep.invstd[i]);
}

// ------------------------------------------------------------------------------------------------
// 2x2/2 max pool (reference models/asn_stacked_hg.py:69,227), input transform applied on load
__global__ void maxpool_fwd_kernel(PaOperand in, bf16* out, int B, int H, int W, int C) {
    const int Ho = H / 2, Wo = W / 2, CG = C / 8;
    const size_t total = (size_t)B * Ho * Wo * CG;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        int cg = (int)(t % CG);
        size_t p = t / CG;
        int xo = (int)(p % Wo);
        size_t q = p / Wo;
        int yo = (int)(q % Ho), b = (int)(q / Ho);
        int c = cg * 8;
        float m[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            size_t idx = (((size_t)b * H + 2 * yo + (k >> 1)) * W + 2 * xo + (k & 1)) * C + c;
            float v[8];
            load8_rt(in, idx, c, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) m[j] = (k == 0) ? v[j] : fmaxf(m[j], v[j]);
        }
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (bf16)m[j];
        *reinterpret_cast<bf16x8*>(out + p * C + c) = o;
    }
}


// ---- streaming forward variants (row indexing, template modes, thread-fixed channel constants in registers, all loads of an
// item issued first): OP 0 = 2x2 max pool of value(in), OP 1 = nearest-upsample(low) + value(skip); item = one low-resolution
// pixel x 8 channels (4 high-resolution pixels).  Operand modes PLAIN / BNRELU.
template <int OP, int MA, int MB>
__global__ __launch_bounds__(256) void pool_up_fwd_s_kernel(PaOperand a, PaOperand b, bf16* __restrict__ out, int rows, int W, int C) {
    PA_SET_ELT_PRIO();
    // rows = B * H/2 (low-resolution rows); W = high-resolution width
    const int CG = C / 8, Wl = W / 2;
    const unsigned row_items = (unsigned)Wl * CG;
    const int c = (threadIdx.x % CG) * 8;
    float a0[8], a1[8], b0[8], b1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if (MA == PA_LD_BNRELU) { a0[j] = a.k0[c + j]; a1[j] = a.k1[c + j]; }
        if (OP == 1 && MB == PA_LD_BNRELU) { b0[j] = b.k0[c + j]; b1[j] = b.k1[c + j]; }
    }
    const unsigned rows_per_it = row_items >= blockDim.x ? 1u : blockDim.x / row_items;
    const unsigned rsub = row_items >= blockDim.x ? 0u : threadIdx.x / row_items;
    const unsigned i0 = row_items >= blockDim.x ? threadIdx.x : threadIdx.x % row_items;
    const bf16* __restrict__ pa = a.p;
    const bf16* __restrict__ pb = b.p;
    for (unsigned rb = blockIdx.x * rows_per_it; rb < (unsigned)rows; rb += gridDim.x * rows_per_it) {
        const unsigned r = rb + rsub;
        if (r >= (unsigned)rows) continue;
        for (unsigned i = i0; i < row_items; i += blockDim.x) {
            const unsigned xl = i / (unsigned)CG;
            const unsigned top = ((2u * r) * (unsigned)W + 2u * xl) * (unsigned)C + (unsigned)c;
            const unsigned wc = (unsigned)W * (unsigned)C;
            const unsigned off[4] = {top, top + (unsigned)C, top + wc, top + wc + (unsigned)C};
            const unsigned li = (r * (unsigned)Wl + xl) * (unsigned)C + (unsigned)c;
            bf16x8 hi[4], lo;
#pragma unroll
            for (int k = 0; k < 4; ++k) hi[k] = *reinterpret_cast<const bf16x8*>((OP == 0 ? pa : pb) + off[k]);
            if (OP == 1) lo = *reinterpret_cast<const bf16x8*>(pa + li);
            if (OP == 0) {
                bf16x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float m = 0.f;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        float v = (float)hi[k][j];
                        if (MA == PA_LD_BNRELU) v = fmaxf(fmaf(a0[j], v, a1[j]), 0.f);
                        m = k == 0 ? v : fmaxf(m, v);
                    }
                    o[j] = (bf16)m;
                }
                *reinterpret_cast<bf16x8*>(out + li) = o;
            } else {
                float l[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float v = (float)lo[j];
                    if (MA == PA_LD_BNRELU) v = fmaxf(fmaf(a0[j], v, a1[j]), 0.f);
                    l[j] = v;
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    bf16x8 o;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        float v = (float)hi[k][j];
                        if (MB == PA_LD_BNRELU) v = fmaxf(fmaf(b0[j], v, b1[j]), 0.f);
                        o[j] = (bf16)(l[j] + v);
                    }
                    *reinterpret_cast<bf16x8*>(out + off[k]) = o;
                }
            }
        }
    }
}

// returns false when the configuration is not covered (caller uses the generic kernel)
template <int OP>
static bool launch_pool_up_fwd_s(const PaOperand& a, const PaOperand& b, bf16* out, int B, int H, int W, int C, hipStream_t st) {
    static int old = -1;
    if (old < 0) old = pa_getenv("PA_ELTWISE_OLD") ? 1 : 0;
    const int threads = 256;
    const size_t row_items = (size_t)(W / 2) * (C / 8);
    const bool ma = a.mode == PA_LD_PLAIN || a.mode == PA_LD_BNRELU, mb = OP == 0 || b.mode == PA_LD_PLAIN || b.mode == PA_LD_BNRELU;
    if (old || !ma || !mb || C % 8 || H % 2 || W % 2 || threads % (C / 8) != 0 || !(row_items % threads == 0 || threads % row_items == 0) ||
        (size_t)B * H * W * C >= ((size_t)1 << 31)) return false;
    const int rows = B * (H / 2);
    const size_t total = (size_t)rows * row_items;
    int blocks = (int)((total + threads - 1) / threads);
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    const bool A = a.mode == PA_LD_BNRELU, Bm = OP == 1 && b.mode == PA_LD_BNRELU;
    if (A && Bm) hipLaunchKernelGGL((pool_up_fwd_s_kernel<OP, PA_LD_BNRELU, PA_LD_BNRELU>), dim3(blocks), dim3(threads), 0, st, a, b, out, rows, W, C);
    else if (A) hipLaunchKernelGGL((pool_up_fwd_s_kernel<OP, PA_LD_BNRELU, PA_LD_PLAIN>), dim3(blocks), dim3(threads), 0, st, a, b, out, rows, W, C);
    else if (Bm) hipLaunchKernelGGL((pool_up_fwd_s_kernel<OP, PA_LD_PLAIN, PA_LD_BNRELU>), dim3(blocks), dim3(threads), 0, st, a, b, out, rows, W, C);
    else hipLaunchKernelGGL((pool_up_fwd_s_kernel<OP, PA_LD_PLAIN, PA_LD_PLAIN>), dim3(blocks), dim3(threads), 0, st, a, b, out, rows, W, C);
    return true;
}

int pa_launch_maxpool_fwd(const PaOperand& in, bf16* out, int B, int H, int W, int C, hipStream_t st) {
    if (launch_pool_up_fwd_s<0>(in, in, out, B, H, W, C, st)) return (int)hipGetLastError();
    size_t total = (size_t)B * (H / 2) * (W / 2) * (C / 8);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(blocks), dim3(256), 0, st, in, out, B, H, W, C);
    return (int)hipGetLastError();
}

// Streaming kernels with one partial-statistics row per workgroup (<= 512 rows): big tensors get larger workgroups so that 512 of
// them still keep enough loads in flight (256-thread ones, measured ALONE, left HBM half idle: 3.3 TB/s vs 6+ TB/s for a plain copy of the
// same size).
static inline void stream_launch_dims(size_t total, int& blocks, int& threads) {
    static int big = -1;
    // 512 threads for the big tensors (1024 until round 4: 6.465 vs 6.434 ms per step over three interleaved pairs -- a 1024-thread
    // workgroup needs a whole CU's wave slots at once next to the other queues' kernels; 512 workgroups x 512 threads x 9 loads of 16 bytes are
    // still 38 MB in flight)
    if (big < 0) { const char* e = pa_getenv("PA_STREAM_THREADS"); big = e ? atoi(e) : 512; }
    threads = total >= (size_t)512 * 1024 ? big : 256;
    blocks = (int)((total + threads - 1) / threads);
    if (blocks > 512) blocks = 512;          // one partial-statistics row per workgroup
    if (blocks < 1) blocks = 1;
}

// gradient of the max pool: routed to the first maximum in scan order (torch semantics), plus an
// optional addend (other consumers' gradient), then the epilogue of the tensor being differentiated
__global__ __launch_bounds__(1024) void maxpool_bwd_kernel(const bf16* dout, PaOperand in, PaOperand add, PaEpilogue ep, bf16* din,
                                   int B, int H, int W, int C) {
    extern __shared__ float red[];
    const int Ho = H / 2, Wo = W / 2, CG = C / 8;
    const size_t total = (size_t)B * Ho * Wo * CG;
    // blockDim (256) is a multiple of CG (<= 32), so a thread keeps its channel group across the loop
    float s1[8], s2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
    const int cg = threadIdx.x % CG, c = cg * 8;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        size_t p = t / CG;
        int xo = (int)(p % Wo);
        size_t q = p / Wo;
        int yo = (int)(q % Ho), b = (int)(q / Ho);
        bf16x8 g = *reinterpret_cast<const bf16x8*>(dout + p * C + c);
        float v[4][8];
        size_t idx[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            idx[k] = (((size_t)b * H + 2 * yo + (k >> 1)) * W + 2 * xo + (k & 1)) * C + c;
            load8_rt(in, idx[k], c, v[k]);
        }
        int arg[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int a = 0; float m = v[0][j];
#pragma unroll
            for (int k = 1; k < 4; ++k) if (v[k][j] > m) { m = v[k][j]; a = k; }
            arg[j] = a;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float e[8], r[8];
            load8_rt(add, idx[k], c, e);
#pragma unroll
            for (int j = 0; j < 8; ++j) r[j] = (arg[j] == k ? (float)g[j] : 0.f) + e[j];
            *reinterpret_cast<bf16x8*>(din + idx[k]) = epilogue8(ep, idx[k], c, r, s1, s2);
        }
    }
    if (ep.mode != PA_OUT_PLAIN) flush_stats(red, ep.stats, C, c, s1, s2);
}


// maxpool backward, streaming variant (see upadd_bwd_bb_kernel): row r of the pooled map covers input rows 2r, 2r+1.
// INMODE: PLAIN / BNRELU value of the pooled tensor (argmax); HASADD: plain addend; EPMODE: PLAIN or BatchNorm-backward.
template <int INMODE, bool HASADD, int EPMODE>
__global__ __launch_bounds__(1024) void maxpool_bwd_s_kernel(const bf16* __restrict__ dout, PaOperand in, const bf16* __restrict__ add,
                                                             PaEpilogue ep, bf16* __restrict__ din, int rows, int W, int C) {
    PA_SET_ELT_PRIO();
    extern __shared__ float red[];     // [nwaves][2*C] partial statistics, then constants: float4 {scale, shift, mean, invstd} or float2 {k0, k1}
    float4* cst = reinterpret_cast<float4*>(red + (blockDim.x / 64) * 2 * C);
    float2* kin = reinterpret_cast<float2*>(cst + (EPMODE == PA_OUT_BWD ? C : 0));
    if (EPMODE == PA_OUT_BWD) stage_bwd_consts(cst, ep, C);
    if (INMODE == PA_LD_BNRELU)
        for (int i = threadIdx.x; i < C; i += blockDim.x) kin[i] = make_float2(in.k0[i], in.k1[i]);
    __syncthreads();
    const int CG = C / 8, Wo = W / 2;
    const unsigned row_items = (unsigned)Wo * CG;
    float s1[8], s2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
    const int c = (threadIdx.x % CG) * 8;
    const unsigned rows_per_it = row_items >= blockDim.x ? 1u : blockDim.x / row_items;
    const unsigned rsub = row_items >= blockDim.x ? 0u : threadIdx.x / row_items;
    const unsigned i0 = row_items >= blockDim.x ? threadIdx.x : threadIdx.x % row_items;
    const bf16* __restrict__ xin = in.p;
    for (unsigned rb = blockIdx.x * rows_per_it; rb < (unsigned)rows; rb += gridDim.x * rows_per_it) {
        const unsigned r = rb + rsub;
        if (r >= (unsigned)rows) continue;
        for (unsigned i = i0; i < row_items; i += blockDim.x) {
            const unsigned xo = i / (unsigned)CG;
            int cc = c;
            asm volatile("" : "+v"(cc));
            const unsigned top = ((2u * r) * (unsigned)W + 2u * xo) * (unsigned)C + (unsigned)c;
            const unsigned wc = (unsigned)W * (unsigned)C;
            const unsigned off[4] = {top, top + (unsigned)C, top + wc, top + wc + (unsigned)C};
            const bf16x8 g = *reinterpret_cast<const bf16x8*>(dout + (r * (unsigned)Wo + xo) * (unsigned)C + (unsigned)c);
            bf16x8 x[4], e[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                x[k] = *reinterpret_cast<const bf16x8*>(xin + off[k]);
                if (HASADD) e[k] = *reinterpret_cast<const bf16x8*>(add + off[k]);
            }
            __builtin_amdgcn_sched_barrier(0);
            float v[4][8];
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float t = (float)x[k][j];
                    if (INMODE == PA_LD_BNRELU) { const float2 kk = kin[cc + j]; t = fmaxf(fmaf(kk.x, t, kk.y), 0.f); }
                    v[k][j] = t;
                }
            int arg[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                int a = 0; float m = v[0][j];
#pragma unroll
                for (int k = 1; k < 4; ++k) if (v[k][j] > m) { m = v[k][j]; a = k; }
                arg[j] = a;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float rr[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) rr[j] = (arg[j] == k ? (float)g[j] : 0.f) + (HASADD ? (float)e[k][j] : 0.f);
                bf16x8 o;
                if (EPMODE == PA_OUT_BWD) {
                    o = Bwd8::apply(cst, cc, x[k], rr, s1, s2);      // the tensor being differentiated IS the pooled tensor (launcher checks)
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] = (bf16)rr[j];
                }
                *reinterpret_cast<bf16x8*>(din + off[k]) = o;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    if (EPMODE == PA_OUT_BWD) flush_stats(red, ep.stats, C, c, s1, s2);
}

template <int INMODE, bool HASADD>
static void launch_maxpool_bwd_s(const bf16* dout, const PaOperand& in, const bf16* add, const PaEpilogue& ep, bf16* din, int rows, int W, int C,
                                 int blocks, int threads, hipStream_t st) {
    const size_t lds = (threads / 64) * 2 * C * sizeof(float) + C * sizeof(float4) + C * sizeof(float2);
    if (ep.mode == PA_OUT_BWD)
        hipLaunchKernelGGL((maxpool_bwd_s_kernel<INMODE, HASADD, PA_OUT_BWD>), dim3(blocks), dim3(threads), lds, st, dout, in, add, ep, din, rows, W, C);
    else
        hipLaunchKernelGGL((maxpool_bwd_s_kernel<INMODE, HASADD, PA_OUT_PLAIN>), dim3(blocks), dim3(threads), lds, st, dout, in, add, ep, din, rows, W, C);
}

// the streaming variants need blockDim and the items of a map row (W/2 * C/8) to divide one another: pick 512 / 256 threads
// where 1024 do not (96 x 96 maps of the 384 x 384 configuration: 1536 items per row)
static inline void fit_threads_to_rows(int& threads, int W, int C) {
    const size_t row_items = (size_t)(W / 2) * (C / 8);
    auto fit = [&](int t) { return t % (C / 8) == 0 && (row_items % t == 0 || t % row_items == 0); };
    if (fit(threads)) return;
    // (384 / 192 / 128: the 24 x 24 and 12 x 12 maps of the 384 x 384 configuration -- 12 / 6 low-resolution pixels x 32 channel groups per row)
    for (int t : {512, 256, 384, 192, 128}) if (fit(t)) { threads = t; return; }
}

int pa_launch_maxpool_bwd(const bf16* dout, const PaOperand& in, const PaOperand& add, const PaEpilogue& ep, bf16* din,
                          int B, int H, int W, int C, hipStream_t st, int* stat_rows) {
    size_t total = (size_t)B * (H / 2) * (W / 2) * (C / 8);
    int blocks, threads;
    stream_launch_dims(total, blocks, threads);
    fit_threads_to_rows(threads, W, C);
    if (stat_rows) *stat_rows = blocks;
    if (ep.rows_out) *ep.rows_out = blocks;
    {
        static int old = -1;
        if (old < 0) old = pa_getenv("PA_ELTWISE_OLD") ? 1 : 0;
        const size_t row_items = (size_t)(W / 2) * (C / 8);
        const bool in_ok = in.mode == PA_LD_PLAIN || in.mode == PA_LD_BNRELU;
        const bool add_ok = add.mode == PA_LD_NONE || add.mode == PA_LD_PLAIN;
        // BatchNorm-backward epilogue: only when it refers to the pooled tensor itself (its raw values are already in registers)
        const bool ep_ok = ep.mode == PA_OUT_PLAIN || (ep.mode == PA_OUT_BWD && ep.xref == in.p);
        if (!old && in_ok && add_ok && ep_ok && threads % (C / 8) == 0 && (row_items % threads == 0 || threads % row_items == 0) &&
            (size_t)B * H * W * C < ((size_t)1 << 31)) {
            const int rows = B * (H / 2);
            const bool has_add = add.mode == PA_LD_PLAIN;
            if (in.mode == PA_LD_BNRELU) {
                if (has_add) launch_maxpool_bwd_s<PA_LD_BNRELU, true>(dout, in, add.p, ep, din, rows, W, C, blocks, threads, st);
                else launch_maxpool_bwd_s<PA_LD_BNRELU, false>(dout, in, nullptr, ep, din, rows, W, C, blocks, threads, st);
            } else {
                if (has_add) launch_maxpool_bwd_s<PA_LD_PLAIN, true>(dout, in, add.p, ep, din, rows, W, C, blocks, threads, st);
                else launch_maxpool_bwd_s<PA_LD_PLAIN, false>(dout, in, nullptr, ep, din, rows, W, C, blocks, threads, st);
            }
            return (int)hipGetLastError();
        }
    }
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(blocks), dim3(threads), (threads / 64) * 2 * C * sizeof(float), st, dout, in, add, ep, din, B, H, W, C);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// out = nearest_upsample_x2(low) + skip   (reference models/asn_stacked_hg.py:193-203)
__global__ void upadd_fwd_kernel(PaOperand low, PaOperand skip, bf16* out, int B, int H, int W, int C) {
    const int CG = C / 8, Hl = H / 2, Wl = W / 2;
    const size_t total = (size_t)B * H * W * CG;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        int cg = (int)(t % CG);
        size_t p = t / CG;
        int x = (int)(p % W);
        size_t q = p / W;
        int y = (int)(q % H), b = (int)(q / H);
        int c = cg * 8;
        float a[8], s[8];
        load8_rt(low, (((size_t)b * Hl + (y >> 1)) * Wl + (x >> 1)) * C + c, c, a);
        load8_rt(skip, p * C + c, c, s);
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (bf16)(a[j] + s[j]);
        *reinterpret_cast<bf16x8*>(out + p * C + c) = o;
    }
}

int pa_launch_upadd_fwd(const PaOperand& low, const PaOperand& skip, bf16* out, int B, int H, int W, int C, hipStream_t st) {
    if (launch_pool_up_fwd_s<1>(low, skip, out, B, H, W, C, st)) return (int)hipGetLastError();
    size_t total = (size_t)B * H * W * (C / 8);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(upadd_fwd_kernel, dim3(blocks), dim3(256), 0, st, low, skip, out, B, H, W, C);
    return (int)hipGetLastError();
}

// backward: dskip = epilogue_skip(dout), dlow = epilogue_low(sum of the 2x2 dout block)
__global__ __launch_bounds__(1024) void upadd_bwd_kernel(const bf16* dout, PaEpilogue epl, bf16* dlow, PaEpilogue eps, bf16* dskip,
                                 int B, int H, int W, int C) {
    extern __shared__ float red[];     // [nwaves][2*C]
    const int CG = C / 8, Hl = H / 2, Wl = W / 2;
    float l1[8], l2[8], k1[8], k2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { l1[j] = l2[j] = k1[j] = k2[j] = 0.f; }
    const int cg = threadIdx.x % CG, c = cg * 8;
    const size_t total = (size_t)B * Hl * Wl * CG;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        size_t p = t / CG;
        int xl = (int)(p % Wl);
        size_t q = p / Wl;
        int yl = (int)(q % Hl), b = (int)(q / Hl);
        float sum[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) sum[j] = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            size_t idx = (((size_t)b * H + 2 * yl + (k >> 1)) * W + 2 * xl + (k & 1)) * C + c;
            bf16x8 g = *reinterpret_cast<const bf16x8*>(dout + idx);
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { v[j] = (float)g[j]; sum[j] += v[j]; }
            *reinterpret_cast<bf16x8*>(dskip + idx) = epilogue8(eps, idx, c, v, k1, k2);
        }
        size_t li = p * C + c;
        *reinterpret_cast<bf16x8*>(dlow + li) = epilogue8(epl, li, c, sum, l1, l2);
    }
    if (epl.mode != PA_OUT_PLAIN) flush_stats(red, epl.stats, C, c, l1, l2);
    if (eps.mode != PA_OUT_PLAIN) flush_stats(red, eps.stats, C, c, k1, k2);
}


// ---- streaming variants for the hot configurations (both epilogues BatchNorm-backward).  Measured COLD (operands not in
// the Infinity Cache, tools/bench_cold.py) the generic kernels above move 1.7-2.1 TB/s: runtime mode branches, 56 scratch
// accesses and per-use loads of the per-channel constants serialise their loads.  Here the modes are template parameters,
// the per-channel constants sit in LDS as float4 {scale, shift, mean, invstd}, and ALL loads of an item (9 x 16 bytes) are
// issued before the first use, so that a 1024-thread workgroup keeps ~150 KB in flight.
// LOW / SKIP: which of the two outputs this launch produces.  Both: the one-launch form.  One each: the hourglass' backward pass needs dlow on
// its main chain (the low-resolution path below) and dskip only on the side stream that runs the skip block's backward pass -- the main
// chain then waits for 75 MB of traffic instead of 175 (round 6)
template <bool LOW, bool SKIP>
__global__ __launch_bounds__(1024) void upadd_bwd_bb_kernel(const bf16* __restrict__ dout, PaEpilogue epl, bf16* __restrict__ dlow,
                                                            PaEpilogue eps, bf16* __restrict__ dskip, int rows, int W, int C) {
    PA_SET_ELT_PRIO();
    // rows = B * H/2 low-resolution rows; row r covers high-resolution rows 2r, 2r+1 (H = 2*Hl: no batch/row split needed).
    // blockDim is a multiple of the items of a row (Wl * C/8) or the other way round (launcher): no per-iteration division.
    extern __shared__ float red[];     // [nwaves][2*C] partial statistics, then the two constant tables
    float4* cl = reinterpret_cast<float4*>(red + (blockDim.x / 64) * 2 * C);
    float4* cs = cl + C;
    if (LOW) stage_bwd_consts(cl, epl, C);
    if (SKIP) stage_bwd_consts(cs, eps, C);
    __syncthreads();
    const int CG = C / 8, Wl = W / 2;
    const unsigned row_items = (unsigned)Wl * CG;
    float l1[8], l2[8], k1[8], k2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { l1[j] = l2[j] = k1[j] = k2[j] = 0.f; }
    const int c = (threadIdx.x % CG) * 8;
    const unsigned rows_per_it = row_items >= blockDim.x ? 1u : blockDim.x / row_items;
    const unsigned rsub = row_items >= blockDim.x ? 0u : threadIdx.x / row_items;
    const unsigned i0 = row_items >= blockDim.x ? threadIdx.x : threadIdx.x % row_items;
    const bf16* __restrict__ xsk = eps.xref;
    const bf16* __restrict__ xlo = epl.xref;
    for (unsigned rb = blockIdx.x * rows_per_it; rb < (unsigned)rows; rb += gridDim.x * rows_per_it) {
        const unsigned r = rb + rsub;
        if (r >= (unsigned)rows) continue;
        for (unsigned i = i0; i < row_items; i += blockDim.x) {
            const unsigned xl = i / (unsigned)CG;
            int cc = c;
            asm volatile("" : "+v"(cc));      // opaque per iteration: the constant tables stay in LDS (hoisted they are 64 registers)
            const unsigned top = ((2u * r) * (unsigned)W + 2u * xl) * (unsigned)C + (unsigned)c;
            const unsigned wc = (unsigned)W * (unsigned)C;
            const unsigned off[4] = {top, top + (unsigned)C, top + wc, top + wc + (unsigned)C};
            const unsigned li = (r * (unsigned)Wl + xl) * (unsigned)C + (unsigned)c;
            bf16x8 g[4], xs[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                g[k] = *reinterpret_cast<const bf16x8*>(dout + off[k]);
                if (SKIP) xs[k] = *reinterpret_cast<const bf16x8*>(xsk + off[k]);
            }
            bf16x8 xlow;
            if (LOW) xlow = *reinterpret_cast<const bf16x8*>(xlo + li);
            __builtin_amdgcn_sched_barrier(0);        // all nine loads are issued; the constant reads below stay next to their use
            float sum[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) sum[j] = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) { v[j] = (float)g[k][j]; sum[j] += v[j]; }
                if (SKIP) {
                    *reinterpret_cast<bf16x8*>(dskip + off[k]) = Bwd8::apply(cs, cc, xs[k], v, k1, k2);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (LOW) *reinterpret_cast<bf16x8*>(dlow + li) = Bwd8::apply(cl, cc, xlow, sum, l1, l2);
        }
    }
    if (LOW) flush_stats(red, epl.stats, C, c, l1, l2);
    if (SKIP) flush_stats(red, eps.stats, C, c, k1, k2);
}

bool pa_upadd_bwd_splits(const PaEpilogue& ep_low, const PaEpilogue& ep_skip, int B, int H, int W, int C) {
    size_t total = (size_t)B * (H / 2) * (W / 2) * (C / 8);
    int blocks, threads;
    stream_launch_dims(total, blocks, threads);
    fit_threads_to_rows(threads, W, C);
    const size_t row_items = (size_t)(W / 2) * (C / 8);
    return !pa_getenv("PA_ELTWISE_OLD") && ep_low.mode == PA_OUT_BWD && ep_skip.mode == PA_OUT_BWD && threads % (C / 8) == 0 &&
           (row_items % threads == 0 || threads % row_items == 0) && (size_t)B * H * W * C < ((size_t)1 << 31);
}

int pa_launch_upadd_bwd(const bf16* dout, const PaEpilogue& ep_low, bf16* dlow, const PaEpilogue& ep_skip, bf16* dskip,
                        int B, int H, int W, int C, hipStream_t st, int* stat_rows, int part) {
    size_t total = (size_t)B * (H / 2) * (W / 2) * (C / 8);
    int blocks, threads;
    stream_launch_dims(total, blocks, threads);
    fit_threads_to_rows(threads, W, C);
    if (stat_rows) *stat_rows = blocks;
    if (ep_low.rows_out && (part & 1)) *ep_low.rows_out = blocks;
    if (ep_skip.rows_out && (part & 2)) *ep_skip.rows_out = blocks;
    if (part != 3) {           // one output only: the streaming kernel or nothing (callers ask pa_upadd_bwd_splits first)
        if (!pa_upadd_bwd_splits(ep_low, ep_skip, B, H, W, C) || (part != 1 && part != 2)) { pa_set_error_msg("pa_launch_upadd_bwd: this launch cannot be split"); return 1; }
        const size_t lds = (threads / 64) * 2 * C * sizeof(float) + 2 * C * sizeof(float4);
        if (part == 1) hipLaunchKernelGGL((upadd_bwd_bb_kernel<true, false>), dim3(blocks), dim3(threads), lds, st, dout, ep_low, dlow, ep_skip, dskip, B * (H / 2), W, C);
        else hipLaunchKernelGGL((upadd_bwd_bb_kernel<false, true>), dim3(blocks), dim3(threads), lds, st, dout, ep_low, dlow, ep_skip, dskip, B * (H / 2), W, C);
        return (int)hipGetLastError();
    }
    static int old = -1;
    if (old < 0) old = pa_getenv("PA_ELTWISE_OLD") ? 1 : 0;
    const size_t row_items = (size_t)(W / 2) * (C / 8);
    if (!old && ep_low.mode == PA_OUT_BWD && ep_skip.mode == PA_OUT_BWD && threads % (C / 8) == 0 &&
        (row_items % threads == 0 || threads % row_items == 0) && (size_t)B * H * W * C < ((size_t)1 << 31)) {
        hipLaunchKernelGGL((upadd_bwd_bb_kernel<true, true>), dim3(blocks), dim3(threads), (threads / 64) * 2 * C * sizeof(float) + 2 * C * sizeof(float4), st,
                           dout, ep_low, dlow, ep_skip, dskip, B * (H / 2), W, C);
        return (int)hipGetLastError();
    }
    hipLaunchKernelGGL(upadd_bwd_kernel, dim3(blocks), dim3(threads), (threads / 64) * 2 * C * sizeof(float), st, dout, ep_low, dlow, ep_skip, dskip,
                       B, H, W, C);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// RMSprop exactly as torch.optim.RMSprop(alpha, eps, momentum=0, weight_decay=0, centered=False)
// (reference stack-hg.py:51-52): v = alpha v + (1-alpha) g^2 ; p -= lr g / (sqrt(v) + eps).
// gscale multiplies the gradient first (1/world_size after a sum all-reduce).
__global__ void rmsprop_kernel(float* p, const float* g, float* v, size_t n, float lr, float alpha, float eps, float gscale) {
    size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const size_t stride = (size_t)gridDim.x * blockDim.x * 4;
    for (; i + 3 < n; i += stride) {
        f32x4 gg = *reinterpret_cast<const f32x4*>(g + i), vv = *reinterpret_cast<f32x4*>(v + i), pp = *reinterpret_cast<f32x4*>(p + i);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float gr = gg[j] * gscale;
            vv[j] = alpha * vv[j] + (1.f - alpha) * gr * gr;
            pp[j] = pp[j] - lr * (gr / (sqrtf(vv[j]) + eps));
        }
        *reinterpret_cast<f32x4*>(v + i) = vv;
        *reinterpret_cast<f32x4*>(p + i) = pp;
    }
    for (; i < n && i + 3 >= n; ++i) {          // tail (at most 3 elements, one thread)
        float gr = g[i] * gscale;
        v[i] = alpha * v[i] + (1.f - alpha) * gr * gr;
        p[i] = p[i] - lr * (gr / (sqrtf(v[i]) + eps));
    }
}

// Half-precision build only (gradients travel multiplied by PA_GRAD_SCALE, common.h): an overflow anywhere in the backward
// pass shows up as inf / NaN in the flat gradient.  A step whose gradient holds a non-finite value is SKIPPED (parameters and
// square_avg untouched, the loss-scaling convention) and counted; pa_rmsprop_skipped_steps reads the counter.
// The two state words belong to ONE optimizer: the caller passes its own device pair (pa_rmsprop_step_state), so that two optimizers
// stepping on different streams (pose net and agent in the joint stage) cannot see each other's flag; the process-wide pair below
// only serves the stateless entry point pa_rmsprop_step.
__device__ int g_grad_state[2];            // [0] this step's gradient holds a non-finite value, [1] steps skipped so far

__global__ void grad_check_reset_kernel(int* state) { (state ? state : g_grad_state)[0] = 0; }

__global__ void grad_check_kernel(const float* g, size_t n, int* state) {
    size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const size_t stride = (size_t)gridDim.x * blockDim.x * 4;
    bool bad = false;
    for (; i + 3 < n; i += stride) {
        const f32x4 gg = *reinterpret_cast<const f32x4*>(g + i);
        bad |= !(isfinite(gg[0]) && isfinite(gg[1]) && isfinite(gg[2]) && isfinite(gg[3]));
    }
    for (; i < n && i + 3 >= n; ++i) bad |= !isfinite(g[i]);
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(&(state ? state : g_grad_state)[0], 1);
}

__global__ void rmsprop_checked_kernel(float* p, const float* g, float* v, size_t n, float lr, float alpha, float eps, float gscale, int* state) {
    int* gs = state ? state : g_grad_state;
    if (gs[0]) {                            // uniform over the grid: every thread reads the same word
        if (blockIdx.x == 0 && threadIdx.x == 0) gs[1] += 1;
        return;
    }
    size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const size_t stride = (size_t)gridDim.x * blockDim.x * 4;
    for (; i + 3 < n; i += stride) {
        f32x4 gg = *reinterpret_cast<const f32x4*>(g + i), vv = *reinterpret_cast<f32x4*>(v + i), pp = *reinterpret_cast<f32x4*>(p + i);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float gr = gg[j] * gscale;
            vv[j] = alpha * vv[j] + (1.f - alpha) * gr * gr;
            pp[j] = pp[j] - lr * (gr / (sqrtf(vv[j]) + eps));
        }
        *reinterpret_cast<f32x4*>(v + i) = vv;
        *reinterpret_cast<f32x4*>(p + i) = pp;
    }
    for (; i < n && i + 3 >= n; ++i) {
        float gr = g[i] * gscale;
        v[i] = alpha * v[i] + (1.f - alpha) * gr * gr;
        p[i] = p[i] - lr * (gr / (sqrtf(v[i]) + eps));
    }
}

int pa_launch_rmsprop(float* p, const float* g, float* v, size_t n, float lr, float alpha, float eps, float gscale, int* state, hipStream_t st) {
    if (n == 0) return 0;
    int blocks = (int)((n / 4 + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
#ifdef PA_FP16
    hipLaunchKernelGGL(grad_check_reset_kernel, dim3(1), dim3(1), 0, st, state);
    hipLaunchKernelGGL(grad_check_kernel, dim3(blocks), dim3(256), 0, st, g, n, state);
    hipLaunchKernelGGL(rmsprop_checked_kernel, dim3(blocks), dim3(256), 0, st, p, g, v, n, lr, alpha, eps, gscale, state);
#else
    (void)state;
    hipLaunchKernelGGL(rmsprop_kernel, dim3(blocks), dim3(256), 0, st, p, g, v, n, lr, alpha, eps, gscale);
#endif
    return (int)hipGetLastError();
}

// plain streaming copy, 16 bytes per lane, 4 independent loads per thread and iteration: what "just moving the bytes" reaches on this
// part at a given size (bench.py roofline.floor calibrates with it instead of a framework kernel)
__global__ __launch_bounds__(256) void copy16_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        const uint4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
    }
    for (; i < n16; i += stride) dst[i] = src[i];
}

// form 1: ONE 16-byte chunk per thread, no loop (the guide's "float4 copy"); form 2: the grid-stride form with non-temporal loads / stores
__global__ __launch_bounds__(256) void copy16_flat_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n16) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void copy16_nt_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    const u4* s = reinterpret_cast<const u4*>(src); u4* d = reinterpret_cast<u4*>(dst);
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        const u4 a = __builtin_nontemporal_load(s + i), b = __builtin_nontemporal_load(s + i + stride), c = __builtin_nontemporal_load(s + i + 2 * stride), e = __builtin_nontemporal_load(s + i + 3 * stride);
        __builtin_nontemporal_store(a, d + i); __builtin_nontemporal_store(b, d + i + stride); __builtin_nontemporal_store(c, d + i + 2 * stride); __builtin_nontemporal_store(e, d + i + 3 * stride);
    }
    for (; i < n16; i += stride) __builtin_nontemporal_store(__builtin_nontemporal_load(s + i), d + i);
}

int pa_launch_copy16(void* dst, const void* src, size_t bytes, hipStream_t st, int form) {
    const size_t n16 = bytes / 16;
    if (n16 == 0) return 0;
    if (form == 1) {
        hipLaunchKernelGGL(copy16_flat_kernel, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, st, reinterpret_cast<const uint4*>(src), reinterpret_cast<uint4*>(dst), n16);
        return (int)hipGetLastError();
    }
    if (form == 2) {
        size_t nb = (n16 + 1023) / 1024;
        if (nb > 256 * 16) nb = 256 * 16;
        hipLaunchKernelGGL(copy16_nt_kernel, dim3((unsigned)nb), dim3(256), 0, st, reinterpret_cast<const uint4*>(src), reinterpret_cast<uint4*>(dst), n16);
        return (int)hipGetLastError();
    }
    size_t blocks = (n16 + 1023) / 1024;                 // 4 chunks per thread
    if (blocks > 256 * 16) blocks = 256 * 16;            // 16 workgroups per CU, grid-stride beyond
    hipLaunchKernelGGL(copy16_kernel, dim3((unsigned)blocks), dim3(256), 0, st, reinterpret_cast<const uint4*>(src), reinterpret_cast<uint4*>(dst), n16);
    return (int)hipGetLastError();
}

int pa_rmsprop_skipped(const int* state, long long* out, hipStream_t st) {
    int h[2] = {0, 0};
#ifdef PA_FP16
    hipError_t e = state ? hipMemcpyAsync(h, state, sizeof h, hipMemcpyDeviceToHost, st)
                         : hipMemcpyFromSymbolAsync(h, HIP_SYMBOL(g_grad_state), sizeof h, 0, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return (int)e;
#endif
    *out = h[1];
    return 0;
}

// ------------------------------------------------------------------------------------------------
// fp32 master weights (PyTorch layout [Cout][Cin][taps]) -> bf16 compute copies
//   wf[n][tap][c]            forward / wgrad layout, zero padded to pad_cout x pad_cin
//   wb[c][taps-1-tap][n]     dgrad layout (transposed, taps flipped)
// taps == 49 marks the 7x7 stem: wf[n][ky*32 + kx*4 + c] (K padded to 256), no wb.
__global__ __launch_bounds__(1024) void weight_prep_kernel(const PaPrepJob* jobs) {
    const PaPrepJob j = jobs[blockIdx.y];
    if (j.taps == 49) {
        const int total = j.pad_cout * 256;
        for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
            int n = e >> 8, k = e & 255;
            int ky = k >> 5, kx = (k & 31) >> 2, c = k & 3;
            float v = 0.f;
            if (n < j.Cout && ky < 7 && kx < 7 && c < j.Cin) v = j.w[((size_t)(n * j.Cin + c) * 7 + ky) * 7 + kx];
            j.wf[e] = (bf16)v;
        }
        return;
    }
    // 32 x 32 (n, c) tiles: the fp32 source is read once, in contiguous runs (taps floats per thread); wf is written along c,
    // wb along n through an LDS transpose (read element-wise per output the strided source fetched 10-20x its bytes)
    __shared__ bf16 tile[9][32][33];
    const int tn = threadIdx.x >> 5, tc = threadIdx.x & 31;
    const int tiles_c = j.pad_cin / 32, ntiles = (j.pad_cout / 32) * tiles_c;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int n0 = (t / tiles_c) * 32, c0 = (t % tiles_c) * 32;
        const int n = n0 + tn, c = c0 + tc;
        const bool ok = n < j.Cout && c < j.Cin;
        const float* src = j.w + ((size_t)n * j.Cin + c) * j.taps;
        for (int tap = 0; tap < j.taps; ++tap) {
            const bf16 v = (bf16)(ok ? src[tap] : 0.f);
            j.wf[((size_t)n * j.taps + tap) * j.pad_cin + c] = v;
            tile[tap][tn][tc] = v;
        }
        __syncthreads();
        if (j.wb) {
            // thread (tn, tc) now writes channel c0 + tn, output channel n0 + tc
            for (int tap = 0; tap < j.taps; ++tap)
                j.wb[((size_t)(c0 + tn) * j.taps + (j.taps - 1 - tap)) * j.pad_cout + n0 + tc] = tile[tap][tc][tn];
        }
        __syncthreads();
    }
}

int pa_launch_weight_prep(const PaPrepJob* jobs_dev, int njobs, int max_elems, hipStream_t st) {
    if (njobs <= 0) return 0;
    (void)max_elems;
    hipLaunchKernelGGL(weight_prep_kernel, dim3(32, njobs), dim3(1024), 0, st, jobs_dev);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// layout conversion
__global__ void nchw_to_nhwc4_kernel(const float* src, bf16* dst, int B, int H, int W) {
    const size_t total = (size_t)B * H * W;
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (size_t)gridDim.x * blockDim.x) {
        size_t hw = (size_t)H * W;
        size_t b = p / hw, r = p - b * hw;
        bf16x4 o;
        o[0] = (bf16)src[(b * 3 + 0) * hw + r];
        o[1] = (bf16)src[(b * 3 + 1) * hw + r];
        o[2] = (bf16)src[(b * 3 + 2) * hw + r];
        o[3] = (bf16)0.f;
        *reinterpret_cast<bf16x4*>(dst + p * 4) = o;
    }
}

int pa_launch_nchw_to_nhwc4(const float* src, bf16* dst, int B, int H, int W, hipStream_t st) {
    size_t total = (size_t)B * H * W;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(nchw_to_nhwc4_kernel, dim3(blocks), dim3(256), 0, st, src, dst, B, H, W);
    return (int)hipGetLastError();
}

__global__ void nhwc_to_nchw_f32_kernel(const float* src, float* dst, int B, int H, int W, int C) {
    const size_t total = (size_t)B * C * H * W;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        size_t hw = (size_t)H * W;
        size_t b = e / (C * hw), r = e - b * C * hw;
        size_t c = r / hw, p = r - c * hw;
        dst[e] = src[(b * hw + p) * C + c];
    }
}

int pa_launch_nhwc_to_nchw_f32(const float* src, float* dst, int B, int H, int W, int C, hipStream_t st) {
    size_t total = (size_t)B * H * W * C;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(nhwc_to_nchw_f32_kernel, dim3(blocks), dim3(256), 0, st, src, dst, B, H, W, C);
    return (int)hipGetLastError();
}

__global__ void nchw_f32_to_nhwc_bf16_kernel(const float* src, bf16* dst, int B, int C, int H, int W) {
    const size_t total = (size_t)B * C * H * W;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        size_t hw = (size_t)H * W;
        size_t c = e % C, p = e / C;           // destination order
        size_t b = p / hw, r = p - b * hw;
        dst[e] = (bf16)src[(b * C + c) * hw + r];
    }
}

int pa_launch_nchw_f32_to_nhwc_bf16(const float* src, bf16* dst, int B, int C, int H, int W, hipStream_t st) {
    size_t total = (size_t)B * H * W * C;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(nchw_f32_to_nhwc_bf16_kernel, dim3(blocks), dim3(256), 0, st, src, dst, B, C, H, W);
    return (int)hipGetLastError();
}

__global__ void nhwc_bf16_to_nchw_f32_kernel(PaOperand src, float* dst, int B, int C, int H, int W) {
    const size_t total = (size_t)B * C * H * W;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        size_t hw = (size_t)H * W;
        size_t c = e % C, p = e / C;           // source order
        size_t b = p / hw, r = p - b * hw;
        float v = (float)src.p[e];
        if (src.mode == PA_LD_BNRELU) v = fmaxf(fmaf(src.k0[c], v, src.k1[c]), 0.f);
        else if (src.mode == PA_LD_LIN2) v = fmaf(src.k0[c], v, fmaf(src.k1[c], (float)src.q[e], src.k2[c]));
        dst[(b * C + c) * hw + r] = v;
    }
}

int pa_launch_nhwc_bf16_to_nchw_f32(const PaOperand& src, float* dst, int B, int C, int H, int W, hipStream_t st) {
    size_t total = (size_t)B * H * W * C;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(nhwc_bf16_to_nchw_f32_kernel, dim3(blocks), dim3(256), 0, st, src, dst, B, C, H, W);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// apply an epilogue to a gradient tensor (e.g. mask + BatchNorm-backward reductions of a gradient
// that arrives from outside the conv kernels)
__global__ __launch_bounds__(1024) void ep_apply_kernel(PaOperand g, PaEpilogue ep, bf16* out, size_t M, int C) {
    extern __shared__ float red[];
    const int CG = C / 8;
    float s1[8], s2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
    const int c = (threadIdx.x % CG) * 8;
    const size_t total = M * CG;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const size_t idx = (t / CG) * C + c;
        float v[8];
        load8_rt(g, idx, c, v);
        *reinterpret_cast<bf16x8*>(out + idx) = epilogue8(ep, idx, c, v, s1, s2);
    }
    if (ep.mode != PA_OUT_PLAIN) flush_stats(red, ep.stats, C, c, s1, s2);
}

int pa_launch_ep_apply(const PaOperand& g, const PaEpilogue& ep, bf16* out, size_t M, int C, hipStream_t st, int* stat_rows) {
    size_t total = M * (C / 8);
    int blocks, threads;
    stream_launch_dims(total, blocks, threads);
    if (stat_rows) *stat_rows = blocks;
    if (ep.rows_out) *ep.rows_out = blocks;
    hipLaunchKernelGGL(ep_apply_kernel, dim3(blocks), dim3(threads), (threads / 64) * 2 * C * sizeof(float), st, g, ep, out, M, C);
    return (int)hipGetLastError();
}

// Occlusion mask of the reference's _Hourglass._dropout (models/asn_stacked_hg.py:79-100): out = ep(value(x) * m),
// m = mask[b][4*(y / (H/4)) + x / (W/4)] -- the n x 1 x 4 x 4 cell mask, nearest-upsampled to the H x W map, times every
// channel.  Forward: ep = plain.  Backward: x = the gradient of the masked tensor, ep = the epilogue that finishes the
// gradient of the tensor that was masked (ReLU mask + BatchNorm-backward reductions).
__global__ __launch_bounds__(1024) void cell_mask_kernel(PaOperand x, const float* mask, PaEpilogue ep, bf16* out, int B, int H, int W, int C) {
    extern __shared__ float red[];
    const int CG = C / 8;
    float s1[8], s2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
    const int c = (threadIdx.x % CG) * 8;
    const size_t total = (size_t)B * H * W * CG;
    const int ch = H / 4, cw = W / 4;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const size_t pix = t / CG;
        const size_t idx = pix * C + c;
        const int xw = (int)(pix % W);
        const size_t q = pix / W;
        const int y = (int)(q % H), b = (int)(q / H);
        const float m = mask[(size_t)b * 16 + 4 * (y / ch) + xw / cw];
        float v[8];
        load8_rt(x, idx, c, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] *= m;
        *reinterpret_cast<bf16x8*>(out + idx) = epilogue8(ep, idx, c, v, s1, s2);
    }
    if (ep.mode != PA_OUT_PLAIN) flush_stats(red, ep.stats, C, c, s1, s2);
}

int pa_launch_cell_mask(const PaOperand& x, const float* mask, const PaEpilogue& ep, bf16* out, int B, int H, int W, int C,
                        hipStream_t st) {
    if (H % 4 != 0 || W % 4 != 0 || C % 8 != 0) { pa_set_error_msg("pa_launch_cell_mask: the map must be a multiple of the 4x4 cell grid"); return 1; }
    size_t total = (size_t)B * H * W * (C / 8);
    int blocks, threads;
    stream_launch_dims(total, blocks, threads);
    if (ep.rows_out) *ep.rows_out = blocks;
    hipLaunchKernelGGL(cell_mask_kernel, dim3(blocks), dim3(threads), (threads / 64) * 2 * C * sizeof(float), st, x, mask, ep, out, B, H, W, C);
    return (int)hipGetLastError();
}

// end of a pose forward pass: hand the per-stack losses out (keep: the engine's own copy [n + 1]; out: the caller's [n], may be NULL), their
// sum (keep[n], *total -- the reference's `loss = sum over stacks`, stack-hg.py:156-159, without a framework reduction kernel between the
// backward pass and the optimizer) and clear the accumulators for the next pass (was a memset launch in front of every step)
__global__ void loss_out_kernel(float* acc, float* keep, float* out, float* total, int n) {
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int i = 0; i < n; ++i) { const float v = acc[i]; s += v; keep[i] = v; if (out) out[i] = v; acc[i] = 0.f; }
        keep[n] = s;
        if (total) *total = s;
    }
}

int pa_launch_loss_out(float* acc, float* keep, float* out, float* total, int n, hipStream_t st) {
    hipLaunchKernelGGL(loss_out_kernel, dim3(1), dim3(64), 0, st, acc, keep, out, total, n);
    return (int)hipGetLastError();
}

__global__ void fill_kernel(float* p, float v, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

int pa_launch_fill(float* p, float v, size_t n, hipStream_t st) {
    int blocks = (int)((n + 255) / 256);
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(fill_kernel, dim3(blocks), dim3(256), 0, st, p, v, n);
    return (int)hipGetLastError();
}
