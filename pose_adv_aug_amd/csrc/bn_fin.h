// BatchNorm finalize shared by the finalize kernels (elementwise.hip) and by the CONSUMER-PROLOGUE form of it
// (conv_igemm.hip, conv3x3_tile.hip): reference semantics torch.nn.BatchNorm2d in training mode
// (models/asn_stacked_hg.py:19,22,25), eps 1e-5, momentum 0.1, biased variance for the normalisation, unbiased for the
// running estimate; backward dx = kA*dz + kB*x + kC.
//
// Why a prologue form: at 16 x 16 and below a convolution is ~6 us of launch / staging / epilogue latency and the finalize
// launch between two of them costs a dependent kernel boundary of its own (4.4 us + gap) for a few hundred additions.  When the
// producer left at most PA_FIN_SMALL_ROWS partial rows, every workgroup of the CONSUMER sums them itself while its weight slices
// are already in flight (<= 98 KB from L2 per workgroup, no atomics, no flag: the kernel boundary between producer and consumer
// is the only synchronisation), keeps the constants in LDS, and workgroup (0, 0) alone stores them (and the running
// estimates / dgamma, dbeta) for the later readers, which are all behind another kernel boundary.
//
// ONE summation order for <= PA_FIN_SMALL_ROWS rows, used by BOTH forms: sixteen interleaved chains p_j = rows j, j+16, j+32, ...
// added in increasing order (<= 8 rows each), combined by a stride-halving tree: q_i = p_i + p_(i+8), r_i = q_i + q_(i+4),
// s_i = r_i + r_(i+2), total = s_0 + s_1.  Sixteen independent chains keep many row loads in flight (the finalize is two L2 round
// trips, not rows / 8 of them).  FOUR threads share a channel pair: thread g sums the chains g, g+4, g+8, g+12 and runs the first two
// tree levels locally, the last two go through LDS.  The prologue and the launch give the same bits, so the choice between them
// (Net::fin_rows_max, PA_FIN_PROLOGUE in tuning builds) never changes a result.
#pragma once
#include "common.h"

#define PA_FIN_SMALL_ROWS 128

// a pending finalize folded into the consumer (rows == 0: none; the operand's k0 / k1 / k2 pointers are used as they are)
struct PaBnFin {
    const float* stats;        // partial rows [rows][C][2]: forward {sum, sum of squares}; backward {sum dz, sum dz * xhat}
    int rows;
    int bwd;                   // 0: scale / shift for a BNRELU operand; 1: kA / kB / kC for a LIN2 operand
    float count;               // elements per channel (B * H * W)
    // forward: inputs gamma, beta; outputs scale, shift, mean, invstd (+ running estimates)
    // backward: inputs scale, mean, invstd; outputs kA, kB, kC, dgamma, dbeta
    const float* gamma; const float* beta;
    float *rmean, *rvar;
    float *scale, *shift, *mean, *invstd;
    float *kA, *kB, *kC, *dgamma, *dbeta;
    float momentum, eps;
    int update_running;
};

// chains g, g + 4, g + 8, g + 12 of the channel PAIR (c2, c2 + 1), first two tree levels applied: the thread's partial of the pair as
// {sum[c2], sq[c2], sum[c2+1], sq[c2+1]}.  Two chains = 16 independent 16-byte loads at a time (64 registers in flight).
__device__ __forceinline__ f32x4 pa_fin_partial4(const float* stats, int rows, int C, int c2, int g) {
    f32x4 p[4];
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    const unsigned rstride = (unsigned)C * 2u;                     // floats per row (32-bit offsets: a statistics buffer is a few MB)
#pragma unroll
    for (int i0 = 0; i0 < 4; i0 += 2) {
        f32x4 v[2][8];
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int r = g + 4 * (i0 + d) + 16 * u;
                const unsigned rc = r < rows ? (unsigned)r : 0u;    // clamped, unconditional: every load of the batch in flight
                v[d][u] = *reinterpret_cast<const f32x4*>(stats + rc * rstride + (unsigned)c2 * 2u);
            }
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            f32x4 acc = zero;
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += (g + 4 * (i0 + d) + 16 * u < rows) ? v[d][u] : zero;      // (rows beyond the end add +0: straight-line code, same bits in both forms)
            p[i0 + d] = acc;
        }
        // (no barrier between the two halves: all 32 loads of the thread in flight at once -- one L2 round trip)
    }
    // chains g, g+4, g+8, g+12: stride 8 first, then stride 4
    return (p[0] + p[2]) + (p[1] + p[3]);
}

// ONE chain (rows j, j + 16, ...: <= 8 loads, one round trip) of the channel pair: the launch form has threads to spare and gives every
// chain its own thread; the sixteen partials go through the same tree (stride 8, 4, 2, 1) in LDS
__device__ __forceinline__ f32x4 pa_fin_chain16(const float* stats, int rows, int C, int c2, int j) {
    f32x4 v[8];
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    const unsigned rstride = (unsigned)C * 2u;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int r = j + 16 * u;
        const unsigned rc = r < rows ? (unsigned)r : 0u;
        v[u] = *reinterpret_cast<const f32x4*>(stats + rc * rstride + (unsigned)c2 * 2u);
    }
    f32x4 acc = zero;
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += (j + 16 * u < rows) ? v[u] : zero;
    return acc;
}

// the arithmetic of the two finalizes, contraction off: the same roundings wherever it is inlined
__device__ __forceinline__ void pa_bn_fwd_consts(float S1, float S2, float count, float eps, float gamma, float beta, float& scale,
                                                 float& shift, float& mu, float& is, float& var) {
#pragma clang fp contract(off)
    mu = S1 / count;
    var = fmaxf(S2 / count - mu * mu, 0.f);
    is = rsqrtf(var + eps);
    scale = gamma * is;
    shift = beta - mu * scale;
}
__device__ __forceinline__ void pa_bn_running(float mu, float var, float count, float momentum, float& rmean, float& rvar) {
#pragma clang fp contract(off)
    const float unb = count > 1.f ? var * count / (count - 1.f) : var;
    rmean = (1.f - momentum) * rmean + momentum * mu;
    rvar = (1.f - momentum) * rvar + momentum * unb;
}
__device__ __forceinline__ void pa_bn_bwd_consts(float S1, float S2, float count, float s, float is, float mu, float& kA, float& kB, float& kC) {
#pragma clang fp contract(off)
    kA = s;
    const float b = -s * is * S2 / count;
    kB = b;
    kC = -s * S1 / count - b * mu;
}

// Consumer prologue: all NT threads of the workgroup; C <= NT channels, C % 2 == 0.  k (LDS, 3 * KS floats: k[c], k[KS + c], k[2 * KS + c])
// receives {scale, shift} or {kA, kB, kC}; scratch = 4 * NT floats of LDS.  `writer`: this workgroup stores the results to global memory.
// Ends with a barrier: k is complete for every thread when it returns.
// Inlined ONLY into kernel instances of their own (template parameter FIN of conv_igemm_kernel / conv3x3_tile_kernel, chosen by the
// launchers when a launch brings a pending finalize): the first version was compiled into every BatchNorm-on-load instance with all
// sixteen chains' loads in flight and cost those kernels 40 - 100 registers, spills and occupancy whether a launch carried a finalize
// or not (+0.15 ms per step, +2 ms at 8 stacks, found by running the round-3 tree on the same box: tools/ab_r3.sh)
template <int NT, int KS>
__device__ __forceinline__ void pa_bn_fin_prologue(const PaBnFin& f, int C, float* k, float* scratch, bool writer) {
    const int tid = threadIdx.x;
    const int CP = C >> 1;                                    // channel pairs; four threads each
    f32x4* sc = reinterpret_cast<f32x4*>(scratch);            // [4][CP]
#pragma unroll 1
    for (int w = tid; w < 4 * CP; w += NT) {                  // (C = 256 at 256 threads: two rounds)
        const int pr = w % CP, g = w / CP;
        sc[g * CP + pr] = pa_fin_partial4(f.stats, f.rows, C, 2 * pr, g);
    }
    __syncthreads();
#pragma unroll 1
    for (int c = tid; c < C; c += NT) {
        const int half = c & 1;
        // the last two tree levels (strides 2, 1) over the four partials of the channel's pair
        float a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x4 v = sc[i * CP + (c >> 1)];
            a[i] = half ? v[2] : v[0]; b[i] = half ? v[3] : v[1];
        }
        a[0] += a[2]; a[1] += a[3]; b[0] += b[2]; b[1] += b[3];
        a[0] += a[1]; b[0] += b[1];
        const f32x2 t = {a[0], b[0]};
        if (!f.bwd) {
            float s, sh, mu, is, var;
            pa_bn_fwd_consts(t[0], t[1], f.count, f.eps, f.gamma[c], f.beta[c], s, sh, mu, is, var);
            k[c] = s; k[KS + c] = sh;
            if (writer) {
                f.scale[c] = s; f.shift[c] = sh; f.mean[c] = mu; f.invstd[c] = is;
                if (f.update_running) {
                    float rm = f.rmean[c], rv = f.rvar[c];
                    pa_bn_running(mu, var, f.count, f.momentum, rm, rv);
                    f.rmean[c] = rm; f.rvar[c] = rv;
                }
            }
        } else {
            float kA, kB, kC;
            pa_bn_bwd_consts(t[0], t[1], f.count, f.scale[c], f.invstd[c], f.mean[c], kA, kB, kC);
            k[c] = kA; k[KS + c] = kB; k[2 * KS + c] = kC;
            if (writer) {
                f.kA[c] = kA; f.kB[c] = kB; f.kC[c] = kC;
                if (f.dgamma) f.dgamma[c] = t[1];
                if (f.dbeta) f.dbeta[c] = t[0];
            }
        }
    }
    __syncthreads();
}
