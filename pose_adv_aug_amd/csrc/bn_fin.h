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
// s_i = r_i + r_(i+2), total = s_0 + s_1.  Sixteen independent chains put every load of a thread in flight at once (the finalize is
// one L2 round trip, not rows / 8 of them), and the tree lets G = 1, 2, 4 or 8 threads share a channel pair -- each sums the chains
// congruent to its index mod G and runs the tree down to stride G locally.  The prologue and the launch give the same bits, so the
// choice between them (Net::fin_rows_max, PA_FIN_PROLOGUE in tuning builds) never changes a result.
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

// chains j0, j0 + G, ... (16 / G of them) of the channel PAIR (c2, c2 + 1), tree run down to stride G: the thread's partial of the pair
// as {sum[c2], sq[c2], sum[c2+1], sq[c2+1]}.  All (16 / G) x 8 row loads are independent 16-byte loads.
template <int G>
__device__ __forceinline__ f32x4 pa_fin_partial(const float* stats, int rows, int C, int c2, int j0) {
    constexpr int NC = 16 / G;
    f32x4 p[NC];
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        f32x4 v[8];
        const int j = j0 + i * G;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int r = j + 16 * u;
            const int rc = r < rows ? r : 0;                       // clamped, unconditional: every load of the thread in flight
            v[u] = *reinterpret_cast<const f32x4*>(stats + ((size_t)rc * C + c2) * 2);
        }
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += (j + 16 * u < rows) ? v[u] : zero;      // (rows beyond the end add +0: straight-line code, same bits in both forms)
        p[i] = acc;
    }
    // stride-halving tree over this thread's chains: chain index j0 + i * G pairs with (j0 + i * G) + 8, + 4, ... while the stride >= G
#pragma unroll
    for (int n = NC; n > 1; n >>= 1)
#pragma unroll
        for (int i = 0; i < n / 2; ++i) p[i] = p[i] + p[i + n / 2];
    return p[0];
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

// Consumer prologue: all NT threads of the workgroup; C <= 2 * NT channels, C % 2 == 0.  k (LDS, 3 * KS floats: k[c], k[KS + c], k[2 * KS + c])
// receives {scale, shift} or {kA, kB, kC}; scratch = 4 * NT floats of LDS.  `writer`: this workgroup stores the results to global memory.
// Ends with a barrier: k is complete for every thread when it returns.
template <int NT, int KS>
__device__ __forceinline__ void pa_bn_fin_prologue(const PaBnFin& f, int C, float* k, float* scratch, bool writer) {
    const int tid = threadIdx.x;
    const int CP = C >> 1;                                    // channel pairs
    const int tpp = NT / CP;                                  // threads per pair available
    const int G = tpp >= 8 ? 8 : (tpp >= 4 ? 4 : (tpp >= 2 ? 2 : 1));
    const int pr = tid % CP, g = tid / CP;
    f32x4* sc = reinterpret_cast<f32x4*>(scratch);            // [G][CP]
    if (g < G) {
        f32x4 s;
        if (G == 8) s = pa_fin_partial<8>(f.stats, f.rows, C, 2 * pr, g);
        else if (G == 4) s = pa_fin_partial<4>(f.stats, f.rows, C, 2 * pr, g);
        else if (G == 2) s = pa_fin_partial<2>(f.stats, f.rows, C, 2 * pr, g);
        else s = pa_fin_partial<1>(f.stats, f.rows, C, 2 * pr, 0);
        sc[g * CP + pr] = s;
    }
    __syncthreads();
    if (tid < C) {
        const int c = tid, half = c & 1;
        // the remaining tree levels (strides G/2 ... 1) over the G partials of the channel's pair
        float a[8], b[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const f32x4 v = sc[(i < G ? i : 0) * CP + (c >> 1)];
            a[i] = half ? v[2] : v[0]; b[i] = half ? v[3] : v[1];
        }
        if (G >= 8) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { a[i] += a[i + 4]; b[i] += b[i + 4]; }
        }
        if (G >= 4) {
#pragma unroll
            for (int i = 0; i < 2; ++i) { a[i] += a[i + 2]; b[i] += b[i + 2]; }
        }
        if (G >= 2) { a[0] += a[1]; b[0] += b[1]; }
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
