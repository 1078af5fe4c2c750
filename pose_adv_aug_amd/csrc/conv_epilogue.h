// Epilogue shared by the MFMA convolution kernels (conv_igemm.hip, conv3x3_tile.hip).
//
// Accumulator layout it expects: a 256-thread workgroup as 2 (wm: pixels) x 2 (wn: channels) waves,
// wave tile (16*MI pixels) x (16*NI channels); fragment (ni, mi) is the 16x16 MFMA output whose ROWS
// are output channels and whose COLUMNS (lane & 15) are pixels.  The weight rows were staged permuted
// (pa_weight_row_of_lds_row) so that lane group q = lane >> 4 of a wave owns the 4*NI CONSECUTIVE
// channels  q*4*NI + 4*ni + reg : all epilogue traffic (addends, xref, stores) is 16-byte accesses.
#pragma once
#include "common.h"
#include "kernels.h"

// LDS row lr (0 .. BN-1) of the weight tile holds which output channel (relative to the tile's n0)?
template <int BN, int NI>
__device__ __forceinline__ int pa_weight_row_of_lds_row(int lr) {
    const int wt = lr / (BN / 2), l = lr - wt * (BN / 2);
    const int ni = l >> 4, rr = l & 15;
    return wt * (BN / 2) + (rr >> 2) * (4 * NI) + 4 * ni + (rr & 3);
}

// pix(mi) -> flattened NHWC pixel index of fragment column (lane & 15) of fragment row-block mi, or -1
// `red` = at least 2*BN*2 floats of LDS that are dead by now; stat_row = this workgroup's partial row
template <int BN, int NI, int MI, class PixFn>
__device__ __forceinline__ void pa_conv_epilogue(const PaConvArgs& a, f32x4 (&acc)[NI][MI], int n0, int wm, int wn,
                                                 PixFn pix, float* red, int stat_row) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int N = a.Cout;
    constexpr int CH = NI / 2;                       // 8-channel chunks per lane
    const int nb = n0 + wn * (BN / 2) + (lane >> 4) * (4 * NI);
    float s1[NI][4], s2[NI][4];
#pragma unroll
    for (int ch = 0; ch < CH; ++ch) {               // chunk-outer / pixel-inner keeps the per-channel constants short-lived
        const int n = nb + 8 * ch;
        float bias[8], es[8], et[8], emu[8], eis[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            s1[2 * ch + (j >> 2)][j & 3] = 0.f; s2[2 * ch + (j >> 2)][j & 3] = 0.f;
            bias[j] = a.bias ? a.bias[n + j] : 0.f;
            if (a.ep.mode == PA_OUT_BWD) {
                es[j] = a.ep.scale[n + j]; et[j] = a.ep.shift[n + j]; emu[j] = a.ep.mean[n + j]; eis[j] = a.ep.invstd[n + j];
            }
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int m = pix(mi);
            if (m < 0) continue;
            const size_t idx = (size_t)m * N + n;
            float e1[8], e2[8];
            pa_read8(a.add1, idx, n, e1);
            pa_read8(a.add2, idx, n, e2);
            bf16x8 o;
            if (a.ep.mode == PA_OUT_BWD) {
                bf16x8 xr = *reinterpret_cast<const bf16x8*>(a.ep.xref + idx);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int ni = 2 * ch + (j >> 2);
                    float v = acc[ni][mi][j & 3] + bias[j] + e1[j] + e2[j];
                    float x = (float)xr[j];
                    float dz = (fmaf(es[j], x, et[j]) > 0.f) ? v : 0.f;
                    o[j] = (bf16)dz;
                    float dzr = (float)o[j];
                    s1[ni][j & 3] += dzr;
                    s2[ni][j & 3] += dzr * (x - emu[j]) * eis[j];
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int ni = 2 * ch + (j >> 2);
                    float v = acc[ni][mi][j & 3] + bias[j] + e1[j] + e2[j];
                    o[j] = (bf16)v;
                    float rv = (float)o[j];
                    s1[ni][j & 3] += rv;
                    s2[ni][j & 3] += rv * rv;
                }
            }
            *reinterpret_cast<bf16x8*>(a.out + idx) = o;
        }
    }
    if (a.ep.mode != PA_OUT_PLAIN) {
        // per-workgroup partial row of the two per-channel reductions: no atomics (a float atomicAdd per
        // channel per wave cost 6-10x the whole conv); the BatchNorm finalize kernel sums the rows.
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float x1 = s1[ni][j], x2 = s2[ni][j];
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) { x1 += __shfl_xor(x1, o, 64); x2 += __shfl_xor(x2, o, 64); }
                if ((lane & 15) == 0) {
                    const int col = wn * (BN / 2) + (lane >> 4) * (4 * NI) + 4 * ni + j;
                    red[(wm * BN + col) * 2] = x1;
                    red[(wm * BN + col) * 2 + 1] = x2;
                }
            }
        }
        __syncthreads();
        for (int c = tid; c < BN; c += 256) {
            f32x2 v = {red[c * 2] + red[(BN + c) * 2], red[c * 2 + 1] + red[(BN + c) * 2 + 1]};
            *reinterpret_cast<f32x2*>(a.ep.stats + ((size_t)stat_row * N + n0 + c) * 2) = v;
        }
    }
}
