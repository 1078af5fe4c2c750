// Epilogues shared by the MFMA convolution kernels (conv_igemm.hip, conv3x3_tile.hip, conv1x1_tile.hip):
// pa_conv_epilogue (direct), pa_conv_epilogue_lds (through an LDS tile), pa_conv_epilogue_auto (picks per mode).
//
// Accumulator layout it expects: a 256-thread workgroup as 2 (wm: pixels) x 2 (wn: channels) waves,
// wave tile (16*MI pixels) x (16*NI channels); fragment (ni, mi) is the 16x16 MFMA output whose ROWS
// are output channels and whose COLUMNS (lane & 15) are pixels.  The weight rows were staged permuted
// (pa_weight_row_of_lds_row) so that lane group q = lane >> 4 of a wave owns, for each 32-channel chunk ch,
// the 8 CONSECUTIVE channels 32*ch + 8*q .. +7: all epilogue traffic (addends, xref, stores) is 16-byte
// accesses and the four lane groups together cover one contiguous 64-byte run per pixel.
#pragma once
#include "common.h"
#include "kernels.h"
#ifndef PA_EPI_STAMP
#define PA_EPI_STAMP(i) do { } while (0)
#endif

// LDS row lr (0 .. BN-1) of the weight tile holds which output channel (relative to the tile's n0)?
template <int BN, int NI>
__device__ __forceinline__ int pa_weight_row_of_lds_row(int lr) {
    const int wt = lr / (BN / 2), l = lr - wt * (BN / 2);
    const int ni = l >> 4, rr = l & 15;
    // lane group q = rr >> 2 of fragment ni owns channels 32*(ni>>1) + 8*q + 4*(ni&1) + (rr&3): the 8-channel chunk ch = ni>>1
    // of the four lane groups is ONE contiguous 64-byte run per pixel (4 x 16 B), not four pieces 32 bytes apart
    return wt * (BN / 2) + 32 * (ni >> 1) + 8 * (rr >> 2) + 4 * (ni & 1) + (rr & 3);
}

// inverse: which LDS row holds output channel c (0 .. BN-1, relative to the tile's n0)?
template <int BN, int NI>
__device__ __forceinline__ int pa_lds_row_of_weight_row(int c) {
    const int wt = c / (BN / 2), l = c - wt * (BN / 2);
    const int ch = l >> 5, rem = l & 31;
    const int q = rem >> 3, h = (rem >> 2) & 1, j = rem & 3;
    return wt * (BN / 2) + (2 * ch + h) * 16 + 4 * q + j;
}

// pix(mi) -> flattened NHWC pixel index of fragment column (lane & 15) of fragment row-block mi, or -1
// `red` = at least 14*BN floats of LDS that are dead by now (every wave is past its last read of them: the callers
// barrier before the epilogue); stat_row = this workgroup's partial row.
// The per-channel constants (BatchNorm-backward scale/shift/mean/invstd, the transforms of the two addends, the bias)
// are staged ONCE per call into LDS: read from global per use they were ~180 of the ~200 load instructions of a
// data-gradient epilogue, all in front of the data loads in the in-order vmcnt queue.  Per 8-channel chunk the
// operand loads of all MI pixels are issued together, before the first use.
template <int BN, int NI, int MI, class PixFn>
__device__ __forceinline__ void pa_conv_epilogue(const PaConvArgs& a, f32x4 (&acc)[NI][MI], int n0, int wm, int wn,
                                                 PixFn pix, float* red, int stat_row) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int N = a.Cout;
    constexpr int CH = NI / 2;                       // 8-channel chunks per lane
    float4* cep = reinterpret_cast<float4*>(red + 4 * BN);      // [BN] {scale, shift, mean, invstd}
    float4* ca1 = cep + BN;                                      // [BN] {k0, k1, k2, bias}
    float4* ca2 = ca1 + BN;                                      // [BN] {k0, k1, k2, -}
    const int m1 = a.add1.mode, m2 = a.add2.mode;
    for (int i = tid; i < BN; i += 256) {
        const int n = n0 + i;
        if (a.ep.mode == PA_OUT_BWD) cep[i] = make_float4(a.ep.scale[n], a.ep.shift[n], a.ep.mean[n], a.ep.invstd[n]);
        float4 k = make_float4(0.f, 0.f, 0.f, a.bias ? a.bias[n] : 0.f);
        if (m1 == PA_LD_BNRELU || m1 == PA_LD_LIN2) { k.x = a.add1.k0[n]; k.y = a.add1.k1[n]; if (m1 == PA_LD_LIN2) k.z = a.add1.k2[n]; }
        ca1[i] = k;
        if (m2 == PA_LD_BNRELU || m2 == PA_LD_LIN2) ca2[i] = make_float4(a.add2.k0[n], a.add2.k1[n], m2 == PA_LD_LIN2 ? a.add2.k2[n] : 0.f, 0.f);
    }
    __syncthreads();
    const int lb = wn * (BN / 2) + (lane >> 4) * 8;              // tile-relative channel of this lane's chunk 0
    const int nb = n0 + lb;
    float s1[NI][4], s2[NI][4];
#pragma unroll
    for (int ch = 0; ch < CH; ++ch) {
        const int n = nb + 32 * ch, ln = lb + 32 * ch;
#pragma unroll
        for (int j = 0; j < 8; ++j) { s1[2 * ch + (j >> 2)][j & 3] = 0.f; s2[2 * ch + (j >> 2)][j & 3] = 0.f; }
        // ---- every operand of the chunk in flight first
        unsigned idx[MI];                            // 32-bit element offsets: pa_launch_conv admits M * Cout < 2^31 only
        bf16x8 xr[MI], p1[MI], q1[MI], p2[MI], q2[MI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int m = pix(mi);
            idx[mi] = m < 0 ? 0u : (unsigned)m * (unsigned)N + (unsigned)n;     // clamped: unconditional loads, the store is guarded
            if (a.ep.mode == PA_OUT_BWD) xr[mi] = *reinterpret_cast<const bf16x8*>(a.ep.xref + idx[mi]);
            if (m1 != PA_LD_NONE) p1[mi] = *reinterpret_cast<const bf16x8*>(a.add1.p + idx[mi]);
            if (m1 == PA_LD_LIN2) q1[mi] = *reinterpret_cast<const bf16x8*>(a.add1.q + idx[mi]);
            if (m2 != PA_LD_NONE) p2[mi] = *reinterpret_cast<const bf16x8*>(a.add2.p + idx[mi]);
            if (m2 == PA_LD_LIN2) q2[mi] = *reinterpret_cast<const bf16x8*>(a.add2.q + idx[mi]);
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            if (pix(mi) < 0) continue;
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int ni = 2 * ch + (j >> 2);
                const float4 k1 = ca1[ln + j];
                float v = acc[ni][mi][j & 3] + k1.w;
                if (m1 == PA_LD_PLAIN) v += (float)p1[mi][j];
                else if (m1 == PA_LD_LIN2) v += fmaf(k1.x, (float)p1[mi][j], fmaf(k1.y, (float)q1[mi][j], k1.z));
                else if (m1 == PA_LD_BNRELU) v += fmaxf(fmaf(k1.x, (float)p1[mi][j], k1.y), 0.f);
                if (m2 == PA_LD_PLAIN) v += (float)p2[mi][j];
                else if (m2 == PA_LD_LIN2) { const float4 k2 = ca2[ln + j]; v += fmaf(k2.x, (float)p2[mi][j], fmaf(k2.y, (float)q2[mi][j], k2.z)); }
                else if (m2 == PA_LD_BNRELU) { const float4 k2 = ca2[ln + j]; v += fmaxf(fmaf(k2.x, (float)p2[mi][j], k2.y), 0.f); }
                if (a.ep.mode == PA_OUT_BWD) {
                    const float4 e = cep[ln + j];
                    const float x = (float)xr[mi][j];
                    const float dz = (fmaf(e.x, x, e.y) > 0.f) ? v : 0.f;
                    o[j] = (bf16)dz;
                    const float dzr = (float)o[j];
                    s1[ni][j & 3] += dzr;
                    s2[ni][j & 3] += dzr * (x - e.z) * e.w;
                } else {
                    o[j] = (bf16)v;
                    const float rv = (float)o[j];
                    s1[ni][j & 3] += rv;
                    s2[ni][j & 3] += rv * rv;
                }
            }
            *reinterpret_cast<bf16x8*>(a.out + idx[mi]) = o;
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
                    const int col = wn * (BN / 2) + 32 * (ni >> 1) + (lane >> 4) * 8 + 4 * (ni & 1) + j;
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

// LDS-only workgroup barrier: __syncthreads() also drains vmcnt, i.e. waits for every global load AND store in flight -- between the passes of
// an epilogue that is the round trip of the previous pass's output stores (and of operand loads requested ahead), for a barrier that only
// orders LDS accesses
__device__ __forceinline__ void pa_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Per-channel sums of the workgroup from the threads' 8-channel partial sums {s1, s2} (thread = chunk tid % CPR of row group tid / CPR), through
// LDS: P >= NT * 16 floats, dead (callers barrier first).  Returns the two sums of channel tid (threads tid < BN; others return 0).
// Order (the one of the shuffle tree this replaces -- 48 ds_bpermute per thread, 1500 cycles of a 5000-cycle epilogue on the one-wave-per-SIMD
// launches): inside a wave the 64 / CPR row groups of a chunk combine as a balanced tree over neighbours (lane ^ CPR, ^ 2 CPR, ...), then the
// waves are added in order.
// TSMALL: P holds NT * 8 floats only (the 64-channel row-tile kernel's ring half): the two sums go through it one after the other.
template <int BN, int NT, bool TSMALL = false>
__device__ __forceinline__ f32x2 pa_stats_reduce(const float (&s1)[8], const float (&s2)[8], float* P) {
    constexpr int CPR = BN / 8, KPW = 64 / CPR, NW = NT / 64;
    const int tid = threadIdx.x;
    const int chunk = tid % CPR, g = tid / CPR;
    f32x2 tot = {0.f, 0.f};
    if constexpr (TSMALL) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (h) pa_lds_barrier();
            f32x4* dst = reinterpret_cast<f32x4*>(P + g * BN + chunk * 8);
            const float (&s)[8] = h ? s2 : s1;
            dst[0] = f32x4{s[0], s[1], s[2], s[3]}; dst[1] = f32x4{s[4], s[5], s[6], s[7]};
            pa_lds_barrier();
            if (tid < BN) {
#pragma unroll
                for (int w = 0; w < NW; ++w) {
                    float v[KPW];
#pragma unroll
                    for (int k = 0; k < KPW; ++k) v[k] = P[(w * KPW + k) * BN + tid];
#pragma unroll
                    for (int o = 1; o < KPW; o <<= 1)
#pragma unroll
                        for (int k = 0; k < KPW; k += 2 * o) v[k] += v[k + o];
                    tot[h] = w == 0 ? v[0] : tot[h] + v[0];
                }
            }
        }
        return tot;
    }
    f32x4* dst = reinterpret_cast<f32x4*>(P + (g * BN + chunk * 8) * 2);
#pragma unroll
    for (int j = 0; j < 4; ++j) dst[j] = f32x4{s1[2 * j], s2[2 * j], s1[2 * j + 1], s2[2 * j + 1]};
    pa_lds_barrier();
    if (tid < BN) {
        f32x2 v[NW][KPW];                            // every read in flight before the first addition
#pragma unroll
        for (int w = 0; w < NW; ++w)
#pragma unroll
            for (int k = 0; k < KPW; ++k) v[w][k] = *reinterpret_cast<const f32x2*>(P + ((w * KPW + k) * BN + tid) * 2);
#pragma unroll
        for (int w = 0; w < NW; ++w) {
#pragma unroll
            for (int o = 1; o < KPW; o <<= 1)
#pragma unroll
                for (int k = 0; k < KPW; k += 2 * o) v[w][k] += v[w][k + o];
            tot = w == 0 ? v[0][0] : tot + v[w][0];
        }
    }
    return tot;
}

// ------------------------------------------------------------------------------------------------------------------
// LDS-TRANSPOSED epilogue.  The direct epilogue above stores, per instruction, 16-byte pieces that lie 32 bytes apart
// in 16 different pixel rows: measured on the 3x3 tile kernel (tools/bench_conv3.py ablation) the plain store of a
// 25 MB output costs 11.6 us = 2.2 TB/s, a third of the kernel.  Here the accumulators go through a 16 KB fp32 LDS
// tile, 32 pixel rows at a time, and are read back so that 16 (BN = 128) consecutive lanes cover one whole 256-byte
// pixel row: every global access of the epilogue (addends, reference tensor, output) is a fully coalesced 16 B/lane
// stream, and a thread owns the SAME 8 channels for all its pixels (bias / BatchNorm constants loaded once,
// per-channel reductions in registers).  Arithmetic and rounding points are those of pa_conv_epilogue.
//   T   : >= max(32 * BN, 16 * NT) floats of LDS that are dead after the K loop (16-byte slots XOR-swizzled by the row; the statistics reduction
//         takes 16 floats per thread -- TSMALL: 8)
//   pix : (wm, mi, p) -> flattened NHWC pixel index of pixel p (0..15) of fragment row-block mi of wave row wm, or -1
// Measured (tools/bench_conv1.py, bench.py): forward epilogues (plain / statistics, with residual addends) gain 10-30 %
// on the 1x1 kernels; the BatchNorm-backward epilogue (reference tensor + mask + two reductions) is SLOWER this way
// (one memory round trip per 32-row pass is exposed to all 256 threads by the pass barrier; prefetching the next
// sweep's operands costs more registers than it hides).  pa_conv_epilogue_auto therefore keeps the direct epilogue
// for PA_OUT_BWD.
// NT = threads of the workgroup: 256 (2 waves along the pixels: 32-row passes) or 512 (4 waves along the pixels: 64-row passes, T >= 64 * BN floats)
// FUSE (MI * BN <= 128, i.e. the whole workgroup tile fits the 16 * NT floats the statistics reduction needs anyway): all MI passes are staged at
// once -- one barrier instead of 2 * MI - 1
template <int BN, int NI, int MI, int NT = 256, bool TSMALL = false, bool FUSE = false, class PixFn>
__device__ __forceinline__ void pa_conv_epilogue_lds(const PaConvArgs& a, f32x4 (&acc)[NI][MI], int n0, int wm, int wn,
                                                     PixFn pix, float* T, int stat_row) {
    constexpr int CPR = BN / 8;                      // 8-channel chunks per pixel row
    constexpr int RPS = NT / CPR;                    // pixel rows per sweep of the NT threads
    constexpr int SW = (NT / 8) / RPS;               // sweeps per pass of NT / 8 rows (16 per wave row)
    constexpr int NW = NT / 64;
    const int tid = threadIdx.x, lane = tid & 63;
    const int N = a.Cout;
    const int chunk = tid % CPR, rsub = tid / CPR;
    const int n = n0 + chunk * 8;
    float bias[8], s1[8], s2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; bias[j] = 0.f; }
    if (a.bias) {                                    // (one uniform branch around the eight loads: a select per element made hipcc wait for them in front of the staging)
#pragma unroll
        for (int j = 0; j < 8; ++j) bias[j] = a.bias[n + j];
    }
    // PRE (workgroup tiles of at most two sweeps: the latency-bound small launches): the raw addend operands of every sweep and the addends'
    // per-channel constants are requested HERE, before the staging barrier, instead of inside each sweep (a memory round trip per sweep with
    // one wave per SIMD).  Same arithmetic as pa_read8 (pa_apply8).
    constexpr int IT = MI * SW;
    constexpr bool PRE = IT <= 2;
    const int m1 = a.add1.mode, m2 = a.add2.mode;
    bf16x8 p1[PRE ? IT : 1], q1[PRE ? IT : 1], p2[PRE ? IT : 1], q2[PRE ? IT : 1];
    float ka1[8], kb1[8], kc1[8], ka2[8], kb2[8], kc2[8];
    if constexpr (PRE) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int sw = 0; sw < SW; ++sw) {
                const int it = mi * SW + sw, r = sw * RPS + rsub;
                const int m = pix(r >> 4, mi, r & 15);
                const unsigned idx = m < 0 ? 0u : (unsigned)m * (unsigned)N + (unsigned)n;      // clamped, unconditional
                if (m1 != PA_LD_NONE) p1[it] = *reinterpret_cast<const bf16x8*>(a.add1.p + idx);
                if (m1 == PA_LD_LIN2) q1[it] = *reinterpret_cast<const bf16x8*>(a.add1.q + idx);
                if (m2 != PA_LD_NONE) p2[it] = *reinterpret_cast<const bf16x8*>(a.add2.p + idx);
                if (m2 == PA_LD_LIN2) q2[it] = *reinterpret_cast<const bf16x8*>(a.add2.q + idx);
            }
        if (m1 == PA_LD_BNRELU || m1 == PA_LD_LIN2) {
            const f32x4 s0 = *reinterpret_cast<const f32x4*>(a.add1.k0 + n), s4 = *reinterpret_cast<const f32x4*>(a.add1.k0 + n + 4);
            const f32x4 t0 = *reinterpret_cast<const f32x4*>(a.add1.k1 + n), t4 = *reinterpret_cast<const f32x4*>(a.add1.k1 + n + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { ka1[j] = s0[j]; ka1[j + 4] = s4[j]; kb1[j] = t0[j]; kb1[j + 4] = t4[j]; }
            if (m1 == PA_LD_LIN2) {
#pragma unroll
                for (int j = 0; j < 8; ++j) kc1[j] = a.add1.k2[n + j];
            }
        }
        if (m2 == PA_LD_BNRELU || m2 == PA_LD_LIN2) {
            const f32x4 s0 = *reinterpret_cast<const f32x4*>(a.add2.k0 + n), s4 = *reinterpret_cast<const f32x4*>(a.add2.k0 + n + 4);
            const f32x4 t0 = *reinterpret_cast<const f32x4*>(a.add2.k1 + n), t4 = *reinterpret_cast<const f32x4*>(a.add2.k1 + n + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { ka2[j] = s0[j]; ka2[j + 4] = s4[j]; kb2[j] = t0[j]; kb2[j + 4] = t4[j]; }
            if (m2 == PA_LD_LIN2) {
#pragma unroll
                for (int j = 0; j < 8; ++j) kc2[j] = a.add2.k2[n + j];
            }
        }
    }
    auto addend = [](int md, const bf16x8& p, const bf16x8& q, const float (&ka)[8], const float (&kb)[8], const float (&kc)[8], float (&v)[8]) {
        if (md == PA_LD_NONE) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = 0.f;
        } else if (md == PA_LD_PLAIN) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (float)p[j];
        } else if (md == PA_LD_BNRELU) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = fmaxf(fmaf(ka[j], (float)p[j], kb[j]), 0.f);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = fmaf(ka[j], (float)p[j], fmaf(kb[j], (float)q[j], kc[j]));
        }
    };
    // writer side: lane (q = lane >> 4, p = lane & 15) of wave (wm, wn) holds channels wn*BN/2 + q*4*NI + 4*ni + reg of pixel p
    const int wrow = wm * 16 + (lane & 15);
    const int wslot0 = (wn * (BN / 2)) / 4 + 2 * (lane >> 4);                // 16-byte slot (4 floats) of fragment 0; fragment ni: + 8*(ni>>1) + (ni&1)
    constexpr int ROWS = NT / 8;                     // pixel rows of a pass (16 per wave row)
    if constexpr (FUSE) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
                *reinterpret_cast<f32x4*>(T + (mi * ROWS + wrow) * BN + (((wslot0 + 8 * (ni >> 1) + (ni & 1)) ^ (wrow & 7)) << 2)) = acc[ni][mi];
        pa_lds_barrier();
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        if constexpr (!FUSE) {
            if (mi) pa_lds_barrier();                // the previous pass has been read
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
                *reinterpret_cast<f32x4*>(T + wrow * BN + (((wslot0 + 8 * (ni >> 1) + (ni & 1)) ^ (wrow & 7)) << 2)) = acc[ni][mi];
            pa_lds_barrier();
        }
        PA_EPI_STAMP(7 + 2 * mi);
        const float* Tp = T + (FUSE ? mi * ROWS * BN : 0);
#pragma unroll
        for (int sw = 0; sw < SW; ++sw) {
            const int r = sw * RPS + rsub;           // 0..31: wave row r >> 4, pixel r & 15
            const int m = pix(r >> 4, mi, r & 15);
            if (m < 0) continue;
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(Tp + r * BN + (((2 * chunk) ^ (r & 7)) << 2));
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(Tp + r * BN + (((2 * chunk + 1) ^ (r & 7)) << 2));
            const unsigned idx = (unsigned)m * (unsigned)N + (unsigned)n;
            float e1[8], e2[8];
            if constexpr (PRE) {
                addend(m1, p1[mi * SW + sw], q1[mi * SW + sw], ka1, kb1, kc1, e1);
                addend(m2, p2[mi * SW + sw], q2[mi * SW + sw], ka2, kb2, kc2, e2);
            } else {
                pa_read8(a.add1, idx, n, e1);
                pa_read8(a.add2, idx, n, e2);
            }
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float v = (j < 4 ? v0[j & 3] : v1[j & 3]) + bias[j] + e1[j] + e2[j];
                o[j] = (bf16)v;
                float rv = (float)o[j];
                s1[j] += rv;
                s2[j] += rv * rv;
            }
            *reinterpret_cast<bf16x8*>(a.out + idx) = o;
        }
        PA_EPI_STAMP(8 + 2 * mi);
    }
    if (a.ep.mode != PA_OUT_PLAIN) {
        PA_EPI_STAMP(11);
        pa_lds_barrier();                            // T is dead
        const f32x2 v = pa_stats_reduce<BN, NT, TSMALL>(s1, s2, T);
        if (tid < BN) *reinterpret_cast<f32x2*>(a.ep.stats + ((size_t)stat_row * N + n0 + tid) * 2) = v;
    }
}

// BatchNorm-BACKWARD epilogue through the same LDS tile (coalesced 256-byte rows for the reference tensor, the addends and
// the output).  Differences from the forward variant that make it pay: (1) the operands of TWO 32-row passes (reference
// tensor, both addends incl. the second operand of a LIN2 addend) and the thread's per-channel constants are requested
// together, before the first pass barrier, so one memory round trip covers the whole workgroup tile of a 64-row kernel;
// (2) a thread owns the same 8 channels for all its pixels, so the constants live in registers; (3) the second reduction
// is accumulated as sum(dz * x) and turned into sum(dz * xhat) = invstd * (sum(dz*x) - mean * sum(dz)) once per
// workgroup row, which removes mean / invstd from the per-element work.
template <int BN, int NI, int MI, bool TAB, int NT = 256, bool TSMALL = false, bool FUSE = false, class PixFn>
__device__ __forceinline__ void pa_conv_epilogue_lds_bwd(const PaConvArgs& a, f32x4 (&acc)[NI][MI], int n0, int wm, int wn,
                                                         PixFn pix, float* T, int stat_row, float4* ctab = nullptr) {
    // ctab (optional, 2 * BN float4 of LDS outside T): the per-channel constants live there instead of in 40 registers
    // (entry j * CPR + chunk: conflict-free for the 16 chunks of a wave) -- for the kernels that run 3 workgroups per CU
    constexpr int CPR = BN / 8;
    constexpr int RPS = NT / CPR;
    constexpr int SW = (NT / 8) / RPS;
    constexpr int NW = NT / 64;
    constexpr int G = FUSE ? MI : ((MI >= 2 && !TAB) ? 2 : 1);    // passes per operand request (1 for the 168-register kernels)
    constexpr int ROWS = NT / 8;
    static_assert(!FUSE || !TAB, "fused staging: constants in registers");
    constexpr int IT = G * SW;
    const int tid = threadIdx.x, lane = tid & 63;
    const int N = a.Cout;
    const int chunk = tid % CPR, rsub = tid / CPR;
    const int n = n0 + chunk * 8;
    const int m1 = a.add1.mode, m2 = a.add2.mode;
    float s1[8], s2[8], es[8], et[8], ka[8], kb[8], kc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
    const int wrow = wm * 16 + (lane & 15);
    const int wslot0 = (wn * (BN / 2)) / 4 + 2 * (lane >> 4);
#pragma unroll
    for (int g = 0; g < MI; g += G) {
        // ---- request everything the next G passes need
        unsigned idx[IT];                            // 32-bit element offsets (the tile launchers admit M * Cout < 2^31 only)
        bool ok[IT];
        bf16x8 xr[IT], p1[IT], q1[IT], p2[IT];
#pragma unroll
        for (int pp = 0; pp < G; ++pp)
#pragma unroll
            for (int sw = 0; sw < SW; ++sw) {
                const int it = pp * SW + sw, r = sw * RPS + rsub;
                const int m = pix(r >> 4, g + pp, r & 15);
                ok[it] = m >= 0;
                idx[it] = ok[it] ? (unsigned)m * (unsigned)N + (unsigned)n : 0u;
#ifdef PA_TUNING
                if (a.dbg & 16) { xr[it] = bf16x8{}; p1[it] = bf16x8{}; q1[it] = bf16x8{}; p2[it] = bf16x8{}; continue; }      // (timing bound, wrong results: the epilogue without its operand round trip)
#endif
                xr[it] = *reinterpret_cast<const bf16x8*>(a.ep.xref + idx[it]);
                if (m1 != PA_LD_NONE) p1[it] = *reinterpret_cast<const bf16x8*>(a.add1.p + idx[it]);
                if (m1 == PA_LD_LIN2) q1[it] = *reinterpret_cast<const bf16x8*>(a.add1.q + idx[it]);
                if (m2 != PA_LD_NONE) p2[it] = *reinterpret_cast<const bf16x8*>(a.add2.p + idx[it]);
            }
        if (g == 0) {
            if (TAB) {
                for (int i = tid; i < BN; i += NT) {
                    const int e = (i & 7) * CPR + (i >> 3);
                    ctab[e] = make_float4(a.ep.scale[n0 + i], a.ep.shift[n0 + i], 0.f, 0.f);
                    ctab[BN + e] = m1 == PA_LD_LIN2 ? make_float4(a.add1.k0[n0 + i], a.add1.k1[n0 + i], a.add1.k2[n0 + i], 0.f) : make_float4(0.f, 0.f, 0.f, 0.f);
                }                                    // (visible after the first pass barrier below)
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    es[j] = a.ep.scale[n + j]; et[j] = a.ep.shift[n + j];
                    if (m1 == PA_LD_LIN2) { ka[j] = a.add1.k0[n + j]; kb[j] = a.add1.k1[n + j]; kc[j] = a.add1.k2[n + j]; }
                }
            }
        }
        if constexpr (FUSE) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    *reinterpret_cast<f32x4*>(T + (mi * ROWS + wrow) * BN + (((wslot0 + 8 * (ni >> 1) + (ni & 1)) ^ (wrow & 7)) << 2)) = acc[ni][mi];
            pa_lds_barrier();
        }
#pragma unroll
        for (int pp = 0; pp < G; ++pp) {
            const int mi = g + pp;
            if constexpr (!FUSE) {
                if (mi) pa_lds_barrier();            // the previous pass has been read
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    *reinterpret_cast<f32x4*>(T + wrow * BN + (((wslot0 + 8 * (ni >> 1) + (ni & 1)) ^ (wrow & 7)) << 2)) = acc[ni][mi];
                pa_lds_barrier();
            }
            const float* Tp = T + (FUSE ? mi * ROWS * BN : 0);
#pragma unroll
            for (int sw = 0; sw < SW; ++sw) {
                const int it = pp * SW + sw, r = sw * RPS + rsub;
                if (!ok[it]) continue;
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(Tp + r * BN + (((2 * chunk) ^ (r & 7)) << 2));
                const f32x4 v1 = *reinterpret_cast<const f32x4*>(Tp + r * BN + (((2 * chunk + 1) ^ (r & 7)) << 2));
                bf16x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float v = (j < 4 ? v0[j & 3] : v1[j & 3]);
                    float ej = TAB ? 0.f : es[j], tj = TAB ? 0.f : et[j];
                    if (TAB) { const float4 e = ctab[j * CPR + chunk]; ej = e.x; tj = e.y; }
                    if (m1 == PA_LD_PLAIN) v += (float)p1[it][j];
                    else if (m1 == PA_LD_LIN2) {
                        if (TAB) { const float4 k = ctab[BN + j * CPR + chunk]; v += fmaf(k.x, (float)p1[it][j], fmaf(k.y, (float)q1[it][j], k.z)); }
                        else v += fmaf(ka[j], (float)p1[it][j], fmaf(kb[j], (float)q1[it][j], kc[j]));
                    }
                    if (m2 != PA_LD_NONE) v += (float)p2[it][j];
                    const float x = (float)xr[it][j];
                    const float dz = (fmaf(ej, x, tj) > 0.f) ? v : 0.f;
                    o[j] = (bf16)dz;
                    const float dzr = (float)o[j];
                    s1[j] += dzr;
                    s2[j] += dzr * x;
                }
                *reinterpret_cast<bf16x8*>(a.out + idx[it]) = o;
            }
        }
    }
    // ---- statistics row (the mean / invstd of the channel requested before the reduction's barrier)
    float cm = 0.f, ci = 0.f;
    if (tid < BN) { cm = a.ep.mean[n0 + tid]; ci = a.ep.invstd[n0 + tid]; }
    pa_lds_barrier();                                // T is dead
    const f32x2 t = pa_stats_reduce<BN, NT, TSMALL>(s1, s2, T);
    if (tid < BN) {
        f32x2 v = {t[0], ci * (t[1] - cm * t[0])};
        *reinterpret_cast<f32x2*>(a.ep.stats + ((size_t)stat_row * N + n0 + tid) * 2) = v;
    }
}

// Plain epilogue (output = accumulators + ONE plain addend, stored bf16) of a tile that owns 2 image rows x 32 columns, PLUS the low-resolution
// half of the upsample-add backward (elementwise.hip upadd_bwd_bb_kernel<LOW>): the 2 x 2 sums of the stored (bf16-rounded) output values, masked by
// the ReLU of the low-resolution tensor they belong to (ep2: BatchNorm-backward mode), stored to out2, with that tensor's two per-channel
// reductions as a partial row.  The main chain of the hourglass' backward pass then has no streaming kernel between this data gradient and the
// low-resolution path (75 MB of traffic, 8 launches per step).
//   256 threads, BN = 128: pass mi holds tile rows {mi*16 + p (image row 0), 32 + mi*16 + p (image row 1)}, p = 0..15; the reading thread
//   (chunk = tid % 16, rsub = tid / 16) owns column x = mi*16 + rsub of BOTH image rows (sweeps 0 / 1) and finds the column's horizontal
//   neighbour x ^ 1 in lane ^ 16 of its own wave.  Sum order = the streaming kernel's: ((top-left + top-right) + bottom-left) + bottom-right.
//   pix(row) -> flattened pixel of tile row `row`; low(x) -> flattened low-resolution pixel of tile column x (even x)
//   ctab: BN float4 of LDS outside T ({scale, shift} of ep2, entry j * 16 + chunk)
template <int BN, int NI, int MI, class PixFn, class LowFn>
__device__ __forceinline__ void pa_conv_epilogue_lds_up(const PaConvArgs& a, f32x4 (&acc)[NI][MI], int n0, int wm, int wn,
                                                        PixFn pix, LowFn low, float* T, int stat_row, float4* ctab) {
    static_assert(BN == 128 && MI == 2, "2 x 32-pixel tiles of the 128-channel row-tile kernel");
    constexpr int CPR = BN / 8, NT = 256;
    const int tid = threadIdx.x, lane = tid & 63;
    const int N = a.Cout;
    const int chunk = tid % CPR, rsub = tid / CPR;
    const int n = n0 + chunk * 8;
    float l1[8], l2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { l1[j] = 0.f; l2[j] = 0.f; }
    for (int i = tid; i < BN; i += NT) ctab[(i & 7) * CPR + (i >> 3)] = make_float4(a.ep2.scale[n0 + i], a.ep2.shift[n0 + i], 0.f, 0.f);      // (visible after the first pass barrier)
    const int wrow = wm * 16 + (lane & 15);
    const int wslot0 = (wn * (BN / 2)) / 4 + 2 * (lane >> 4);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        // ---- operands of the pass: the addend of both image rows, the low-resolution reference tensor
        const int x = mi * 16 + rsub;
        const unsigned i0 = (unsigned)pix(x) * (unsigned)N + (unsigned)n, i1 = (unsigned)pix(32 + x) * (unsigned)N + (unsigned)n;
        const unsigned il = (unsigned)low(x & ~1) * (unsigned)N + (unsigned)n;
        const bf16x8 p0 = *reinterpret_cast<const bf16x8*>(a.add1.p + i0), p1 = *reinterpret_cast<const bf16x8*>(a.add1.p + i1);
        const bf16x8 xl = *reinterpret_cast<const bf16x8*>(a.ep2.xref + il);
        if (mi) pa_lds_barrier();                    // the previous pass has been read
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
            *reinterpret_cast<f32x4*>(T + wrow * BN + (((wslot0 + 8 * (ni >> 1) + (ni & 1)) ^ (wrow & 7)) << 2)) = acc[ni][mi];
        pa_lds_barrier();
        float r0[8], r1[8];
        {
            const int ra = rsub, rb = 16 + rsub;
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(T + ra * BN + (((2 * chunk) ^ (ra & 7)) << 2));
            const f32x4 a1 = *reinterpret_cast<const f32x4*>(T + ra * BN + (((2 * chunk + 1) ^ (ra & 7)) << 2));
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(T + rb * BN + (((2 * chunk) ^ (rb & 7)) << 2));
            const f32x4 b1 = *reinterpret_cast<const f32x4*>(T + rb * BN + (((2 * chunk + 1) ^ (rb & 7)) << 2));
            bf16x8 o0, o1;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                o0[j] = (bf16)((j < 4 ? a0[j & 3] : a1[j & 3]) + (float)p0[j]);
                o1[j] = (bf16)((j < 4 ? b0[j & 3] : b1[j & 3]) + (float)p1[j]);
                r0[j] = (float)o0[j]; r1[j] = (float)o1[j];
            }
            *reinterpret_cast<bf16x8*>(a.out + i0) = o0;
            *reinterpret_cast<bf16x8*>(a.out + i1) = o1;
        }
        // ---- 2 x 2 sum with the neighbouring column (lane ^ 16), mask, store, reductions (even columns)
        bf16x8 ol;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float q0 = __shfl_xor(r0[j], 16, 64), q1 = __shfl_xor(r1[j], 16, 64);
            const float sum = ((r0[j] + q0) + r1[j]) + q1;
            const float4 e = ctab[j * CPR + chunk];
            const float xv = (float)xl[j];
            const float dz = (fmaf(e.x, xv, e.y) > 0.f) ? sum : 0.f;
            ol[j] = (bf16)dz;
            if (!(rsub & 1)) { const float dzr = (float)ol[j]; l1[j] += dzr; l2[j] += dzr * xv; }
        }
        if (!(rsub & 1)) *reinterpret_cast<bf16x8*>(a.out2 + il) = ol;
    }
    float cm = 0.f, ci = 0.f;
    if (tid < BN) { cm = a.ep2.mean[n0 + tid]; ci = a.ep2.invstd[n0 + tid]; }
    pa_lds_barrier();                                // T is dead
    const f32x2 t = pa_stats_reduce<BN, NT, false>(l1, l2, T);
    if (tid < BN) {
        f32x2 v = {t[0], ci * (t[1] - cm * t[0])};
        *reinterpret_cast<f32x2*>(a.ep2.stats + ((size_t)stat_row * N + n0 + tid) * 2) = v;
    }
}

// host side of the dispatch below: can the LDS variant of the BatchNorm-backward epilogue take this launch?
__host__ __device__ inline bool pa_bwd_epilogue_lds_ok(const PaConvArgs& a) {
    return a.xcd < 2 && a.bias == nullptr && (a.add1.mode == PA_LD_NONE || a.add1.mode == PA_LD_PLAIN || a.add1.mode == PA_LD_LIN2) &&
           (a.add2.mode == PA_LD_NONE || a.add2.mode == PA_LD_PLAIN);
}

// forward epilogues through LDS (coalesced rows), backward epilogue direct; pix(wm, mi, p) as above
// T: >= NT * 16 floats, or (TSMALL) >= NT * 8
template <int BN, int NI, int MI, bool BWD_LDS = true, bool TAB = false, int NT = 256, bool TSMALL = false, class PixFn>
__device__ __forceinline__ void pa_conv_epilogue_auto(const PaConvArgs& a, f32x4 (&acc)[NI][MI], int n0, int wm, int wn,
                                                      PixFn pix, float* T, int stat_row, float4* ctab = nullptr) {
    constexpr bool FUSE = !TSMALL && !TAB && MI > 1 && MI * BN <= 128;
    if (a.ep.mode == PA_OUT_BWD) {
        // supported operand modes of the LDS variant: addend 1 plain / LIN2, addend 2 plain, no bias (every data gradient of
        // the networks); anything else takes the direct epilogue
        const bool lds_ok = BWD_LDS && pa_bwd_epilogue_lds_ok(a);
        if (lds_ok) {
            pa_conv_epilogue_lds_bwd<BN, NI, MI, TAB, NT, TSMALL, FUSE>(a, acc, n0, wm, wn, pix, T, stat_row, ctab);
        } else if constexpr (TAB || NT != 256) {
            __builtin_trap();       // the 168-register / 512-thread kernels carry no direct epilogue: pa_bwd_epilogue_lds_ok() routes such launches elsewhere
        } else {
            const int p = threadIdx.x & 15;
            pa_conv_epilogue<BN, NI, MI>(a, acc, n0, wm, wn, [&](int mi) { return pix(wm, mi, p); }, T, stat_row);
        }
    } else {
        pa_conv_epilogue_lds<BN, NI, MI, NT, TSMALL, FUSE>(a, acc, n0, wm, wn, pix, T, stat_row);
    }
}
