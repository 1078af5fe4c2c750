// Weight gradients of the 1x1 / 3x3 convolutions, TILE version (gfx950):
//
//   dw[n][tap][c] = sum_p dy[p][n] * x[p + tap][c]
//
// The reduction index is the pixel, the slow index of both NHWC operands.  conv_wgrad.hip transposes
// 8x8 blocks in registers while staging (VALU bound) and re-loads/re-transforms both operands for each
// of the 9 taps.  Here
//   * both operand tiles are staged in their NATURAL [pixel][channel] layout (one coalesced 16-byte load
//     and one ds_write_b128 per chunk, transform applied once) and the MFMA fragments -- 8 consecutive
//     PIXELS of one channel per lane -- are read with ds_read_b64_tr_b16, the LDS transpose read of gfx950;
//   * for the 3x3 layers a workgroup stages the 8x16-pixel dy tile and the 10x18 halo of x ONCE and
//     accumulates all 9 taps from them (9 x fewer dy loads, 6.4 x fewer x loads and transforms);
//   * a workgroup owns a (n-block, c-block) of the gradient for ALL taps in registers and walks over
//     many pixel tiles before it writes its fp32 partial slab once (same slab layout / reducer as
//     conv_wgrad.hip: part[split][n][tap*Cin + c]).
// LDS images: [pixel][CH] bf16 rows; the 32-byte granule (16 channels = what 4 lanes of a transpose read
// fetch) is XOR-swizzled with pixel bits (0,1,3) [CH = 128] or (1,3) [CH = 64], which makes the 8 pixel
// rows touched by a 32-lane half of ds_read_b64_tr_b16 fall into 8 distinct bank groups for any tap shift.
#include "common.h"
#include "kernels.h"
#include "wgrad_reduce.h"
#include <stdlib.h>
#include <string.h>
#include <hip/hip_ext.h>

#define PA_WG_GROUP_DEFAULT 1
#define PA_WGRAD_MINPER1_DEFAULT 1
#define PA_WGRAD_MINPER9_DEFAULT 1
// operand loads / slab stores of the weight gradients: PA_WG_NT = 1 marks them non-temporal (streamed once: do not keep them in L2 / the
// Infinity Cache, where the main chain's freshly produced tensors live)
#ifndef PA_WG_NT
#define PA_WG_NT 0
#endif
#if PA_WG_NT
#define PA_WG_LOAD(p) __builtin_nontemporal_load(p)
#define PA_WG_STORE(p, v) __builtin_nontemporal_store(v, p)
#else
#define PA_WG_LOAD(p) (*(p))
#define PA_WG_STORE(p, v) (*(p) = (v))
#endif
#define PA_WG_RT 7            // PMODE / QMODE "decided at run time from the operand" (the grouped kernel: one body per tile shape)
#define PA_WG_STEM 3          // QMODE of the stem: x is the 4-channel image, gathered as 7x7/2 patches (K = 256)
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

template <int CH>
__device__ __forceinline__ int wg_sw(int p) {
    return CH >= 128 ? ((p & 3) | ((p >> 1) & 4)) : (((p >> 1) & 1) | ((p >> 2) & 2));
}

// fragment: lane l receives 8 consecutive pixels (p0 .. p0+7, p0 = base + 8*(l>>4)) of channel 16*gran + (l&15)
template <int CH, class PixFn>
__device__ __forceinline__ bf16x8 wg_tr_frag(const bf16* tile, int gran, PixFn pix) {
    const int l = threadIdx.x & 63;
    s16x4 h[2];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        const int p = pix(8 * (l >> 4) + 4 * hh + ((l & 15) >> 2));
        const bf16* ptr = tile + p * CH + ((gran ^ wg_sw<CH>(p)) << 4) + 4 * (l & 3);
        h[hh] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(ptr));
    }
    s16x8 v = __builtin_shufflevector(h[0], h[1], 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

__device__ __forceinline__ void wg_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// NF/CF: 16-channel fragments per wave along n / c; WNW waves along n (4/WNW along c)
// PIPE (round 5): a launch is ONE workgroup per CU (<= 256 workgroups: the fp32 slabs grow with the workgroup count), i.e. one wave per
// SIMD, and a wave that runs alone is the plain sum of its phases: every tile paid the full global-load round trip of its operands
// (2 - 8 K cycles cold) in front of 1 - 2.3 K cycles of MFMA.  The pipelined form requests tile t+1's 16-byte chunks into registers
// (all of them at once: with one workgroup per CU the kernel may use 512 registers) right before the MFMA section of tile t, and
// transforms + writes them to LDS after it; the barriers are LDS-only (s_waitcnt lgkmcnt(0); s_barrier -- __syncthreads() also drains
// vmcnt, i.e. would wait for the prefetch), the transform constants stay in registers for the whole kernel.  Same sums in the same
// order as the plain form: bitwise the same slabs.
// OCC: workgroups per CU the register allocation must leave room for (PIPE only; 1 = up to 512 registers, the CU is this kernel's alone;
// 2 = at most 256, another queue's workgroup fits beside it)
// LDS elements of an instance: dy tile [128][NB] + x tile / halo image [HPL][CB]
template <int TAPS, int NF, int CF, int WNW>
constexpr int wg_lds_elems() { return 128 * (16 * NF * WNW) + (TAPS == 9 ? 10 * 32 : 128) * (16 * CF * (4 / WNW)); }

// The work of ONE workgroup: partial gradient block (by, bz) of split `split` of S (tiles split, split + S, ...).  Called by the
// one-layer kernel below with its block indices and by the grouped kernel (several layers in one launch) with a decoded job.
template <int TAPS, int NF, int CF, int WNW, int PMODE, int QMODE, bool DB, bool PIPE>
__device__ __forceinline__ void wgrad_tile_body(const PaWgradArgs& a, int ntiles, int split, int S, int by, int bz, bf16* lds) {
    constexpr int WCW = 4 / WNW;
    constexpr int NB = 16 * NF * WNW, CB = 16 * CF * WCW;
    // 3x3: 10 x 18 halo pixels, stored with a row pitch of 32 pixels: the swizzle bits (<= bit 3) and the pixel-in-row
    // part of a fragment address then do not depend on the halo ROW, so the 72 (k-step, tap, half) addresses of a lane
    // are 6 registers + compile-time offsets (with pitch 18 they were recomputed or spilled: 36 spill slots)
    constexpr int PW = 18, PWL = 32, HP = TAPS == 9 ? 180 : 128, HPL = TAPS == 9 ? 10 * PWL : 128;
    constexpr int CPN = NB / 8, CPC = CB / 8;                   // 16-byte chunks per row
    constexpr int PASS_N = 128 * CPN / 256;                     // dy chunks per thread
    constexpr int PASS_C = (HP * CPC + 255) / 256;              // x chunks per thread
    static_assert(128 * NB + HPL * CB == wg_lds_elems<TAPS, NF, CF, WNW>(), "LDS size");
    bf16* dyT = lds;
    bf16* xT = lds + 128 * NB;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave % WNW, wc = wave / WNW;
    const int n0 = by * NB, c0 = bz * CB;
    const int M = a.B * a.H * a.W;
    const int tiles_x = TAPS == 9 ? a.W / 16 : 1, tiles_y = TAPS == 9 ? a.H / 8 : 1;
    const int Kfull = TAPS * a.Cin;

    // a thread always handles the same channel chunk; its transform constants are re-read per tile
    // (L1 hits) so that they are not live across the MFMA section
    const int nchunk = tid % CPN, cchunk = tid % CPC;
    const bool want_db = DB && a.dbpart != nullptr && bz == 0;
    // operand modes: template constants, or (PA_WG_RT, plain form) read from the operands -- wave-uniform branches
    const bool p_lin2 = PMODE == PA_LD_LIN2 || (PMODE == PA_WG_RT && a.dy.mode == PA_LD_LIN2);
    const bool q_bnrelu = QMODE == PA_LD_BNRELU || (QMODE == PA_WG_RT && a.x.mode == PA_LD_BNRELU);
    static_assert(!PIPE || (PMODE != PA_WG_RT && QMODE != PA_WG_RT), "run-time modes: plain form only");
    float colsum[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) colsum[j] = 0.f;

    f32x4 acc[TAPS][CF][NF];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int cf = 0; cf < CF; ++cf)
#pragma unroll
            for (int f = 0; f < NF; ++f) acc[t][cf][f] = f32x4{0.f, 0.f, 0.f, 0.f};

    // the MFMA section of one staged tile (both forms)
    auto mfma_tile = [&]() {
        if (TAPS == 9) {
            // per-lane element offsets for k-step 0 / halo row 0; everything else is a compile-time offset
            int doff[NF][2], xoff[3][2];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int kl = 8 * (lane >> 4) + 4 * hh + ((lane & 15) >> 2);
#pragma unroll
                for (int f = 0; f < NF; ++f) doff[f][hh] = kl * NB + (((wn * NF + f) ^ wg_sw<NB>(kl)) << 4) + 4 * (lane & 3);
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    const int col = (kl & 15) + d;                                   // + 1 + dx, dx = d - 1
                    xoff[d][hh] = ((kl >> 4) * PWL + col) * CB + ((wc ^ wg_sw<CB>(col)) << 4) + 4 * (lane & 3);
                }
            }
            auto tr8 = [&](const bf16* base, int o0, int o1) {
                s16x4 h0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + o0));
                s16x4 h1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + o1));
                s16x8 v = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
                return __builtin_bit_cast(bf16x8, v);
            };
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                bf16x8 fd[NF];
#pragma unroll
                for (int f = 0; f < NF; ++f) fd[f] = tr8(dyT + ks * 32 * NB, doff[f][0], doff[f][1]);
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int row = 2 * ks + 1 + (t / 3 - 1);                        // halo row of the first 16 pixels
                    bf16x8 fx = tr8(xT + row * PWL * CB, xoff[t % 3][0], xoff[t % 3][1]);
#pragma unroll
                    for (int f = 0; f < NF; ++f)
                        acc[t][0][f] = PA_MFMA_16x16x32(fx, fd[f], acc[t][0][f]);
                }
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                bf16x8 fd[NF];
#pragma unroll
                for (int f = 0; f < NF; ++f) fd[f] = wg_tr_frag<NB>(dyT, wn * NF + f, [&](int kl) { return 32 * ks + kl; });
#pragma unroll
                for (int cf = 0; cf < CF; ++cf) {
                    bf16x8 fx = wg_tr_frag<CB>(xT, wc * CF + cf, [&](int kl) { return 32 * ks + kl; });
#pragma unroll
                    for (int f = 0; f < NF; ++f)
                        acc[0][cf][f] = PA_MFMA_16x16x32(fx, fd[f], acc[0][cf][f]);
                }
            }
        }
    };

    if constexpr (PIPE) {
        constexpr bool LIN2 = PMODE == PA_LD_LIN2, STEM = QMODE == PA_WG_STEM;
        bf16x8 rp[PASS_N], rq[LIN2 ? PASS_N : 1], rx[STEM ? 1 : PASS_C];
        bf16x4 rxl[STEM ? PASS_C : 1], rxh[STEM ? PASS_C : 1];          // stem: the two 4-channel input pixels of a patch chunk
        unsigned lokm = 0, hokm = 0;                                   // ... and which of them lie inside the image (bit u)
        float pk0[8], pk1[8], pk2[8], qk0[8], qk1[8];
        if (LIN2) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { pk0[j] = a.dy.k0[n0 + nchunk * 8 + j]; pk1[j] = a.dy.k1[n0 + nchunk * 8 + j]; pk2[j] = a.dy.k2[n0 + nchunk * 8 + j]; }
        }
        if (QMODE == PA_LD_BNRELU) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { qk0[j] = a.x.k0[c0 + cchunk * 8 + j]; qk1[j] = a.x.k1[c0 + cchunk * 8 + j]; }
        }
        auto coords = [&](int tile, int& b, int& y0, int& x0) {
            b = y0 = x0 = 0;
            if (TAPS == 9) { int t = tile; x0 = (t % tiles_x) * 16; t /= tiles_x; y0 = (t % tiles_y) * 8; b = t / tiles_y; }
        };
        // request every 16-byte chunk of a tile (clamped, unconditional loads: out-of-range chunks read element 0 and are zeroed at staging)
        auto request = [&](int tile) {
            int b, y0, x0; coords(tile, b, y0, x0);
#pragma unroll
            for (int u = 0; u < PASS_N; ++u) {
                const int r = u * (256 / CPN) + tid / CPN;
                int m; bool ok;
                if (TAPS == 9) { m = (b * a.H + y0 + (r >> 4)) * a.W + x0 + (r & 15); ok = true; }
                else { m = tile * 128 + r; ok = m < M; }
                const size_t idx = ok ? (size_t)m * a.Cout + n0 + nchunk * 8 : 0;
                rp[u] = PA_WG_LOAD(reinterpret_cast<const bf16x8*>(a.dy.p + idx));
                if (LIN2) rq[u] = PA_WG_LOAD(reinterpret_cast<const bf16x8*>(a.dy.q + idx));
            }
            if constexpr (STEM) {
                // 7x7 stride-2 stem (see the plain form below): the thread's pixel of pass 0 by two divisions per TILE, then walked
                const int m0 = tile * 128 + tid / CPC, HWo = a.H * a.W;
                int sb = m0 / HWo; const int rem = m0 - sb * HWo; int sy = rem / a.W, sx = rem - sy * a.W;
                const int chunk = c0 / 8 + cchunk, ky = chunk >> 2, q = chunk & 3;
                const int Hin = 2 * a.H, Win = 2 * a.W;
                lokm = hokm = 0;
#pragma unroll
                for (int u = 0; u < PASS_C; ++u) {
                    const int hp = u * (256 / CPC) + tid / CPC;
                    const bool ok = tile * 128 + hp < M;
                    const int bb = ok ? sb : 0, y = ok ? sy : 0, x = ok ? sx : 0;
                    sx += 256 / CPC;                 // the next pass's pixel
                    while (sx >= a.W) { sx -= a.W; if (++sy >= a.H) { sy = 0; ++sb; } }
                    const int yi = 2 * y + ky - 3, xi = 2 * x + 2 * q - 3;
                    const bool rowok = ok && ky < 7 && (unsigned)yi < (unsigned)Hin;
                    const bool lok = rowok && (unsigned)xi < (unsigned)Win, hok = rowok && (unsigned)(xi + 1) < (unsigned)Win;
                    const bf16* rowp = a.x.p + ((size_t)bb * Hin + (rowok ? yi : 0)) * Win * 4;
                    rxl[u] = *reinterpret_cast<const bf16x4*>(rowp + (size_t)(lok ? xi : 0) * 4);          // clamped, unconditional
                    rxh[u] = *reinterpret_cast<const bf16x4*>(rowp + (size_t)(hok ? xi + 1 : 0) * 4);
                    lokm |= (lok ? 1u : 0u) << u; hokm |= (hok ? 1u : 0u) << u;
                }
            } else {
#pragma unroll
            for (int u = 0; u < PASS_C; ++u) {
                const int hp = u * (256 / CPC) + tid / CPC;
                int m; bool ok;
                if (TAPS == 9) {
                    const int hy = hp / PW, hx = hp - hy * PW;
                    const int y = y0 + hy - 1, x = x0 + hx - 1;
                    ok = hp < HP && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
                    m = (b * a.H + y) * a.W + x;
                } else { m = tile * 128 + hp; ok = m < M; }
                const size_t idx = ok ? (size_t)m * a.Cin + c0 + cchunk * 8 : 0;
                rx[u] = PA_WG_LOAD(reinterpret_cast<const bf16x8*>(a.x.p + idx));
            }
            }
        };
        // transform the requested chunks and write them to the LDS images (the same arithmetic, element for element, as the plain form)
        auto stage = [&](int tile) {
            int b, y0, x0; coords(tile, b, y0, x0);
#pragma unroll
            for (int u = 0; u < PASS_N; ++u) {
                const int r = u * (256 / CPN) + tid / CPN;
                const bool ok = TAPS == 9 ? true : (tile * 128 + r < M);
                bf16x8 o;
                if (LIN2) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] = (bf16)fmaf(pk0[j], (float)rp[u][j], fmaf(pk1[j], (float)rq[u][j], pk2[j]));
                } else {
                    o = rp[u];
                }
                if (!ok) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] = (bf16)0.f;
                }
                if (DB && want_db) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) colsum[j] += (float)o[j];
                }
                *reinterpret_cast<bf16x8*>(dyT + r * NB + (((nchunk >> 1) ^ wg_sw<NB>(r)) << 4) + (nchunk & 1) * 8) = o;
            }
            if constexpr (STEM) {
                const bf16x4 z = {(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
#pragma unroll
                for (int u = 0; u < PASS_C; ++u) {
                    const int hp = u * (256 / CPC) + tid / CPC;
                    const bf16x4 lo = ((lokm >> u) & 1u) ? rxl[u] : z, hi4 = ((hokm >> u) & 1u) ? rxh[u] : z;
                    const bf16x8 o = __builtin_shufflevector(lo, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
                    *reinterpret_cast<bf16x8*>(xT + hp * CB + (((cchunk >> 1) ^ wg_sw<CB>(hp)) << 4) + (cchunk & 1) * 8) = o;
                }
            } else {
#pragma unroll
            for (int u = 0; u < PASS_C; ++u) {
                const int hi = u * (256 / CPC) + tid / CPC;
                bool ok;
                if (TAPS == 9) {
                    const int hy = hi / PW, hx = hi - hy * PW;
                    const int y = y0 + hy - 1, x = x0 + hx - 1;
                    ok = hi < HP && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
                } else { ok = tile * 128 + hi < M; }
                const int hp = TAPS == 9 ? (hi / PW) * PWL + hi % PW : hi;          // pixel index in the LDS image
                if (hi < HP) {
                    bf16x8 o;
                    if (QMODE == PA_LD_BNRELU) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) o[j] = (bf16)fmaxf(fmaf(qk0[j], (float)rx[u][j], qk1[j]), 0.f);
                    } else {
                        o = rx[u];
                    }
                    if (!ok) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) o[j] = (bf16)0.f;
                    }
                    *reinterpret_cast<bf16x8*>(xT + hp * CB + (((cchunk >> 1) ^ wg_sw<CB>(hp)) << 4) + (cchunk & 1) * 8) = o;
                }
            }
            }
        };
        if (split < ntiles) request(split);
        for (int tile = split; tile < ntiles; tile += S) {
            if (tile != split) wg_lds_barrier();          // the previous tile's fragments have been read
            stage(tile);                                   // (the compiler's vmcnt waits sit here, behind a whole MFMA section)
            if (tile + S < ntiles) request(tile + S);      // in flight during this tile's MFMA section
            __builtin_amdgcn_sched_barrier(0);
            wg_lds_barrier();
            mfma_tile();
        }
    } else
    for (int tile = split; tile < ntiles; tile += S) {
        int b = 0, y0 = 0, x0 = 0;
        if (TAPS == 9) {
            int t = tile;
            x0 = (t % tiles_x) * 16; t /= tiles_x;
            y0 = (t % tiles_y) * 8;
            b = t / tiles_y;
        }
        if (tile != split) __syncthreads();          // the previous tile's fragments have been read
        // ---- dy tile [128][NB]
        {
            bf16x8 rp[PASS_N], rq[PASS_N];
            bool ok[PASS_N];
            float pk0[8], pk1[8], pk2[8];
            if (p_lin2) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    pk0[j] = a.dy.k0[n0 + nchunk * 8 + j]; pk1[j] = a.dy.k1[n0 + nchunk * 8 + j]; pk2[j] = a.dy.k2[n0 + nchunk * 8 + j];
                }
            }
            // 3x3: 144 accumulator registers are live -> stage in batches of 2 loads (+2 for LIN2's second operand)
            constexpr int UNP = TAPS == 9 ? 2 : PASS_N;
#pragma unroll
            for (int u0 = 0; u0 < PASS_N; u0 += UNP) {
#pragma unroll
                for (int v = 0; v < UNP; ++v) {
                    const int u = u0 + v;
                    const int r = u * (256 / CPN) + tid / CPN;
                    int m;
                    if (TAPS == 9) { m = (b * a.H + y0 + (r >> 4)) * a.W + x0 + (r & 15); ok[u] = true; }
                    else { m = tile * 128 + r; ok[u] = m < M; }
                    const size_t idx = ok[u] ? (size_t)m * a.Cout + n0 + nchunk * 8 : 0;      // clamped, unconditional
                    rp[u] = PA_WG_LOAD(reinterpret_cast<const bf16x8*>(a.dy.p + idx));
                    if (p_lin2) rq[u] = PA_WG_LOAD(reinterpret_cast<const bf16x8*>(a.dy.q + idx));
                }
#pragma unroll
                for (int v = 0; v < UNP; ++v) {
                    const int u = u0 + v;
                    const int r = u * (256 / CPN) + tid / CPN;
                    bf16x8 o;
                    if (p_lin2) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) o[j] = (bf16)fmaf(pk0[j], (float)rp[u][j], fmaf(pk1[j], (float)rq[u][j], pk2[j]));
                    } else {
                        o = rp[u];
                    }
                    if (!ok[u]) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) o[j] = (bf16)0.f;
                    }
                    if (DB && want_db) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) colsum[j] += (float)o[j];
                    }
                    *reinterpret_cast<bf16x8*>(dyT + r * NB + (((nchunk >> 1) ^ wg_sw<NB>(r)) << 4) + (nchunk & 1) * 8) = o;
                }
            }
        }
        // ---- x tile / halo [HP][CB]
        {
            bf16x8 rx[PASS_C];
            bool ok[PASS_C];
            float qk0[8], qk1[8];
            if (q_bnrelu) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { qk0[j] = a.x.k0[c0 + cchunk * 8 + j]; qk1[j] = a.x.k1[c0 + cchunk * 8 + j]; }
            }
            // stem: (image, output row, output column) of the thread's pixel of pass u, walked from pass to pass (the pixel advances by 256 / CPC
            // per pass) -- two integer divisions per pixel and pass were most of this kernel's instructions (32 per thread and tile)
            int sb = 0, sy = 0, sx = 0;
            if (QMODE == PA_WG_STEM) {
                const int m0 = tile * 128 + tid / CPC, HWo = a.H * a.W;
                sb = m0 / HWo; const int rem = m0 - sb * HWo; sy = rem / a.W; sx = rem - sy * a.W;
            }
#pragma unroll
            for (int u = 0; u < PASS_C; ++u) {
                const int hp = u * (256 / CPC) + tid / CPC;
                int m;
                if (TAPS == 9) {
                    const int hy = hp / PW, hx = hp - hy * PW;
                    const int y = y0 + hy - 1, x = x0 + hx - 1;
                    ok[u] = hp < HP && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
                    m = (b * a.H + y) * a.W + x;
                } else { m = tile * 128 + hp; ok[u] = m < M; }
                if (QMODE == PA_WG_STEM) {
                    // 7x7 stride-2 stem: the 'channels' are the 256 patch elements ky*32 + kx*4 + c of the 4-channel image; one
                    // 16-byte chunk = input pixels (2x + 2q - 3, +1) of input row 2y + ky - 3; a.H / a.W are the OUTPUT dims
                    const int chunk = c0 / 8 + cchunk, ky = chunk >> 2, q = chunk & 3;
                    const int Hin = 2 * a.H, Win = 2 * a.W;
                    const int bb = ok[u] ? sb : 0, y = ok[u] ? sy : 0, x = ok[u] ? sx : 0;
                    sx += 256 / CPC;                 // the next pass's pixel
                    while (sx >= a.W) { sx -= a.W; if (++sy >= a.H) { sy = 0; ++sb; } }
                    const int yi = 2 * y + ky - 3, xi = 2 * x + 2 * q - 3;
                    const bool rowok = ok[u] && ky < 7 && (unsigned)yi < (unsigned)Hin;
                    const bool lok = rowok && (unsigned)xi < (unsigned)Win, hok = rowok && (unsigned)(xi + 1) < (unsigned)Win;
                    const bf16* rowp = a.x.p + ((size_t)bb * Hin + (rowok ? yi : 0)) * Win * 4;
                    bf16x4 lo = *reinterpret_cast<const bf16x4*>(rowp + (size_t)(lok ? xi : 0) * 4);          // clamped, unconditional
                    bf16x4 hi = *reinterpret_cast<const bf16x4*>(rowp + (size_t)(hok ? xi + 1 : 0) * 4);
                    const bf16x4 z = {(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
                    if (!lok) lo = z;
                    if (!hok) hi = z;
                    rx[u] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                    ok[u] = true;              // (zero padding already applied)
                    continue;
                }
                const size_t idx = ok[u] ? (size_t)m * a.Cin + c0 + cchunk * 8 : 0;
                rx[u] = PA_WG_LOAD(reinterpret_cast<const bf16x8*>(a.x.p + idx));
            }
#pragma unroll
            for (int u = 0; u < PASS_C; ++u) {
                const int hi = u * (256 / CPC) + tid / CPC;
                const int hp = TAPS == 9 ? (hi / PW) * PWL + hi % PW : hi;          // pixel index in the LDS image
                if (hi < HP) {
                    bf16x8 o;
                    if (q_bnrelu) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) o[j] = (bf16)fmaxf(fmaf(qk0[j], (float)rx[u][j], qk1[j]), 0.f);
                    } else {
                        o = rx[u];
                    }
                    if (!ok[u]) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) o[j] = (bf16)0.f;
                    }
                    *reinterpret_cast<bf16x8*>(xT + hp * CB + (((cchunk >> 1) ^ wg_sw<CB>(hp)) << 4) + (cchunk & 1) * 8) = o;
                }
            }
        }
        __syncthreads();
        // ---- MFMA: 4 steps of 32 pixels
        mfma_tile();
    }

    // ---- partial slab part[split][n][tap*Cin + c]: D rows = c (4 consecutive per lane), columns = n
    float* slab = a.part + (size_t)split * a.Cout * Kfull;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        const int n = n0 + (wn * NF + f) * 16 + (lane & 15);
#pragma unroll
        for (int t = 0; t < TAPS; ++t)
#pragma unroll
            for (int cf = 0; cf < CF; ++cf) {
                const int c = c0 + (wc * CF + cf) * 16 + (lane >> 4) * 4;
                PA_WG_STORE(reinterpret_cast<f32x4*>(slab + (size_t)n * Kfull + t * a.Cin + c), acc[t][cf][f]);
            }
    }
    if (DB && want_db) {
        // column sums of dy: threads with the same chunk (tid % CPN) hold partial sums over their rows
        __syncthreads();
        float* red = reinterpret_cast<float*>(lds);          // [256][8]
#pragma unroll
        for (int j = 0; j < 8; ++j) red[tid * 8 + j] = colsum[j];
        __syncthreads();
        if (tid < NB) {
            const int ch = tid >> 3, j = tid & 7;
            float s = 0.f;
            for (int r = ch; r < 256; r += CPN) s += red[r * 8 + j];
            a.dbpart[(size_t)split * a.Cout + n0 + tid] = s;
        }
    }
}

// one layer per launch: grid (splits, n-blocks, c-blocks)
template <int TAPS, int NF, int CF, int WNW, int PMODE, int QMODE, bool DB, bool PIPE = false, int OCC = 1>
__global__ __launch_bounds__(256, PIPE ? OCC : 2) void wgrad_tile_kernel(PaWgradArgs a, int ntiles) {
    __shared__ __attribute__((aligned(16))) bf16 lds[wg_lds_elems<TAPS, NF, CF, WNW>()];
    wgrad_tile_body<TAPS, NF, CF, WNW, PMODE, QMODE, DB, PIPE>(a, ntiles, (int)blockIdx.x, (int)gridDim.x, (int)blockIdx.y, (int)blockIdx.z, lds);
}

// Several INDEPENDENT weight gradients in ONE launch (the three or four of a residual block, reference models/asn_stacked_hg.py:17-24,
// together with whatever else was waiting for the same event: the head's linear / out_conv / forth_conv / in_conv layers).  A launch per
// layer spreads each layer over every CU as many short splits (one fp32 slab per workgroup) and runs the layers one after the other, each
// with its own ramp and tail; here the layers run SIDE BY SIDE on a share of the CUs each, with few, long splits -- a third of the slab
// bytes, and the launch lasts as long as its longest job instead of the sum.  One body per tile shape (kind), operand modes read at run time.
// Plain (two workgroups per CU, <= 244 registers) rather than pipelined bodies: 6.27 vs 6.31 ms per step -- a pipelined workgroup owns its
// CU's registers, and the main chain's kernels then find no room beside the group.
#define PA_WG_MAXJOBS 8
#define PA_WG_RED_WGS_DEFAULT 256
enum { PA_WGK_9_44 = 0, PA_WGK_9_44L, PA_WGK_1_44, PA_WGK_1_42, PA_WGK_1_24, PA_WGK_1_22, PA_WGK_N };      // 3x3 64 x 64 (n x c) with a plain / BatchNorm-backward dy operand (fixed modes: the 144-accumulator body has no registers to spare for both) | 1x1 128 x 128 | 128 x 64 | 64 x 128 | 64 x 64
struct PaWgradGroup {
    PaWgradArgs a[PA_WG_MAXJOBS];
    int ntiles[PA_WG_MAXJOBS], S[PA_WG_MAXJOBS], ny[PA_WG_MAXJOBS], kind[PA_WG_MAXJOBS];
    int begin[PA_WG_MAXJOBS + 1];              // first workgroup of job j (begin[njobs] = first workgroup behind the weight-gradient jobs)
    int njobs;
    // one more job kind (round 6): the slab reduction of EARLIER launches' layers (red_n entries of the reduce table, red_wgs workgroups at the
    // end of the grid) -- 16 reduction launches per step on the queue that decides the tail of the backward pass become part of the group
    // launches that follow them; same body, same (split) order of every sum as the launch of its own
    const PaWgradReduceJob* red_jobs;
    PaRedList red;
    int red_n, red_wgs;
};

__global__ __launch_bounds__(256, 2) void wgrad_group_kernel(PaWgradGroup g) {
    constexpr int L9 = wg_lds_elems<9, 4, 1, 1>(), L1 = wg_lds_elems<1, 4, 4, 2>();
    __shared__ __attribute__((aligned(16))) bf16 lds[L9 > L1 ? L9 : L1];
    const int id = (int)blockIdx.x;
    if (id >= g.begin[g.njobs]) {                  // the reduction job
        const int local = id - g.begin[g.njobs];
        for (int i = 0; i < g.red_n; ++i) wgrad_reduce_body(g.red_jobs[g.red.idx[i]], local, g.red_wgs);
        return;
    }
    int j = 0;
#pragma unroll
    for (int k = 1; k < PA_WG_MAXJOBS; ++k) if (k < g.njobs && id >= g.begin[k]) j = k;
    const int local = id - g.begin[j];
    const int S = g.S[j], split = local % S, rest = local / S, by = rest % g.ny[j], bz = rest / g.ny[j];
    const PaWgradArgs& a = g.a[j];
    const int nt = g.ntiles[j];
    switch (g.kind[j]) {
        case PA_WGK_9_44: wgrad_tile_body<9, 4, 1, 1, PA_LD_PLAIN, PA_LD_BNRELU, false, false>(a, nt, split, S, by, bz, lds); break;
        case PA_WGK_9_44L: wgrad_tile_body<9, 4, 1, 1, PA_LD_LIN2, PA_LD_BNRELU, false, false>(a, nt, split, S, by, bz, lds); break;
        case PA_WGK_1_44: wgrad_tile_body<1, 4, 4, 2, PA_WG_RT, PA_WG_RT, true, false>(a, nt, split, S, by, bz, lds); break;
        case PA_WGK_1_42: wgrad_tile_body<1, 4, 2, 2, PA_WG_RT, PA_WG_RT, true, false>(a, nt, split, S, by, bz, lds); break;
        case PA_WGK_1_24: wgrad_tile_body<1, 2, 4, 2, PA_WG_RT, PA_WG_RT, true, false>(a, nt, split, S, by, bz, lds); break;
        default: wgrad_tile_body<1, 2, 2, 2, PA_WG_RT, PA_WG_RT, true, false>(a, nt, split, S, by, bz, lds); break;
    }
}

// ------------------------------------------------------------------------------------------------ host side
struct WgTileCfg { int nb, cb, ntiles, splits; };

static bool wg_tile_cfg(int B, int H, int W, int Cin, int Cout, int taps, WgTileCfg& c) {
    static int off = -1, target1 = 0, target9 = 0;
    if (off < 0) {
        off = pa_getenv("PA_WGRAD_OLD") ? 1 : 0;
        const char* e = pa_getenv("PA_WGRAD_WGS1"); target1 = e ? atoi(e) : 256;
        e = pa_getenv("PA_WGRAD_WGS9"); target9 = e ? atoi(e) : 256;
    }
    if (off || H <= 0 || W <= 0) return false;
    const int M = B * H * W;
    if (taps == 9) {
        if (Cin % 64 || Cout % 64 || H % 8 || W % 16) return false;
        c.nb = 64; c.cb = 64; c.ntiles = B * (H / 8) * (W / 16);
        if (c.ntiles < 48) return false;
    } else if (taps == 1) {
        if (Cin % 64 || Cout % 64) return false;
        c.nb = Cout % 128 == 0 ? 128 : 64; c.cb = Cin % 128 == 0 ? 128 : 64; c.ntiles = (M + 127) / 128;
        static int min1 = -1;
        if (min1 < 0) min1 = pa_getenv("PA_WGRAD_MIN1") ? atoi(pa_getenv("PA_WGRAD_MIN1")) : 3;      // also the 16x16 ... 4x4 levels: 6.90 vs 6.97 ms (96)
        if (c.ntiles < min1) return false;
    } else return false;
    const int types = (Cout / c.nb) * (Cin / c.cb);
    int s = (taps == 9 ? target9 : target1) / types;
    if (s < 1) s = 1;
    if (s > c.ntiles) s = c.ntiles;
    int per = (c.ntiles + s - 1) / s;                // tiles per workgroup
    // fewer, longer splits: a pipelined workgroup (PIPE above) runs at its MFMA / per-CU load rate from the second tile on, so a launch
    // of few workgroups with many tiles each takes about as long as one that spreads the same tiles over every CU -- on a fraction of
    // the CUs and with that fraction of the fp32 slabs (at 32x32 and below the slabs were 3 - 9 x the operand bytes)
    static int minper1 = -1, minper9 = -1;
    if (minper1 < 0) {
        const char* e = pa_getenv("PA_WGRAD_MINPER"); const int both = e ? atoi(e) : 0;
        e = pa_getenv("PA_WGRAD_MINPER1"); minper1 = e ? atoi(e) : (both ? both : PA_WGRAD_MINPER1_DEFAULT);
        e = pa_getenv("PA_WGRAD_MINPER9"); minper9 = e ? atoi(e) : (both ? both : PA_WGRAD_MINPER9_DEFAULT);
    }
    const int minper = taps == 9 ? minper9 : minper1;
    if (per < minper) per = minper;
    if (per > c.ntiles) per = c.ntiles;
    c.splits = (c.ntiles + per - 1) / per;           // balanced
    return true;
}

int pa_wgrad_tile_splits(int B, int H, int W, int Cin, int Cout, int taps) {
    WgTileCfg c;
    return wg_tile_cfg(B, H, W, Cin, Cout, taps, c) ? c.splits : 0;
}

// Launch flags of the tile weight gradients (hip_ext.h): Net::flush_wgrads sets hipExtAnyOrderLaunch for the 2nd, 3rd ... launch of a
// group of INDEPENDENT weight gradients behind one event.  Measured on gfx950 / ROCm 7.2 (tools/ubench/anyorder.hip): the flag never
// overlaps two kernels of a stream, but the next one starts the moment the previous one's last wave ends instead of 3.6 us later (256
// workgroups) -- the completion signal / cache write-back / acquire round trip between two kernels that share no data.
static thread_local unsigned g_wt_launch_flags = 0;
void pa_wgrad_set_launch_flags(unsigned flags) { g_wt_launch_flags = flags; }

template <int TAPS, int NF, int CF, int WNW, bool PIPE, int OCC = 1>
static void launch_wt_modes2(const PaWgradArgs& a, dim3 grid, int ntiles, hipStream_t st) {
    const bool lin2 = a.dy.mode == PA_LD_LIN2, bnrelu = a.x.mode == PA_LD_BNRELU;
    constexpr bool DB = TAPS == 1;          // bias gradients: only convs without a BatchNorm behind them (all 1x1 here)
    const unsigned fl = g_wt_launch_flags;
    auto go = [&](auto kernel) {
        if (fl) hipExtLaunchKernelGGL(kernel, grid, dim3(256), 0, st, nullptr, nullptr, fl, a, ntiles);
        else hipLaunchKernelGGL(kernel, grid, dim3(256), 0, st, a, ntiles);
    };
    if (lin2 && bnrelu) go(wgrad_tile_kernel<TAPS, NF, CF, WNW, PA_LD_LIN2, PA_LD_BNRELU, DB, PIPE, OCC>);
    else if (lin2) go(wgrad_tile_kernel<TAPS, NF, CF, WNW, PA_LD_LIN2, PA_LD_PLAIN, DB, PIPE, OCC>);
    else if (bnrelu) go(wgrad_tile_kernel<TAPS, NF, CF, WNW, PA_LD_PLAIN, PA_LD_BNRELU, DB, PIPE, OCC>);
    else go(wgrad_tile_kernel<TAPS, NF, CF, WNW, PA_LD_PLAIN, PA_LD_PLAIN, DB, PIPE, OCC>);
}

// the pipelined form wherever a workgroup walks over at least two tiles (one tile: nothing to prefetch, and the plain form's two
// workgroups per CU are the better shape for launches with more workgroups than CUs)
template <int TAPS, int NF, int CF, int WNW>
static void launch_wt_modes(const PaWgradArgs& a, dim3 grid, int ntiles, hipStream_t st) {
    static int nopipe = -1, occ = 1;
    if (nopipe < 0) { nopipe = pa_getenv("PA_WGRAD_NOPIPE") ? 1 : 0; const char* e = pa_getenv("PA_WGRAD_PIPE_OCC"); occ = e ? atoi(e) : 1; }
    if (!nopipe && ntiles >= 2 * (int)grid.x) {
        if (occ == 2 && TAPS == 1) launch_wt_modes2<TAPS, NF, CF, WNW, true, TAPS == 1 ? 2 : 1>(a, grid, ntiles, st);      // (the 3x3 form spills at 256 registers)
        else launch_wt_modes2<TAPS, NF, CF, WNW, true, 1>(a, grid, ntiles, st);
    } else launch_wt_modes2<TAPS, NF, CF, WNW, false>(a, grid, ntiles, st);
}

// 7x7/2 stem weight gradient on the tile kernel: NB = 64 output channels x CB = 256 patch elements per workgroup, dy read once
int pa_launch_stem_wgrad_tile(const PaWgradArgs& a, hipStream_t st) {
    static int off = -1;
    if (off < 0) off = (pa_getenv("PA_WGRAD_OLD") || pa_getenv("PA_STEM_WGRAD_OLD")) ? 1 : 0;
    const int M = a.B * a.H * a.W, ntiles = (M + 127) / 128;
    if (off || a.Cin != 256 || a.Cout != 64 || a.splits > ntiles) return -1;
    dim3 grid(a.splits, 1, 1);
    // the pipelined form needs the CU to itself (297 registers): with the stem's 512 splits that is two rounds of workgroups, 95 us
    // against 45 us for the plain form at two workgroups per CU (measured in the step, where this launch is the tail everything waits
    // for); it pays at <= 256 splits only
    static int pipe = -1;
    if (pipe < 0) { const char* e = pa_getenv("PA_STEM_PIPE"); pipe = e ? atoi(e) : 0; if (pa_getenv("PA_WGRAD_NOPIPE")) pipe = 0; }
    if (pipe && ntiles >= 2 * a.splits) {
        if (a.dy.mode == PA_LD_LIN2) hipLaunchKernelGGL((wgrad_tile_kernel<1, 2, 8, 2, PA_LD_LIN2, PA_WG_STEM, false, true>), grid, dim3(256), 0, st, a, ntiles);
        else hipLaunchKernelGGL((wgrad_tile_kernel<1, 2, 8, 2, PA_LD_PLAIN, PA_WG_STEM, false, true>), grid, dim3(256), 0, st, a, ntiles);
        return (int)hipGetLastError();
    }
    if (a.dy.mode == PA_LD_LIN2) hipLaunchKernelGGL((wgrad_tile_kernel<1, 2, 8, 2, PA_LD_LIN2, PA_WG_STEM, false>), grid, dim3(256), 0, st, a, ntiles);
    else hipLaunchKernelGGL((wgrad_tile_kernel<1, 2, 8, 2, PA_LD_PLAIN, PA_WG_STEM, false>), grid, dim3(256), 0, st, a, ntiles);
    return (int)hipGetLastError();
}

// returns -1 when the shape is not handled here (caller falls back to conv_wgrad.hip)
int pa_launch_wgrad_tile(const PaWgradArgs& a, hipStream_t st) {
    WgTileCfg c;
    if (!wg_tile_cfg(a.B, a.H, a.W, a.Cin, a.Cout, a.taps, c) || a.splits < 1 || a.splits > c.ntiles) return -1;
    if (a.taps == 9 && a.dbpart) return -1;
    c.splits = a.splits;                 // (the layer's slab was laid out for this count: wg_tile_cfg's own, or the grouped policy's)
    dim3 grid(c.splits, a.Cout / c.nb, a.Cin / c.cb);
    if (a.taps == 9) launch_wt_modes<9, 4, 1, 1>(a, grid, c.ntiles, st);
    else if (c.nb == 128 && c.cb == 128) launch_wt_modes<1, 4, 4, 2>(a, grid, c.ntiles, st);
    else if (c.nb == 128) launch_wt_modes<1, 4, 2, 2>(a, grid, c.ntiles, st);
    else if (c.cb == 128) launch_wt_modes<1, 2, 4, 2>(a, grid, c.ntiles, st);
    else launch_wt_modes<1, 2, 2, 2>(a, grid, c.ntiles, st);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ grouped launch
static int wg_group_on() {
    static int on = -1;
    if (on < 0) { const char* e = pa_getenv("PA_WG_GROUP"); on = e ? atoi(e) : PA_WG_GROUP_DEFAULT; }
    return on;
}

static int wg_kind(const WgTileCfg& c, int taps, int pmode = PA_LD_PLAIN) {
    if (taps == 9) return (c.nb == 64 && c.cb == 64) ? (pmode == PA_LD_LIN2 ? PA_WGK_9_44L : PA_WGK_9_44) : -1;
    if (c.nb == 128 && c.cb == 128) return PA_WGK_1_44;
    if (c.nb == 128 && c.cb == 64) return PA_WGK_1_42;
    if (c.nb == 64 && c.cb == 128) return PA_WGK_1_24;
    if (c.nb == 64 && c.cb == 64) return PA_WGK_1_22;
    return -1;
}

// split count of a layer whose weight gradient is launched in a group (0: the shape is not one the grouped kernel takes -- the layer
// keeps pa_wgrad_splits' count and its own launch): a WORKGROUP budget per job (64 each: three jobs of a residual block stay below the
// 256-workgroup cap of a group launch, Net::flush_wgrads -- 5.91 vs 5.98 ms for 96 / 128 without a cap), at least minper tiles per workgroup
int pa_wgrad_group_splits(int B, int H, int W, int Cin, int Cout, int taps) {
    if (!wg_group_on() || H <= 0 || W <= 0) return 0;
    WgTileCfg c;
    if (!wg_tile_cfg(B, H, W, Cin, Cout, taps, c) || wg_kind(c, taps) < 0) return 0;
    static int wg9 = -1, wg1 = -1, mp9 = -1, mp1 = -1;
    if (wg9 < 0) {
        const char* e = pa_getenv("PA_WG_GROUP_WGS9"); wg9 = e ? atoi(e) : 64;
        e = pa_getenv("PA_WG_GROUP_WGS1"); wg1 = e ? atoi(e) : 64;
        e = pa_getenv("PA_WG_GROUP_MINPER9"); mp9 = e ? atoi(e) : 4;
        e = pa_getenv("PA_WG_GROUP_MINPER1"); mp1 = e ? atoi(e) : 2;
    }
    const int types = (Cout / c.nb) * (Cin / c.cb);
    int s = (taps == 9 ? wg9 : wg1) / types;
    const int minper = taps == 9 ? mp9 : mp1;
    if (s * minper > c.ntiles) s = c.ntiles / minper;
    if (s < 1) s = 1;
    const int per = (c.ntiles + s - 1) / s;
    return (c.ntiles + per - 1) / per;               // balanced
}

// can this launch be a job of a group (its slab laid out with a grouped split count, tile shape known to the grouped kernel)?
bool pa_wgrad_group_takes(const PaWgradArgs& a) {
    if (!wg_group_on() || (a.taps == 9 && a.dbpart)) return false;
    if (a.dy.mode != PA_LD_PLAIN && a.dy.mode != PA_LD_LIN2) return false;
    if (a.x.mode != PA_LD_PLAIN && a.x.mode != PA_LD_BNRELU) return false;
    if (a.taps == 9 && a.x.mode != PA_LD_BNRELU) return false;          // (the 3x3 bodies are compiled for a BatchNorm+ReLU x operand: every conv2 of the networks)
    WgTileCfg c;
    if (!wg_tile_cfg(a.B, a.H, a.W, a.Cin, a.Cout, a.taps, c) || wg_kind(c, a.taps) < 0) return false;
    return a.splits >= 1 && a.splits <= c.ntiles;
}

int pa_wgrad_job_workgroups(const PaWgradArgs& a) {
    WgTileCfg c;
    if (!wg_tile_cfg(a.B, a.H, a.W, a.Cin, a.Cout, a.taps, c)) return 0;
    return a.splits * (a.Cout / c.nb) * (a.Cin / c.cb);
}

// n <= PA_WG_MAXJOBS launches that pa_wgrad_group_takes() admitted, in one launch (the longest jobs first: their workgroups start first)
int pa_launch_wgrad_group(const PaWgradArgs* const* as, int n, hipStream_t st, bool keep_order, const PaWgradReduceJob* red_jobs, const int* red_idx, int red_n) {
    if (n < 1 || n > PA_WG_MAXJOBS) { pa_set_error_msg("pa_launch_wgrad_group: 1 .. 8 jobs"); return 1; }
    if (red_n < 0 || red_n > PA_RED_LIST_MAX || (red_n > 0 && (!red_jobs || !red_idx))) { pa_set_error_msg("pa_launch_wgrad_group: bad reduction list"); return 1; }
    PaWgradGroup g;
    memset(&g, 0, sizeof g);
    int order[PA_WG_MAXJOBS]; long work[PA_WG_MAXJOBS];
    WgTileCfg cs[PA_WG_MAXJOBS];
    for (int j = 0; j < n; ++j) {
        if (!pa_wgrad_group_takes(*as[j])) { pa_set_error_msg("pa_launch_wgrad_group: a job the grouped kernel does not take"); return 1; }
        wg_tile_cfg(as[j]->B, as[j]->H, as[j]->W, as[j]->Cin, as[j]->Cout, as[j]->taps, cs[j]);
        order[j] = j;
        work[j] = (long)((cs[j].ntiles + as[j]->splits - 1) / as[j]->splits) * (as[j]->taps == 9 ? 3 : 2);      // tiles per workgroup x relative tile cost
    }
    for (int i = 1; i < n && !keep_order; ++i)       // insertion sort, descending work
        for (int k = i; k > 0 && work[order[k]] > work[order[k - 1]]; --k) { const int t = order[k]; order[k] = order[k - 1]; order[k - 1] = t; }
    int begin = 0;
    for (int jj = 0; jj < n; ++jj) {
        const int j = order[jj];
        g.a[jj] = *as[j]; g.ntiles[jj] = cs[j].ntiles; g.S[jj] = as[j]->splits; g.ny[jj] = as[j]->Cout / cs[j].nb; g.kind[jj] = wg_kind(cs[j], as[j]->taps, as[j]->dy.mode);
        g.begin[jj] = begin;
        begin += as[j]->splits * (as[j]->Cout / cs[j].nb) * (as[j]->Cin / cs[j].cb);
    }
    g.begin[n] = begin; g.njobs = n;
    if (red_n > 0) {
        static int rw = -1;
        if (rw < 0) { const char* e = pa_getenv("PA_WG_RED_WGS"); rw = e ? atoi(e) : PA_WG_RED_WGS_DEFAULT; if (rw < 1) rw = 1; }
        g.red_jobs = red_jobs; g.red_n = red_n; g.red_wgs = rw;
        for (int i = 0; i < PA_RED_LIST_MAX; ++i) g.red.idx[i] = red_idx[i < red_n ? i : 0];
        begin += rw;
    }
    hipLaunchKernelGGL(wgrad_group_kernel, dim3(begin), dim3(256), 0, st, g);
    return (int)hipGetLastError();
}
