// Implicit-GEMM convolution (1x1 and 3x3 'same', stride 1) on bf16 MFMA for gfx950.
//
//   out[m][n] = epilogue( sum_{tap,c} load(in)[pixel(m)+tap][c] * w[n][tap][c] + bias[n] + add1 + add2 )
//
// m = flattened NHWC pixel (B*H*W rows), n = output channel.  The same kernel serves
//   * forward convs   (reference models/asn_stacked_hg.py:17-24, 241-248): the input's pending
//     BatchNorm+ReLU is applied while the tile is staged (PA_LD_BNRELU), the residual add and the
//     per-channel sum / sum^2 needed by the NEXT BatchNorm are done in the epilogue (PA_OUT_STATS);
//   * data gradients  (dgrad): `in` is the output gradient with the BatchNorm backward applied on
//     load (PA_LD_LIN2), `w` is the transposed/tap-flipped weight copy, and the epilogue masks by the
//     ReLU of the tensor the gradient belongs to and accumulates the two BatchNorm-backward
//     reductions (PA_OUT_BWD).
//
// Tiling: BMxBN output tile per 256-thread workgroup (4 waves as 2x2), K step 64.  Both operand
// tiles are staged global -> registers (transform) -> LDS as [row][64] bf16 with the 16-byte chunk
// index XOR-swizzled by (row & 7), so the ds_read_b128 fragment reads of
// v_mfma_f32_16x16x32_bf16 are bank-conflict free.  The weight fragment is the MFMA A operand and
// the activation fragment the B operand, so each lane ends up with 4 consecutive output channels of
// one pixel (8-byte bf16x4 stores, per-channel reductions over the 16 pixel lanes by DPP shuffles).
#include "common.h"
#include "kernels.h"
#include "common.h"
// cycle stamps (tuning builds, a.dbg & 64): entry / loads issued / constants (finalize prologue) / staged / barrier / K loop / epilogue parts
PA_STAMP_DECL(pa_conv1_clk, pa_debug_conv1_clocks)
#define PA_STAMP1(i) PA_STAMP_AT(pa_conv1_clk, a.dbg & 64, i)
#define PA_EPI_STAMP(i) PA_STAMP1(i)
#include "conv_epilogue.h"

// STEM: the A operand is the 4-channel-padded input image and the kernel computes the 7x7 stride-2
// stem conv (reference models/asn_stacked_hg.py:223) as a K=256 GEMM: k = ky*32 + kx*4 + c, i.e. one
// 16-byte chunk = 2 horizontally adjacent input pixels; a.H/a.W are the OUTPUT dims.
// FIN: the instance that can carry the input's BatchNorm finalize in its prologue (bn_fin.h) -- the 64 x 64 tiles of the small maps only;
// every other instance is the plain kernel, its register allocation untouched
template <int BM, int BN, int LDMODE, int TAPS, bool STEM = false, bool FIN = false>
__global__ __launch_bounds__(256, FIN ? 2 : 1) void conv_igemm_kernel(PaConvArgs a) {      // (FIN: left alone the prologue takes every register it can get -- 256, one workgroup per CU; the plain 64 x 64 instance runs at 116)
    constexpr int AI = BM / 32, BI = BN / 32;      // staged 16-byte chunks per thread (A / B tile)
    constexpr int MI = BM / 32, NI = BN / 32;      // 16x16 fragments per wave (wave tile BM/2 x BN/2)
    __shared__ __attribute__((aligned(16))) bf16 lds[(BM + BN) * 64];
    __shared__ float kst[LDMODE == PA_LD_PLAIN ? 4 : 3 * 512 + (FIN ? 2048 : 0)];      // per-channel constants of the input transform (+ the finalize prologue's scratch)
    bf16* As = lds;
    bf16* Bs = lds + BM * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int M = a.B * a.H * a.W, HW = a.H * a.W;
    const int K = TAPS * a.Cin;
    const int nK = K / 64;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int cc = tid & 7, r = tid >> 3;

    int am[AI], ay[AI], ax[AI], ab[AI];
#pragma unroll
    for (int i = 0; i < AI; ++i) {
        int m = m0 + r + 32 * i;
        am[i] = m < M ? m : -1;
        ab[i] = m / HW;
        int rem = m - ab[i] * HW;
        ay[i] = rem / a.W;
        ax[i] = rem - ay[i] * a.W;
    }

    bf16x8 ra[AI], rq[AI], rb[BI];
    unsigned okmask = 0;
    int cur_c = 0;

    auto gload = [&](int kt) {
        if (STEM) {
            const int cidx = kt * 8 + cc, ky = cidx >> 2, q = cidx & 3;
            const int Hin = 2 * a.H, Win = 2 * a.W;
            okmask = 0;
#pragma unroll
            for (int i = 0; i < AI; ++i) {
                bf16x4 lo = {(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f}, hi = lo;
                const int yi = 2 * ay[i] + ky - 3, xi = 2 * ax[i] + 2 * q - 3;
                if (am[i] >= 0 && ky < 7 && (unsigned)yi < (unsigned)Hin) {
                    const bf16* rowp = a.in.p + ((size_t)ab[i] * Hin + yi) * Win * 4;
                    if ((unsigned)xi < (unsigned)Win) lo = *reinterpret_cast<const bf16x4*>(rowp + (size_t)xi * 4);
                    if ((unsigned)(xi + 1) < (unsigned)Win) hi = *reinterpret_cast<const bf16x4*>(rowp + (size_t)(xi + 1) * 4);
                }
                ra[i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                okmask |= 1u << i;
            }
#pragma unroll
            for (int i = 0; i < BI; ++i)
                rb[i] = *reinterpret_cast<const bf16x8*>(a.w + (size_t)(n0 + r + 32 * i) * K + kt * 64 + cc * 8);
            return;
        }
        int tap = 0, c0 = kt * 64, dy = 0, dx = 0;
        if (TAPS == 9) { tap = c0 / a.Cin; c0 -= tap * a.Cin; dy = tap / 3 - 1; dx = tap - (tap / 3) * 3 - 1; }
        cur_c = c0 + cc * 8;
        okmask = 0;
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            bool ok = am[i] >= 0;
            if (TAPS == 9) ok = ok && (unsigned)(ay[i] + dy) < (unsigned)a.H && (unsigned)(ax[i] + dx) < (unsigned)a.W;
            if (ok) {
                size_t idx = (size_t)(am[i] + dy * a.W + dx) * a.Cin + cur_c;
                ra[i] = *reinterpret_cast<const bf16x8*>(a.in.p + idx);
                if (LDMODE == PA_LD_LIN2) rq[i] = *reinterpret_cast<const bf16x8*>(a.in.q + idx);
                okmask |= 1u << i;
            }
        }
#pragma unroll
        for (int i = 0; i < BI; ++i)
            rb[i] = *reinterpret_cast<const bf16x8*>(a.w + (size_t)(n0 + r + 32 * i) * K + kt * 64 + cc * 8);
    };

    auto lstore = [&]() {
        float k0[8], k1[8], k2[8];
        if (LDMODE != PA_LD_PLAIN) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { k0[j] = kst[cur_c + j]; k1[j] = kst[512 + cur_c + j]; }
            if (LDMODE == PA_LD_LIN2) {
#pragma unroll
                for (int j = 0; j < 8; ++j) k2[j] = kst[1024 + cur_c + j];
            }
        }
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            int row = r + 32 * i;
            bf16x8 o;
            if (okmask & (1u << i)) {
                if (LDMODE == PA_LD_PLAIN) {
                    o = ra[i];
                } else if (LDMODE == PA_LD_BNRELU) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] = (bf16)fmaxf(fmaf(k0[j], (float)ra[i][j], k1[j]), 0.f);
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        o[j] = (bf16)fmaf(k0[j], (float)ra[i][j], fmaf(k1[j], (float)rq[i][j], k2[j]));
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (bf16)0.f;
            }
            *reinterpret_cast<bf16x8*>(As + row * 64 + ((cc ^ (row & 7)) << 3)) = o;
        }
#pragma unroll
        for (int i = 0; i < BI; ++i) {
            // weight rows are STORED permuted (conv_epilogue.h: pa_lds_row_of_weight_row) so that a lane's fragments
            // hold 8 CONSECUTIVE channels per 32-channel chunk (16-byte epilogue accesses) while the fragment reads
            // stay conflict-free
            const int lrow = pa_lds_row_of_weight_row<BN, NI>(r + 32 * i);
            *reinterpret_cast<bf16x8*>(Bs + lrow * 64 + ((cc ^ (lrow & 7)) << 3)) = rb[i];
        }
    };

    f32x4 acc[NI][MI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};

    gload(0);
    // per-channel constants of the input transform, behind the first tile's loads (they are in flight meanwhile)
    bool fin_done = false;
    if constexpr (FIN && LDMODE != PA_LD_PLAIN) {
        if (a.fin.rows > 0) {
            // the input's BatchNorm finalize (forward: scale / shift; backward: kA / kB / kC) from the producer's partial rows, bn_fin.h
            pa_bn_fin_prologue<256, 512>(a.fin, a.Cin, kst, kst + 3 * 512, blockIdx.x == 0 && blockIdx.y == 0);
            fin_done = true;
        }
    }
    if (LDMODE != PA_LD_PLAIN && !fin_done) {
        for (int c = threadIdx.x; c < a.Cin; c += 256) {
            kst[c] = a.in.k0[c]; kst[512 + c] = a.in.k1[c];
            if (LDMODE == PA_LD_LIN2) kst[1024 + c] = a.in.k2[c];
        }
    }
    __syncthreads();            // kst visible
    lstore();
    __syncthreads();

    const int frow = lane & 15, fchk = lane >> 4;
    for (int kt = 0; kt < nK; ++kt) {
        if (kt + 1 < nK) gload(kt + 1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 fa[MI], fw[NI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                int row = wm * (BM / 2) + mi * 16 + frow;
                fa[mi] = *reinterpret_cast<const bf16x8*>(As + row * 64 + (((fchk + 4 * kk) ^ (row & 7)) << 3));
            }
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                int row = wn * (BN / 2) + ni * 16 + frow;
                fw[ni] = *reinterpret_cast<const bf16x8*>(Bs + row * 64 + (((fchk + 4 * kk) ^ (row & 7)) << 3));
            }
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
                    acc[ni][mi] = PA_MFMA_16x16x32(fw[ni], fa[mi], acc[ni][mi]);
        }
        __syncthreads();
        if (kt + 1 < nK) {
            lstore();
            __syncthreads();
        }
    }

    // ---------------------------------------------------------------- epilogue (conv_epilogue.h)
    pa_conv_epilogue_auto<BN, NI, MI>(a, acc, n0, wm, wn,
                                     [&](int wr, int mi, int p) { const int m = m0 + wr * (BM / 2) + mi * 16 + p; return m < M ? m : -1; },
                                     reinterpret_cast<float*>(lds), (int)blockIdx.x);
}

// ------------------------------------------------------------------------------------------------------------------
// ONE-SHOT 1x1 kernel for the small maps (64 x 64 tiles, Cin = 64 * NK <= 256): the whole K range of both operand tiles is requested
// at once (NK x 4 ... 6 sixteen-byte loads per thread in flight), transformed and staged into LDS in one pass, and the K loop runs
// without a global load or a barrier.  The generic kernel above walks K in 64-channel steps with a dependent memory round trip and
// two barriers per step: at 16 x 16 and below that chain of 2 - 4 round trips IS the kernel (6.5 - 7.8 us for 0.1 - 0.8 GFLOP).
// Same tile, same fragment layout, same epilogue and weight-row permutation as conv_igemm_kernel<64, 64, ...>: same results bit for bit
// (the MFMA accumulation order over K is unchanged).
template <int NK, int LDMODE, bool FIN>
__global__ __launch_bounds__(256, 2) void conv1x1_oneshot_kernel(PaConvArgs a) {
    PA_SET_MAIN_PRIO();
    PA_STAMP1(0);
    constexpr int BM = 64, BN = 64, AI = 2, BI = 2, MI = 2, NI = 2;
    __shared__ __attribute__((aligned(16))) bf16 lds[(BM + BN) * 64 * NK];
    __shared__ float kst[LDMODE == PA_LD_PLAIN ? 4 : 3 * 512 + (FIN ? 2048 : 0)];
    bf16* As = lds;
    bf16* Bs = lds + NK * BM * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int M = a.B * a.H * a.W;
    const int K = NK * 64;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int cc = tid & 7, r = tid >> 3;

    // ---- every load of the thread first
    bf16x8 xa[NK][AI], xq[NK][AI], xb[NK][BI];
    bool ok[AI];
#pragma unroll
    for (int i = 0; i < AI; ++i) ok[i] = m0 + r + 32 * i < M;
#pragma unroll
    for (int kt = 0; kt < NK; ++kt) {
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            const size_t idx = ok[i] ? (size_t)(m0 + r + 32 * i) * K + kt * 64 + cc * 8 : 0;       // clamped, unconditional
            xa[kt][i] = *reinterpret_cast<const bf16x8*>(a.in.p + idx);
            if (LDMODE == PA_LD_LIN2) xq[kt][i] = *reinterpret_cast<const bf16x8*>(a.in.q + idx);
        }
#pragma unroll
        for (int i = 0; i < BI; ++i)
            xb[kt][i] = *reinterpret_cast<const bf16x8*>(a.w + (size_t)(n0 + r + 32 * i) * K + kt * 64 + cc * 8);
    }
    PA_STAMP1(1);
    // ---- per-channel constants of the input transform (or the pending BatchNorm finalize, bn_fin.h), while the loads travel
    if (LDMODE != PA_LD_PLAIN) {
        bool fin_done = false;
        if constexpr (FIN) {
            if (a.fin.rows > 0) { pa_bn_fin_prologue<256, 512>(a.fin, K, kst, kst + 3 * 512, blockIdx.x == 0 && blockIdx.y == 0); fin_done = true; }
        }
        if (!fin_done) {
            for (int c = tid; c < K; c += 256) {
                kst[c] = a.in.k0[c]; kst[512 + c] = a.in.k1[c];
                if (LDMODE == PA_LD_LIN2) kst[1024 + c] = a.in.k2[c];
            }
            __syncthreads();
        }
    }
    PA_STAMP1(2);
    // ---- transform + stage the whole K range
#pragma unroll
    for (int kt = 0; kt < NK; ++kt) {
        const int c = kt * 64 + cc * 8;
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            const int row = r + 32 * i;
            bf16x8 o;
            if (LDMODE == PA_LD_PLAIN) {
                o = xa[kt][i];
            } else if (LDMODE == PA_LD_BNRELU) {
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (bf16)fmaxf(fmaf(kst[c + j], (float)xa[kt][i][j], kst[512 + c + j]), 0.f);
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (bf16)fmaf(kst[c + j], (float)xa[kt][i][j], fmaf(kst[512 + c + j], (float)xq[kt][i][j], kst[1024 + c + j]));
            }
            if (!ok[i]) {
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (bf16)0.f;
            }
            *reinterpret_cast<bf16x8*>(As + kt * (BM * 64) + row * 64 + ((cc ^ (row & 7)) << 3)) = o;
        }
#pragma unroll
        for (int i = 0; i < BI; ++i) {
            const int lrow = pa_lds_row_of_weight_row<BN, NI>(r + 32 * i);
            *reinterpret_cast<bf16x8*>(Bs + kt * (BN * 64) + lrow * 64 + ((cc ^ (lrow & 7)) << 3)) = xb[kt][i];
        }
    }
    PA_STAMP1(3);
    __syncthreads();
    PA_STAMP1(4);

    f32x4 acc[NI][MI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int frow = lane & 15, fchk = lane >> 4;
#pragma unroll
    for (int kt = 0; kt < NK; ++kt)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 fa[MI], fw[NI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int row = wm * (BM / 2) + mi * 16 + frow;
                fa[mi] = *reinterpret_cast<const bf16x8*>(As + kt * (BM * 64) + row * 64 + (((fchk + 4 * kk) ^ (row & 7)) << 3));
            }
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const int row = wn * (BN / 2) + ni * 16 + frow;
                fw[ni] = *reinterpret_cast<const bf16x8*>(Bs + kt * (BN * 64) + row * 64 + (((fchk + 4 * kk) ^ (row & 7)) << 3));
            }
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
                    acc[ni][mi] = PA_MFMA_16x16x32(fw[ni], fa[mi], acc[ni][mi]);
        }
    PA_STAMP1(5);
    __syncthreads();            // every wave is done with the tiles before the epilogue reuses the LDS
    pa_conv_epilogue_auto<BN, NI, MI>(a, acc, n0, wm, wn,
                                     [&](int wr, int mi, int p) { const int m = m0 + wr * (BM / 2) + mi * 16 + p; return m < M ? m : -1; },
                                     reinterpret_cast<float*>(lds), (int)blockIdx.x);
    PA_STAMP1(6);
}

template <int NK>
static void launch_oneshot(const PaConvArgs& a0, dim3 grid, hipStream_t st) {
    PaConvArgs a = a0;
    static int dbg = -1;
    if (dbg < 0) { const char* e = pa_getenv("PA_CONV1_DBG"); dbg = e ? atoi(e) : 0; }      // tuning builds: cycle stamps
    a.dbg = dbg;
    if (a.fin.rows > 0) {
        if (a.in.mode == PA_LD_BNRELU) hipLaunchKernelGGL((conv1x1_oneshot_kernel<NK, PA_LD_BNRELU, true>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((conv1x1_oneshot_kernel<NK, PA_LD_LIN2, true>), grid, dim3(256), 0, st, a);
        return;
    }
    switch (a.in.mode) {
        case PA_LD_PLAIN: hipLaunchKernelGGL((conv1x1_oneshot_kernel<NK, PA_LD_PLAIN, false>), grid, dim3(256), 0, st, a); break;
        case PA_LD_BNRELU: hipLaunchKernelGGL((conv1x1_oneshot_kernel<NK, PA_LD_BNRELU, false>), grid, dim3(256), 0, st, a); break;
        default: hipLaunchKernelGGL((conv1x1_oneshot_kernel<NK, PA_LD_LIN2, false>), grid, dim3(256), 0, st, a); break;
    }
}

template <int BM, int BN, int TAPS>
static void launch_ld(const PaConvArgs& a, dim3 grid, hipStream_t st) {
    if constexpr (BM == 64 && BN == 64) {
        if (a.fin.rows > 0) {          // (pa_conv_takes_fin admits BNRELU / LIN2 inputs only)
            if (a.in.mode == PA_LD_BNRELU) hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, PA_LD_BNRELU, TAPS, false, true>), grid, dim3(256), 0, st, a);
            else hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, PA_LD_LIN2, TAPS, false, true>), grid, dim3(256), 0, st, a);
            return;
        }
    }
    switch (a.in.mode) {
        case PA_LD_PLAIN: hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, PA_LD_PLAIN, TAPS>), grid, dim3(256), 0, st, a); break;
        case PA_LD_BNRELU: hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, PA_LD_BNRELU, TAPS>), grid, dim3(256), 0, st, a); break;
        default: hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, PA_LD_LIN2, TAPS>), grid, dim3(256), 0, st, a); break;
    }
}

template <int BM, int BN>
static void launch_taps(const PaConvArgs& a, dim3 grid, hipStream_t st) {
    if (a.taps == 1) launch_ld<BM, BN, 1>(a, grid, st);
    else launch_ld<BM, BN, 9>(a, grid, st);
}

int pa_launch_stem_conv(const PaConvArgs& a, hipStream_t st, int* stat_rows) {
    // a.in.p = img4 [B][2H][2W][4]; a.w = [64][256]; a.Cin must be 256 (virtual), a.Cout 64
    if (a.Cin != 256 || a.Cout != 64 || a.in.mode != PA_LD_PLAIN) { pa_set_error_msg("pa_launch_stem_conv: bad arguments"); return 1; }
    const int M = a.B * a.H * a.W;
    const int bm = (M >= 128 * 256) ? 128 : 64;
    if (stat_rows) *stat_rows = (M + bm - 1) / bm;
    if (a.ep.rows_out) *a.ep.rows_out = (M + bm - 1) / bm;
    if (bm == 128)
        hipLaunchKernelGGL((conv_igemm_kernel<128, 64, PA_LD_PLAIN, 1, true>), dim3((M + 127) / 128, 1), dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL((conv_igemm_kernel<64, 64, PA_LD_PLAIN, 1, true>), dim3((M + 63) / 64, 1), dim3(256), 0, st, a);
    return (int)hipGetLastError();
}

// mirrors the dispatch below: the row-tile 1x1 kernel carries no finalize prologue
bool pa_conv_takes_fin(const PaConvArgs& a) {
    if (a.in.mode != PA_LD_BNRELU && a.in.mode != PA_LD_LIN2) return false;
    if (a.Cin > 256) return false;
    const bool old3 = pa_getenv("PA_CONV3_OLD") != nullptr, old1 = pa_getenv("PA_CONV1_OLD") != nullptr;
    if (!old3 && pa_conv3x3_tile_supported(a)) return pa_conv3x3_tile_takes_fin(a);
    if (!old1 && pa_conv1x1_tile_supported(a)) return false;
    // generic kernel: only its 64 x 64-tile instance carries the prologue (the dispatch below: small M, or 64-channel outputs)
    const int M = a.B * a.H * a.W;
    const bool bigM = M >= 128 * 256, bigN = a.Cout % 128 == 0;
    return !bigM && !(bigN && M >= 64 * 256);
}

int pa_launch_conv(const PaConvArgs& a0, hipStream_t st, int* stat_rows) {
    static int bwd_direct = -1;
    if (bwd_direct < 0) bwd_direct = pa_getenv("PA_EPI_BWD_DIRECT") ? 2 : 0;      // A/B: direct BatchNorm-backward epilogue
    PaConvArgs a = a0;
    a.xcd = bwd_direct;
    if ((a.taps != 1 && a.taps != 9) || a.Cin % 64 != 0 || a.Cout % 64 != 0 || a.Cin > 512) {
        pa_set_error_msg("pa_launch_conv: channel counts must be multiples of 64 and taps 1 or 9");
        return 1;
    }
    if ((size_t)a.B * a.H * a.W * (size_t)(a.Cin > a.Cout ? a.Cin : a.Cout) >= ((size_t)1 << 31)) {      // the epilogues index with 32 bits
        pa_set_error_msg("pa_launch_conv: tensors of 2^31 elements or more are not supported (split the batch)");
        return 1;
    }
    static int old3 = -1;
    if (old3 < 0) old3 = pa_getenv("PA_CONV3_OLD") ? 1 : 0;          // experiments: force the generic kernel
    if (!old3 && pa_conv3x3_tile_supported(a)) return pa_launch_conv3x3_tile(a, st, stat_rows);
    static int old1 = -1;
    if (old1 < 0) old1 = pa_getenv("PA_CONV1_OLD") ? 1 : 0;
    if (!old1 && pa_conv1x1_tile_supported(a)) {
        if (a.fin.rows > 0) { pa_set_error_msg("pa_launch_conv: a pending finalize was handed to the row-tile 1x1 kernel (pa_conv_takes_fin)"); return 1; }
        return pa_launch_conv1x1_tile(a, st, stat_rows);
    }
    if (a.fin.rows > 0 && (a.fin.rows > PA_FIN_SMALL_ROWS || a.Cin > 256 || (a.Cin & 1) || !pa_conv_takes_fin(a))) { pa_set_error_msg("pa_launch_conv: finalize prologue needs <= 128 partial rows, <= 256 channels and a launch pa_conv_takes_fin() admits"); return 1; }
    const int M = a.B * a.H * a.W;
    // small problems get the 64-row tile so that the grid still covers the 256 CUs
    const bool bigM = M >= 128 * 256;
    const bool bigN = (a.Cout % 128 == 0);
    if (stat_rows) *stat_rows = bigM ? (M + 127) / 128 : (M + 63) / 64;
    if (a.ep.rows_out) *a.ep.rows_out = bigM ? (M + 127) / 128 : (M + 63) / 64;
    if (bigM && bigN) launch_taps<128, 128>(a, dim3((M + 127) / 128, a.Cout / 128), st);
    else if (bigM) launch_taps<128, 64>(a, dim3((M + 127) / 128, a.Cout / 64), st);
    else if (bigN && M >= 64 * 256) launch_taps<64, 128>(a, dim3((M + 63) / 64, a.Cout / 128), st);
    else {
        // 64 x 64 tiles: the 1x1 layers with 128 / 256 input channels take the one-shot kernel (PA_IGEMM_ONESHOT=0: the K-stepping one)
        static int oneshot = -1;
        if (oneshot < 0) { const char* e = pa_getenv("PA_IGEMM_ONESHOT"); oneshot = e ? atoi(e) : 1; }
        const dim3 grid((M + 63) / 64, a.Cout / 64);
        if (oneshot && a.taps == 1 && a.Cin == 256) launch_oneshot<4>(a, grid, st);
        else if (oneshot && a.taps == 1 && a.Cin == 128) launch_oneshot<2>(a, grid, st);
        else launch_taps<64, 64>(a, grid, st);
    }
    return (int)hipGetLastError();
}
