// 1x1 convolution (forward and data gradient) as a ROW-TILE GEMM on bf16 MFMA, gfx950.
//
// These layers (reference models/asn_stacked_hg.py:17,23,25 conv1/conv3/adapter, :241-248 linear /
// forth_conv) are HBM-bound (85 FLOP/B at 256->128 channels): the kernel is organised around reading the
// activation rows ONCE and keeping many loads in flight, not around the MFMA.
//   * a workgroup owns BM consecutive NHWC pixels and stages the whole [BM][CIN] activation tile once
//     (all of a thread's 16-byte loads are issued before the first transform; the pending BatchNorm+ReLU
//     / BatchNorm backward is applied in this pass), then loops over the output-channel blocks itself,
//     so the input is not re-read per 128 output channels;
//   * the weight slices [BN][64] are streamed with global_load_lds, double buffered (conv3x3_tile.hip);
//   * two or three workgroups per CU (48-64 KB LDS): one stages while the others compute; 64-row tiles so that a
//     24 x 64 x 64 map is 1536 workgroups = full rounds (row_bm below).
// LDS images: activation tile [BM][CIN] and weight slice [BN][64] bf16, 16-byte slots XOR-swizzled by the
// row so that the ds_read_b128 fragment reads are bank-conflict free.  Epilogue: conv_epilogue.h (forward modes through
// an fp32 LDS tile with fully coalesced rows, BatchNorm-backward mode direct).
#include "common.h"
#include "kernels.h"
#include "conv_epilogue.h"
#include <stdlib.h>

// cycle stamps of ONE workgroup (tuning builds, tools/conv1t_clocks.py): PA_CONV1T_DBG = 64 + 256 * <blockIdx.x to stamp>
PA_STAMP_DECL(pa_conv1t_clk, pa_debug_conv1t_clocks)
#ifdef PA_TUNING
#define PA_STAMPT(i) do { if ((a.dbg & 64) && blockIdx.x == (unsigned)(a.dbg >> 8) && blockIdx.y == 0 && threadIdx.x == 0) { \
        pa_conv1t_clk[2 * (i)] = __builtin_amdgcn_s_memtime(); pa_conv1t_clk[2 * (i) + 1] = wall_clock64(); } } while (0)
#else
#define PA_STAMPT(i) do { } while (0)
#endif

#define PA_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define PA_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

template <int CIN, int BM, int BN, int LDMODE>
__global__ __launch_bounds__(256, (CIN == 128 && BM == 64) ? 3 : 2) void conv1x1_tile_kernel(PaConvArgs a, int nb_per_wg) {
    constexpr int CPP = CIN / 8;                     // 16-byte chunks per pixel row
    constexpr int NI = BN / 32, MI = BM / 32;
    constexpr int KT = CIN / 64;
    constexpr int PSTEP = 256 / CPP;                 // rows staged per pass
    constexpr int NPASS = BM / PSTEP;
    // ONE shared object: [A tile][2 weight slices]; the epilogue borrows the ring half that was read last
    // 128 input channels + BatchNorm-backward: 4 KB more for the epilogue's constant table (3 workgroups x 52 KB still fit a CU)
    constexpr bool CTAB = CIN == 128 && BM == 64 && LDMODE == PA_LD_LIN2;
    __shared__ __attribute__((aligned(16))) bf16 lds[BM * CIN + 2 * BN * 64 + (CTAB ? 2 * BN * 8 : 0)];
    bf16* As = lds;
    bf16* wbuf = lds + BM * CIN;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;
    const int M = a.B * a.H * a.W;
    const int m0 = blockIdx.x * BM;
    const int nb0 = blockIdx.y * nb_per_wg;
    const int nit = nb_per_wg * KT;
    PA_STAMPT(0);

    // ---- weight slices: iteration it -> n-block nb0 + it / KT, k-slice it % KT
    int wrow[NI], wcol[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int lr = wave * (BN / 4) + i * 8 + (lane >> 3);
        wrow[i] = pa_weight_row_of_lds_row<BN, NI>(lr);
        wcol[i] = ((lane & 7) ^ (lr & 7)) << 3;
    }
    auto issue_w = [&](int it, int buf) {
        const int nb = nb0 + it / KT, kh = it % KT;
#pragma unroll
        for (int i = 0; i < NI; ++i)
            __builtin_amdgcn_global_load_lds(PA_GLOBAL_PTR(a.w + (size_t)(nb * BN + wrow[i]) * CIN + kh * 64 + wcol[i]),
                                             PA_LDS_PTR(wbuf + buf * (BN * 64) + (wave * (BN / 4) + i * 8) * 64), 16, 0, 0);
    };
    issue_w(0, 0);

    // ---- activation tile: one pass, every load of a thread in flight before the first transform
    {
        const int chunk = tid % CPP, c = chunk * 8;
        float k0[8], k1[8], k2[8];
        if (LDMODE != PA_LD_PLAIN) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                k0[j] = a.in.k0[c + j]; k1[j] = a.in.k1[c + j];
                if (LDMODE == PA_LD_LIN2) k2[j] = a.in.k2[c + j];
            }
        }
        constexpr int UN = (LDMODE == PA_LD_LIN2 && NPASS > 4) ? NPASS / 2 : NPASS;
#pragma unroll
        for (int p0 = 0; p0 < NPASS; p0 += UN) {
            bf16x8 ra[UN], rq[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int m = m0 + (p0 + u) * PSTEP + tid / CPP;
                const size_t idx = m < M ? (size_t)m * CIN + c : 0;        // clamped, unconditional loads
                ra[u] = *reinterpret_cast<const bf16x8*>(a.in.p + idx);
                if (LDMODE == PA_LD_LIN2) rq[u] = *reinterpret_cast<const bf16x8*>(a.in.q + idx);
            }
            if (p0 == 0) PA_STAMPT(1);                // the first batch's loads are issued
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int row = (p0 + u) * PSTEP + tid / CPP;
                bf16x8 o;
                if (LDMODE == PA_LD_PLAIN) {
                    o = ra[u];
                } else if (LDMODE == PA_LD_BNRELU) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] = (bf16)fmaxf(fmaf(k0[j], (float)ra[u][j], k1[j]), 0.f);
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        o[j] = (bf16)fmaf(k0[j], (float)ra[u][j], fmaf(k1[j], (float)rq[u][j], k2[j]));
                }
                if (m0 + row >= M) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] = (bf16)0.f;
                }
                const int sw = CPP >= 16 ? (row & 15) : ((row >> 1) & 7);
                *reinterpret_cast<bf16x8*>(As + row * CIN + ((chunk ^ sw) << 3)) = o;
                if (LDMODE == PA_LD_LIN2 && a.dz_out && blockIdx.y == 0 && m0 + row < M)
                    *reinterpret_cast<bf16x8*>(a.dz_out + (size_t)(m0 + row) * CIN + c) = o;
            }
        }
    }

    PA_STAMPT(2);                                     // transformed and written to LDS (this thread)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    PA_STAMPT(3);

    const int frow = lane & 15, fchk = lane >> 4;
    for (int nbi = 0; nbi < nb_per_wg; ++nbi) {
        f32x4 acc[NI][MI];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kh = 0; kh < KT; ++kh) {
            const int it = nbi * KT + kh;
            if (it + 1 < nit) issue_w(it + 1, (it + 1) & 1);
            const bf16* Bs = wbuf + (it & 1) * (BN * 64);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                bf16x8 fa[MI], fw[NI];
                const int chunk = kh * 8 + kk * 4 + fchk;
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    const int row = wm * (BM / 2) + mi * 16 + frow;
                    const int sw = CPP >= 16 ? (row & 15) : ((row >> 1) & 7);
                    fa[mi] = *reinterpret_cast<const bf16x8*>(As + row * CIN + ((chunk ^ sw) << 3));
                }
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    const int row = wn * (BN / 2) + ni * 16 + frow;
                    fw[ni] = *reinterpret_cast<const bf16x8*>(Bs + row * 64 + (((fchk + 4 * kk) ^ (row & 7)) << 3));
                }
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
                        acc[ni][mi] = PA_MFMA_16x16x32(fw[ni], fa[mi], acc[ni][mi]);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        if (nbi == 0) PA_STAMPT(4);                   // K loop of the first channel block
        // the ring half that the last K step read is free now (the other one holds the next block's first slice)
        float* T = reinterpret_cast<float*>(wbuf + ((nbi * KT + KT - 1) & 1) * (BN * 64));
        // BatchNorm-backward epilogue through LDS only for 256 input channels (cold 70 vs 76 us); the 128-channel kernels
        // (3 workgroups per CU, 168 registers) spill with it: 98 vs 70 us
        pa_conv_epilogue_auto<BN, NI, MI, (CIN == 256 || CTAB), CTAB, 256, (BN < 128)>(a, acc, (nb0 + nbi) * BN, wm, wn,
                                         [&](int wr, int mi, int p) { const int m = m0 + wr * (BM / 2) + mi * 16 + p; return m < M ? m : -1; },
                                         T, (int)blockIdx.x, CTAB ? reinterpret_cast<float4*>(lds + BM * CIN + 2 * BN * 64) : nullptr);
        if (nbi == 0) PA_STAMPT(5);                   // its epilogue
        __syncthreads();            // T is handed back to the weight ring
    }
    PA_STAMPT(6);
}

template <int CIN, int BM, int BN>
static void launch_row_ld(const PaConvArgs& a0, dim3 grid, int nbw, hipStream_t st) {
    PaConvArgs a = a0;
    static int dbg = -1;
    if (dbg < 0) { const char* e = pa_getenv("PA_CONV1T_DBG"); dbg = e ? atoi(e) : 0; }      // tuning builds: cycle stamps
    a.dbg = dbg;
    switch (a.in.mode) {
        case PA_LD_PLAIN: hipLaunchKernelGGL((conv1x1_tile_kernel<CIN, BM, BN, PA_LD_PLAIN>), grid, dim3(256), 0, st, a, nbw); break;
        case PA_LD_BNRELU: hipLaunchKernelGGL((conv1x1_tile_kernel<CIN, BM, BN, PA_LD_BNRELU>), grid, dim3(256), 0, st, a, nbw); break;
        default: hipLaunchKernelGGL((conv1x1_tile_kernel<CIN, BM, BN, PA_LD_LIN2>), grid, dim3(256), 0, st, a, nbw); break;
    }
}

// 64-row tiles also for 128 input channels: at B = 24, 64x64 maps the 768 tiles of 128 rows fill 256 CUs x 2
// workgroups 1.5 times (the last round runs half empty); 1536 tiles of 64 rows at 3 workgroups/CU are 2 full rounds
static int row_bm(int Cin) {
    static int big = -1;
    if (big < 0) big = pa_getenv("PA_CONV1_BM128") ? 1 : 0;
    return (Cin == 256 || (Cin == 128 && !big)) ? 64 : 128;
}

bool pa_conv1x1_tile_supported(const PaConvArgs& a) {
    if (a.taps != 1 || (a.Cin != 64 && a.Cin != 128 && a.Cin != 256) || a.Cout % 64 != 0) return false;
    const int M = a.B * a.H * a.W;
    if ((size_t)M * (size_t)(a.Cin > a.Cout ? a.Cin : a.Cout) >= ((size_t)1 << 31)) return false;      // 32-bit element offsets in the epilogue
    return (M + row_bm(a.Cin) - 1) / row_bm(a.Cin) >= 192;          // smaller problems: generic kernel with 64x64 tiles
}

int pa_launch_conv1x1_tile(const PaConvArgs& a, hipStream_t st, int* stat_rows) {
    if (!pa_conv1x1_tile_supported(a)) { pa_set_error_msg("pa_launch_conv1x1_tile: unsupported shape"); return 1; }
    const int M = a.B * a.H * a.W;
    int bm = row_bm(a.Cin);
    // the 64-row, 128-channel BatchNorm-backward instance (3 workgroups per CU) has the LDS epilogue only
    if (a.Cin == 128 && bm == 64 && a.in.mode == PA_LD_LIN2 && a.ep.mode == PA_OUT_BWD && !pa_bwd_epilogue_lds_ok(a)) bm = 128;
    const int tiles = (M + bm - 1) / bm;
    if (stat_rows) *stat_rows = tiles;
    if (a.ep.rows_out) *a.ep.rows_out = tiles;
    const bool bigN = a.Cout % 128 == 0;
    const int nb = a.Cout / (bigN ? 128 : 64);
    const int nbw = tiles >= 512 ? nb : 1;          // enough row tiles: loop over the channel blocks inside (input read once)
    dim3 grid(tiles, nb / nbw);
    if (a.Cin == 256) { if (bigN) launch_row_ld<256, 64, 128>(a, grid, nbw, st); else launch_row_ld<256, 64, 64>(a, grid, nbw, st); }
    else if (a.Cin == 128 && bm == 64) { if (bigN) launch_row_ld<128, 64, 128>(a, grid, nbw, st); else launch_row_ld<128, 64, 64>(a, grid, nbw, st); }
    else if (a.Cin == 128) { if (bigN) launch_row_ld<128, 128, 128>(a, grid, nbw, st); else launch_row_ld<128, 128, 64>(a, grid, nbw, st); }
    else { if (bigN) launch_row_ld<64, 128, 128>(a, grid, nbw, st); else launch_row_ld<64, 128, 64>(a, grid, nbw, st); }
    return (int)hipGetLastError();
}
