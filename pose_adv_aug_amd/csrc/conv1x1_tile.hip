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

#define PA_CONV1_K32_DEFAULT 1
#ifndef PA_CONV1_UP_OCC
#define PA_CONV1_UP_OCC(up) 3          // workgroups per CU of the 3-workgroup instances (the UP instance included)
#endif
#define PA_CONV1_C64_BM64_DEFAULT 1
#define PA_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define PA_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

// KS: channels per weight slice of the ring.  64 (two 32-wide MFMA steps per slice) everywhere but the 256-channel, 64-row instance of
// round 5 (KS = 32): its LDS is 32 KB of activations + 2 x 8 KB of ring (+ 4 KB table) = 52 KB and its registers <= 168, so that THREE
// workgroups share a CU like the 128-channel instance's -- these kernels are chains of memory round trips (stage, slices, epilogue) and
// what a CU moves is proportional to the chains it interleaves (128-channel instance: 5.4 TB/s hot, the two-workgroup 256-channel one 4.0)
// UP (round 6; CIN = 128, BM = 64, BN = 128, LIN2): the workgroup's 64 rows are 2 IMAGE ROWS x 32 columns instead of 64 consecutive pixels, and
// the epilogue is pa_conv_epilogue_lds_up: d(merged) plus the low-resolution half of the upsample-add backward (a.out2 / a.ep2)
template <int CIN, int BM, int BN, int LDMODE, int KS = 64, bool UP = false>
__global__ __launch_bounds__(256, (((CIN == 128 || KS == 32) && BM == 64) || (CIN == 64 && (BN == 64 || BM == 64))) ? PA_CONV1_UP_OCC(UP) : 2) void conv1x1_tile_kernel(PaConvArgs a, int nb_per_wg) {
    constexpr int CPP = CIN / 8;                     // 16-byte chunks per pixel row
    constexpr int NI = BN / 32, MI = BM / 32;
    constexpr int KT = CIN / KS;
    constexpr int WRI = 512 / KS;                    // weight rows per LDS-DMA wave instruction (1 KB)
    constexpr int NIW = (BN / 4) / WRI;              // ... and instructions per wave and slice
    constexpr int PSTEP = 256 / CPP;                 // rows staged per pass
    constexpr int NPASS = BM / PSTEP;
    // ONE shared object: [A tile][2 weight slices]; the epilogue borrows the ring half that was read last
    // 128 input channels + BatchNorm-backward: 4 KB more for the epilogue's constant table (3 workgroups x 52 KB still fit a CU)
    // (the table form is the only BatchNorm-backward epilogue these instances carry; 64 input channels -- the 128 x 128 maps of residual1:
    // 64 output channels in 128-row tiles or 128 in 64-row tiles, three workgroups per CU either way)
    constexpr bool CTAB = (BM == 64 && ((CIN == 128 && LDMODE == PA_LD_LIN2) || KS == 32)) || (CIN == 64 && (BN == 64 || BM == 64));
    __shared__ __attribute__((aligned(16))) bf16 lds[BM * CIN + 2 * BN * KS + (CTAB ? 2 * BN * 8 : 0)];
    bf16* As = lds;
    bf16* wbuf = lds + BM * CIN;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;
    const int M = a.B * a.H * a.W;
    const int m0 = blockIdx.x * BM;
    const int nb0 = blockIdx.y * nb_per_wg;
    const int nit = nb_per_wg * KT;
    PA_SET_MAIN_PRIO();
    PA_STAMPT(0);
#ifdef PA_TUNING
    // a.dbg >> 16 (tuning builds): stagger -- workgroups of the 2nd / 3rd slot of a CU start 1x / 2x (dbg >> 16) x 1024 cycles late
    if ((a.dbg >> 16) > 0 && gridDim.x > 256) {
        const int slot = ((int)blockIdx.x / 256) % 3;
        for (int i = 0; i < slot * (a.dbg >> 16); ++i) __builtin_amdgcn_s_sleep(16);
    }
#endif
    // tile row -> flattened pixel: consecutive pixels, or (UP) row r of the tile = image row 2 * rp + r / 32, column cx * 32 + r % 32
    // (B * H rows in one column: H is even, a row pair never straddles two images)
    int up_rp = 0, up_cx = 0;
    if constexpr (UP) { const int tpr = a.W / (BM / 2); up_rp = (int)blockIdx.x / tpr; up_cx = (int)blockIdx.x - up_rp * tpr; }
    auto pixel_of_row = [&](int row) -> int {
        if constexpr (UP) return (2 * up_rp + row / (BM / 2)) * a.W + up_cx * (BM / 2) + row % (BM / 2);
        else return m0 + row;
    };

    // ---- weight slices: iteration it -> n-block nb0 + it / KT, k-slice it % KT
    // slot swizzle of a slice row: 128-byte rows (KS = 64): 16-byte slot ^ (row & 7); 64-byte rows (KS = 32): slot ^ 3 * bit 3 of the row --
    // the four lane groups of a ds_read_b128 (16 rows x one slot per quarter wave) then hit 16 distinct 16-byte bank groups
    int wrow[NIW], wcol[NIW];
#pragma unroll
    for (int i = 0; i < NIW; ++i) {
        const int lr = wave * (BN / 4) + i * WRI + (KS == 64 ? (lane >> 3) : (lane >> 2));
        wrow[i] = pa_weight_row_of_lds_row<BN, NI>(lr);
        wcol[i] = KS == 64 ? (((lane & 7) ^ (lr & 7)) << 3) : (((lane & 3) ^ (((lr >> 3) & 1) * 3)) << 3);
    }
    auto issue_w = [&](int it, int buf) {
        const int nb = nb0 + it / KT, kh = it % KT;
#pragma unroll
        for (int i = 0; i < NIW; ++i)
            __builtin_amdgcn_global_load_lds(PA_GLOBAL_PTR(a.w + (size_t)(nb * BN + wrow[i]) * CIN + kh * KS + wcol[i]),
                                             PA_LDS_PTR(wbuf + buf * (BN * KS) + (wave * (BN / 4) + i * WRI) * KS), 16, 0, 0);
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
                const int m = pixel_of_row((p0 + u) * PSTEP + tid / CPP);
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
                if (pixel_of_row(row) >= M) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] = (bf16)0.f;
                }
                const int sw = CPP >= 16 ? (row & 15) : ((row >> 1) & 7);
                *reinterpret_cast<bf16x8*>(As + row * CIN + ((chunk ^ sw) << 3)) = o;
                if (LDMODE == PA_LD_LIN2 && a.dz_out && blockIdx.y == 0 && pixel_of_row(row) < M)
                    *reinterpret_cast<bf16x8*>(a.dz_out + (size_t)pixel_of_row(row) * CIN + c) = o;
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
        if (KS == 32 && nbi > 0) {                    // (the epilogue of the previous block used the WHOLE ring: this block's first slice starts here)
            issue_w(nbi * KT, (nbi * KT) & 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
#pragma unroll
        for (int kh = 0; kh < KT; ++kh) {
            const int it = nbi * KT + kh;
            if (it + 1 < nit && (KS == 64 || kh + 1 < KT)) issue_w(it + 1, (it + 1) & 1);
            const bf16* Bs = wbuf + (it & 1) * (BN * KS);
#pragma unroll
            for (int kk = 0; kk < KS / 32; ++kk) {
                bf16x8 fa[MI], fw[NI];
                const int chunk = kh * (KS / 8) + kk * 4 + fchk;
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    const int row = wm * (BM / 2) + mi * 16 + frow;
                    const int sw = CPP >= 16 ? (row & 15) : ((row >> 1) & 7);
                    fa[mi] = *reinterpret_cast<const bf16x8*>(As + row * CIN + ((chunk ^ sw) << 3));
                }
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    const int row = wn * (BN / 2) + ni * 16 + frow;
                    fw[ni] = KS == 64 ? *reinterpret_cast<const bf16x8*>(Bs + row * 64 + (((fchk + 4 * kk) ^ (row & 7)) << 3))
                                      : *reinterpret_cast<const bf16x8*>(Bs + row * 32 + ((fchk ^ (((row >> 3) & 1) * 3)) << 3));
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
        // (KS = 32: a ring half is 8 KB, an epilogue pass of 32 pixel rows x 128 channels needs 16 -- the whole ring, so the next block's
        // first slice is not requested across the epilogue)
        float* T = reinterpret_cast<float*>(KS == 32 ? wbuf : wbuf + ((nbi * KT + KT - 1) & 1) * (BN * KS));
        // BatchNorm-backward epilogue through LDS for 256 input channels (cold 70 vs 76 us) and for 64 (round 5: the 128 x 128 maps of
        // residual1, where the direct form's 64-byte runs are HALF a pixel row); the 128-channel kernels (3 workgroups per CU,
        // 168 registers) spill with the plain LDS form (98 vs 70 us) and take the table form (CTAB)
        if constexpr (UP) {
            pa_conv_epilogue_lds_up<BN, NI, MI>(a, acc, (nb0 + nbi) * BN, wm, wn, pixel_of_row,
                                                [&](int x) { return up_rp * (a.W / 2) + up_cx * (BM / 4) + (x >> 1); },
                                                T, (int)blockIdx.x, reinterpret_cast<float4*>(lds + BM * CIN + 2 * BN * KS));
        } else
        pa_conv_epilogue_auto<BN, NI, MI, (CIN == 256 || CIN == 64 || CTAB), CTAB, 256, (BN < 128)>(a, acc, (nb0 + nbi) * BN, wm, wn,
                                         [&](int wr, int mi, int p) { const int m = m0 + wr * (BM / 2) + mi * 16 + p; return m < M ? m : -1; },
                                         T, (int)blockIdx.x, CTAB ? reinterpret_cast<float4*>(lds + BM * CIN + 2 * BN * KS) : nullptr);
        if (nbi == 0) PA_STAMPT(5);                   // its epilogue
        __syncthreads();            // T is handed back to the weight ring
    }
    PA_STAMPT(6);
}

template <int CIN, int BM, int BN, int KS = 64>
static void launch_row_ld(const PaConvArgs& a0, dim3 grid, int nbw, hipStream_t st) {
    PaConvArgs a = a0;
    static int dbg = -1;
    if (dbg < 0) { const char* e = pa_getenv("PA_CONV1T_DBG"); dbg = e ? atoi(e) : 0; }      // tuning builds: cycle stamps
    a.dbg = dbg;
    switch (a.in.mode) {
        case PA_LD_PLAIN: hipLaunchKernelGGL((conv1x1_tile_kernel<CIN, BM, BN, PA_LD_PLAIN, KS>), grid, dim3(256), 0, st, a, nbw); break;
        case PA_LD_BNRELU: hipLaunchKernelGGL((conv1x1_tile_kernel<CIN, BM, BN, PA_LD_BNRELU, KS>), grid, dim3(256), 0, st, a, nbw); break;
        default: hipLaunchKernelGGL((conv1x1_tile_kernel<CIN, BM, BN, PA_LD_LIN2, KS>), grid, dim3(256), 0, st, a, nbw); break;
    }
}

// 64-row tiles also for 128 input channels: at B = 24, 64x64 maps the 768 tiles of 128 rows fill 256 CUs x 2
// workgroups 1.5 times (the last round runs half empty); 1536 tiles of 64 rows at 3 workgroups/CU are 2 full rounds
static int row_bm(int Cin) {
    static int big = -1;
    if (big < 0) big = pa_getenv("PA_CONV1_BM128") ? 1 : 0;
    return (Cin == 256 || (Cin == 128 && !big)) ? 64 : 128;
}

// the UP instance: 128 -> 256 (any multiple of 128) channels, BatchNorm backward on load, one plain addend, plain output, maps whose rows are
// multiples of 32 pixels with an even row count, enough tiles for the row-tile kernel
bool pa_conv1x1_tile_up_supported(const PaConvArgs& a) {
    if (!a.out2 || a.taps != 1 || a.Cin != 128 || a.Cout % 128 != 0 || a.in.mode != PA_LD_LIN2 || a.bias || a.dz_out || a.fin.rows > 0) return false;
    if (a.add1.mode != PA_LD_PLAIN || a.add2.mode != PA_LD_NONE || a.ep.mode != PA_OUT_PLAIN || a.ep2.mode != PA_OUT_BWD) return false;
    if (a.W % 32 != 0 || a.H % 2 != 0) return false;
    const long M = (long)a.B * a.H * a.W;
    if ((size_t)M * (size_t)a.Cout >= ((size_t)1 << 31)) return false;
    return M / 64 >= 192;
}

static int launch_conv1x1_tile_up(const PaConvArgs& a, hipStream_t st, int* stat_rows) {
    const int M = a.B * a.H * a.W, tiles = M / 64;
    if (stat_rows) *stat_rows = tiles;
    if (a.ep2.rows_out) *a.ep2.rows_out = tiles;
    const int nb = a.Cout / 128;
    const int nbw = tiles >= 512 ? nb : 1;
    hipLaunchKernelGGL((conv1x1_tile_kernel<128, 64, 128, PA_LD_LIN2, 64, true>), dim3(tiles, nb / nbw), dim3(256), 0, st, a, nbw);
    return (int)hipGetLastError();
}

bool pa_conv1x1_tile_supported(const PaConvArgs& a) {
    if (a.taps != 1 || (a.Cin != 64 && a.Cin != 128 && a.Cin != 256) || a.Cout % 64 != 0) return false;
    const int M = a.B * a.H * a.W;
    if ((size_t)M * (size_t)(a.Cin > a.Cout ? a.Cin : a.Cout) >= ((size_t)1 << 31)) return false;      // 32-bit element offsets in the epilogue
    // (64 -> 64 channels: every instance carries the table form of the BatchNorm-backward epilogue only)
    if (a.Cin == 64 && a.Cout % 128 != 0 && a.ep.mode == PA_OUT_BWD && !pa_bwd_epilogue_lds_ok(a)) return false;
    return (M + row_bm(a.Cin) - 1) / row_bm(a.Cin) >= 192;          // smaller problems: generic kernel with 64x64 tiles
}

int pa_launch_conv1x1_tile(const PaConvArgs& a, hipStream_t st, int* stat_rows) {
    if (a.out2) {
        if (!pa_conv1x1_tile_up_supported(a)) { pa_set_error_msg("pa_launch_conv1x1_tile: not a launch of the 2 x 32-pixel instance (pa_conv1x1_tile_up_supported)"); return 1; }
        return launch_conv1x1_tile_up(a, st, stat_rows);
    }
    if (!pa_conv1x1_tile_supported(a)) { pa_set_error_msg("pa_launch_conv1x1_tile: unsupported shape"); return 1; }
    const int M = a.B * a.H * a.W;
    int bm = row_bm(a.Cin);
    // 64 input channels, 128 output channels: 64-row tiles at three workgroups per CU (the 128-row instance spills at 168 registers)
    static int c64bm = -1;
    if (c64bm < 0) { const char* e = pa_getenv("PA_CONV1_C64_BM64"); c64bm = e ? atoi(e) : PA_CONV1_C64_BM64_DEFAULT; }
    if (a.Cin == 64 && a.Cout % 128 == 0 && c64bm && !(a.ep.mode == PA_OUT_BWD && !pa_bwd_epilogue_lds_ok(a))) bm = 64;
    // the 64-row, 128-channel BatchNorm-backward instance (3 workgroups per CU) has the LDS epilogue only
    if (a.Cin == 128 && bm == 64 && a.in.mode == PA_LD_LIN2 && a.ep.mode == PA_OUT_BWD && !pa_bwd_epilogue_lds_ok(a)) bm = 128;
    const int tiles = (M + bm - 1) / bm;
    if (stat_rows) *stat_rows = tiles;
    if (a.ep.rows_out) *a.ep.rows_out = tiles;
    const bool bigN = a.Cout % 128 == 0;
    const int nb = a.Cout / (bigN ? 128 : 64);
    const int nbw = tiles >= 512 ? nb : 1;          // enough row tiles: loop over the channel blocks inside (input read once)
    dim3 grid(tiles, nb / nbw);
    static int k32 = -1;
    if (k32 < 0) { const char* e = pa_getenv("PA_CONV1_K32"); k32 = e ? atoi(e) : PA_CONV1_K32_DEFAULT; }
    // 256 channels: the three-workgroup instance (32-channel weight slices) unless the launch needs the direct BatchNorm-backward epilogue
    const bool three = k32 && bigN && !(a.ep.mode == PA_OUT_BWD && !pa_bwd_epilogue_lds_ok(a));
    if (a.Cin == 256) { if (bigN && three) launch_row_ld<256, 64, 128, 32>(a, grid, nbw, st); else if (bigN) launch_row_ld<256, 64, 128>(a, grid, nbw, st); else launch_row_ld<256, 64, 64>(a, grid, nbw, st); }
    else if (a.Cin == 128 && bm == 64) { if (bigN) launch_row_ld<128, 64, 128>(a, grid, nbw, st); else launch_row_ld<128, 64, 64>(a, grid, nbw, st); }
    else if (a.Cin == 128) { if (bigN) launch_row_ld<128, 128, 128>(a, grid, nbw, st); else launch_row_ld<128, 128, 64>(a, grid, nbw, st); }
    else if (bm == 64) launch_row_ld<64, 64, 128>(a, grid, nbw, st);
    else { if (bigN) launch_row_ld<64, 128, 128>(a, grid, nbw, st); else launch_row_ld<64, 128, 64>(a, grid, nbw, st); }
    return (int)hipGetLastError();
}
