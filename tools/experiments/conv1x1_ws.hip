// 1x1 convolution (forward and data gradient), WEIGHT-STATIONARY row-tile GEMM on bf16 MFMA, gfx950.
//
// The row-tile kernel of conv1x1_tile.hip streams the [128][64] weight slices of a workgroup through a two-deep LDS ring:
// four (256 input channels) dependent L2 round trips with a barrier each per 64-row tile, and 64 KB of weights through
// L2 for every 32 KB of activations.  The bottleneck layers (reference models/asn_stacked_hg.py:17,25 conv1 256->128,
// conv3 128->256 and their data gradients) have 32 K weights: a WAVE's share of them -- 64 output channels x 256 inputs,
// or 2 x 64 x 128 -- is 128 registers as MFMA A-operand fragments.  Here every wave loads its share ONCE, straight from
// global memory into registers, and the workgroup then walks over row tiles:
//   * K loop of a tile: 2 ds_read_b128 + 8 MFMA per 32 input channels, no barrier, no memory access;
//   * the raw rows of the NEXT tile of the workgroup are requested (registers) before the K loop of the current one and
//     are in flight during its K loop and epilogue (the pending BatchNorm+ReLU / BatchNorm backward is applied when they
//     are written to LDS, as in the row-tile kernel);
//   * the accumulation order per output element is that of the row-tile kernel (32-channel steps in ascending order,
//     same MFMA, same operand roles) and the epilogue is conv_epilogue.h's: the results are BITWISE those of
//     conv1x1_tile_kernel, statistics rows included (one per 64-row tile, row index = tile index).
// Grid: min(tiles, OCC * 256) workgroups of 256 threads; tile t of workgroup g: g, g + grid, ...
#include "common.h"
#include "kernels.h"
#include "conv_epilogue.h"
#include <stdlib.h>

template <int CIN, int COUT, int LDMODE, int OCC>
__global__ __launch_bounds__(256, OCC) void conv1x1_ws_kernel(PaConvArgs a, int tiles) {
    constexpr int BM = 64, BN = 128, NI = 4, MI = 2;
    constexpr int CPP = CIN / 8;                     // 16-byte chunks per pixel row
    constexpr int KS = CIN / 32;                     // MFMA K steps
    constexpr int NBLK = COUT / BN;
    constexpr int PSTEP = 256 / CPP;                 // rows staged per pass
    constexpr int NPASS = BM / PSTEP;
    __shared__ __attribute__((aligned(16))) bf16 As[BM * CIN];
    __shared__ __attribute__((aligned(16))) float T[32 * BN];
    __shared__ float4 ctab[2 * BN];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;
    const int frow = lane & 15, fchk = lane >> 4;
    const int M = a.B * a.H * a.W;

    // ---- raw rows of a tile -> registers (clamped, unconditional loads)
    const int chunk = tid % CPP, c = chunk * 8, rsub = tid / CPP;
    bf16x8 ra[NPASS], rq[NPASS];
    auto request = [&](int t) {
        const int m0 = t * BM;
#pragma unroll
        for (int u = 0; u < NPASS; ++u) {
            const int m = m0 + u * PSTEP + rsub;
            const size_t idx = m < M ? (size_t)m * CIN + c : 0;
            ra[u] = *reinterpret_cast<const bf16x8*>(a.in.p + idx);
            if (LDMODE == PA_LD_LIN2) rq[u] = *reinterpret_cast<const bf16x8*>(a.in.q + idx);
        }
    };
    int t = blockIdx.x;
    if (t < tiles) request(t);

    // ---- this wave's weights: fragment (block nbi, ni, step s) = rows of 16 output channels x 32 input channels
    bf16x8 wr[NBLK][NI][KS];
#pragma unroll
    for (int nbi = 0; nbi < NBLK; ++nbi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int n = nbi * BN + pa_weight_row_of_lds_row<BN, NI>(wn * (BN / 2) + ni * 16 + frow);
#pragma unroll
            for (int s = 0; s < KS; ++s) wr[nbi][ni][s] = *reinterpret_cast<const bf16x8*>(a.w + (size_t)n * CIN + s * 32 + fchk * 8);
        }

    // per-channel constants of the pending transform (the thread's chunk is the same for every row it stages)
    float k0[8], k1[8], k2[8];
    if (LDMODE != PA_LD_PLAIN) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            k0[j] = a.in.k0[c + j]; k1[j] = a.in.k1[c + j];
            if (LDMODE == PA_LD_LIN2) k2[j] = a.in.k2[c + j];
        }
    }

    for (; t < tiles; t += gridDim.x) {
        const int m0 = t * BM;
        // ---- tile t: transform -> LDS
#pragma unroll
        for (int u = 0; u < NPASS; ++u) {
            const int row = u * PSTEP + rsub;
            bf16x8 o;
            if (LDMODE == PA_LD_PLAIN) {
                o = ra[u];
            } else if (LDMODE == PA_LD_BNRELU) {
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (bf16)fmaxf(fmaf(k0[j], (float)ra[u][j], k1[j]), 0.f);
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (bf16)fmaf(k0[j], (float)ra[u][j], fmaf(k1[j], (float)rq[u][j], k2[j]));
            }
            if (m0 + row >= M) {
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (bf16)0.f;
            }
            const int sw = CPP >= 16 ? (row & 15) : ((row >> 1) & 7);
            *reinterpret_cast<bf16x8*>(As + row * CIN + ((chunk ^ sw) << 3)) = o;
            if (LDMODE == PA_LD_LIN2 && a.dz_out && m0 + row < M) *reinterpret_cast<bf16x8*>(a.dz_out + (size_t)(m0 + row) * CIN + c) = o;
        }
        __syncthreads();
        // ---- the next tile's rows: in flight from here to the next transform
        if (t + (int)gridDim.x < tiles) request(t + gridDim.x);

#pragma unroll
        for (int nbi = 0; nbi < NBLK; ++nbi) {
            f32x4 acc[NI][MI];
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                bf16x8 fa[MI];
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    const int row = wm * (BM / 2) + mi * 16 + frow;
                    const int sw = CPP >= 16 ? (row & 15) : ((row >> 1) & 7);
                    fa[mi] = *reinterpret_cast<const bf16x8*>(As + row * CIN + (((s * 4 + fchk) ^ sw) << 3));
                }
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = PA_MFMA_16x16x32(wr[nbi][ni][s], fa[mi], acc[ni][mi]);
            }
            pa_conv_epilogue_auto<BN, NI, MI, true, true>(a, acc, nbi * BN, wm, wn,
                                                         [&](int wr_, int mi, int p) { const int m = m0 + wr_ * (BM / 2) + mi * 16 + p; return m < M ? m : -1; },
                                                         T, t, ctab);
            __syncthreads();            // T is free again; after the last block: every wave is past its reads of the tile
        }
    }
}

template <int CIN, int COUT, int OCC>
static void launch_ws_ld(const PaConvArgs& a, int grid, int tiles, hipStream_t st) {
    switch (a.in.mode) {
        case PA_LD_PLAIN: hipLaunchKernelGGL((conv1x1_ws_kernel<CIN, COUT, PA_LD_PLAIN, OCC>), dim3(grid), dim3(256), 0, st, a, tiles); break;
        case PA_LD_BNRELU: hipLaunchKernelGGL((conv1x1_ws_kernel<CIN, COUT, PA_LD_BNRELU, OCC>), dim3(grid), dim3(256), 0, st, a, tiles); break;
        default: hipLaunchKernelGGL((conv1x1_ws_kernel<CIN, COUT, PA_LD_LIN2, OCC>), dim3(grid), dim3(256), 0, st, a, tiles); break;
    }
}

static int ws_mode() {                               // PA_CONV1_WS: 0 = off, 1 = one workgroup per CU, 2 = two (default)
    static int m = -1;
    if (m < 0) { const char* e = pa_getenv("PA_CONV1_WS"); m = e ? atoi(e) : 2; }
    return m;
}

bool pa_conv1x1_ws_supported(const PaConvArgs& a) {
    if (!ws_mode() || a.taps != 1) return false;
    if (!((a.Cin == 256 && a.Cout == 128) || (a.Cin == 128 && a.Cout == 256))) return false;
    if (a.in.mode != PA_LD_PLAIN && a.in.mode != PA_LD_BNRELU && a.in.mode != PA_LD_LIN2) return false;
    const int M = a.B * a.H * a.W;
    if ((size_t)M * 256 >= ((size_t)1 << 31)) return false;               // 32-bit element offsets in the epilogue
    return (M + 63) / 64 >= 192;
}

int pa_launch_conv1x1_ws(const PaConvArgs& a, hipStream_t st, int* stat_rows) {
    if (!pa_conv1x1_ws_supported(a)) { pa_set_error_msg("pa_launch_conv1x1_ws: unsupported shape"); return 1; }
    const int M = a.B * a.H * a.W;
    const int tiles = (M + 63) / 64;
    if (stat_rows) *stat_rows = tiles;
    if (a.ep.rows_out) *a.ep.rows_out = tiles;
    const int occ = ws_mode() == 1 ? 1 : 2;
    static int cap = -1;
    if (cap < 0) { const char* e = pa_getenv("PA_CONV1_WS_GRID"); cap = e ? atoi(e) : 0; }
    int grid = cap > 0 ? cap : occ * 256;
    if (grid > tiles) grid = tiles;
    if (a.Cin == 256) { if (occ == 1) launch_ws_ld<256, 128, 1>(a, grid, tiles, st); else launch_ws_ld<256, 128, 2>(a, grid, tiles, st); }
    else { if (occ == 1) launch_ws_ld<128, 256, 1>(a, grid, tiles, st); else launch_ws_ld<128, 256, 2>(a, grid, tiles, st); }
    return (int)hipGetLastError();
}
