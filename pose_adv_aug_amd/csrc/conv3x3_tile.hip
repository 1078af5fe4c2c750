// 3x3 'same' convolution (forward and data gradient) as a HALO-TILE implicit GEMM on bf16 MFMA, gfx950.
//
// The generic kernel (conv_igemm.hip) re-loads and re-transforms the activation tile for each of the 9
// taps and spends ~10 VALU instructions per MFMA doing so.  Here a workgroup owns an 8 x 16 block of
// output pixels of one image and
//   * stages the (8+2) x (16+2) input halo ONCE into LDS -- the pending BatchNorm+ReLU (PA_LD_BNRELU,
//     forward: reference models/asn_stacked_hg.py:36-41) or BatchNorm backward (PA_LD_LIN2, dgrad) is
//     applied during this single pass, zero padding is written as zeros, 1.4x read amplification
//     instead of 9x;
//   * streams the weight tile [BN][64] of every (tap, 64-channel slice) with global_load_lds
//     (16 B per lane, no VGPR staging, no ds_write), double buffered: slice t+1 is in flight while slice
//     t feeds the MFMAs; the LDS image is [row][64] with the 16-byte slot XOR-swizzled by (row & 7),
//     realised on the per-lane SOURCE address (the LDS side of global_load_lds is lane-linear);
//   * reads the activation fragment of tap (dy,dx) straight from the halo at pixel offset
//     (dy*18 + dx): no per-tap address or bounds arithmetic besides one add and the swizzle.
// Halo image: [180 pixels][CIN] bf16, 16-byte slot XOR-swizzled by the pixel index so that the 16
// lanes of a fragment read (16 consecutive pixels, same channel chunk) hit 16 distinct bank quads.
// Wave layout / accumulators / epilogue are those of conv_igemm.hip (conv_epilogue.h).
#include "common.h"
#include "kernels.h"
#include "conv_epilogue.h"
#include <stdlib.h>

#define PA_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define PA_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

// 16-byte slot swizzle of halo pixel p.  128 channels (16 slots = one 256-byte bank row per pixel): bits (0,1,2,0)
// of p -- found by exhaustive search over linear maps: with ds_read_b128's lane groups {0-3,12-15,20-27},... a
// fragment read touches pixels b+{0..3,12..15} at chunk c and b+{4..11} at chunk c^1, and this map keeps the 16
// slots distinct for EVERY base b, i.e. for every tap shift (p & 15 conflicts 2-way for odd shifts: 10 % of the
// LDS cycles, SQ_LDS_BANK_CONFLICT).  64 channels (two pixels per bank row): (p >> 1) & 7.
template <int CPP>
__device__ __forceinline__ int halo_sw(int p) { return CPP == 16 ? ((p & 7) | ((p & 1) << 3)) : ((p >> 1) & 7); }

template <int CIN, int BN, int LDMODE>
__global__ __launch_bounds__(256, 2) void conv3x3_tile_kernel(PaConvArgs a) {
    constexpr int TH = 8, TW = 16, PW = TW + 2, HP = (TH + 2) * PW;       // 180 halo pixels
    constexpr int CPP = CIN / 8;                                         // 16-byte chunks per pixel
    constexpr int NI = BN / 32, MI = 4;
    constexpr int KT = CIN / 64;                                         // 64-channel slices per tap
    constexpr int PSTEP = 256 / CPP;                                     // halo pixels staged per pass
    constexpr int NPASS = (HP + PSTEP - 1) / PSTEP;
    // ONE shared object (a second one makes hipcc drain vmcnt(0) before every ds_read of the pipeline)
    __shared__ __attribute__((aligned(16))) bf16 lds[HP * CIN + 2 * BN * 64];
    bf16* halo = lds;
    bf16* wbuf = lds + HP * CIN;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;
    const int tiles_x = a.W / TW, tiles_y = a.H / TH;
    int t = blockIdx.x;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    const int b = t / tiles_y;
    const int y0 = ty * TH, x0 = tx * TW;
    const int n0 = blockIdx.y * BN;
    const int K = 9 * CIN;

    // ---- weight slices: wave w streams LDS rows [w*BN/4, (w+1)*BN/4) in NI instructions of 8 rows x 8 slots
    const bf16* wsrc[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int lr = wave * (BN / 4) + i * 8 + (lane >> 3);
        const int slot = lane & 7;
        wsrc[i] = a.w + (size_t)(n0 + pa_weight_row_of_lds_row<BN, NI>(lr)) * K + ((slot ^ (lr & 7)) << 3);
    }
    auto issue_w = [&](int it, int buf) {
        const int koff = (it / KT) * CIN + (it % KT) * 64;
#pragma unroll
        for (int i = 0; i < NI; ++i)
            __builtin_amdgcn_global_load_lds(PA_GLOBAL_PTR(wsrc[i] + koff),
                                             PA_LDS_PTR(wbuf + buf * (BN * 64) + (wave * (BN / 4) + i * 8) * 64), 16, 0, 0);
    };
    issue_w(0, 0);

    // ---- halo staging (single pass over the input, transform applied here)
    {
        const int chunk = tid % CPP;                 // the same for every pass of a thread (256 % CPP == 0)
        const int c = chunk * 8;
        float k0[8], k1[8], k2[8];
        if (LDMODE != PA_LD_PLAIN) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                k0[j] = a.in.k0[c + j]; k1[j] = a.in.k1[c + j];
                if (LDMODE == PA_LD_LIN2) k2[j] = a.in.k2[c + j];
            }
        }
        const size_t img = (size_t)b * a.H * a.W;
        constexpr int UN = LDMODE == PA_LD_LIN2 ? 6 : 12;      // loads in flight per thread before the first transform
#pragma unroll
        for (int p0 = 0; p0 < NPASS; p0 += UN) {
            bf16x8 ra[UN], rq[UN];
            bool ok[UN];
            int hp[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                hp[u] = (p0 + u) * PSTEP + tid / CPP;
                const int hy = hp[u] / PW, hx = hp[u] - hy * PW;
                const int y = y0 + hy - 1, x = x0 + hx - 1;
                ok[u] = hp[u] < HP && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
                // unconditional (clamped) loads: a branch around a load makes hipcc wait vmcnt(0) per element
                const size_t idx = ok[u] ? (img + (size_t)y * a.W + x) * CIN + c : 0;
                ra[u] = *reinterpret_cast<const bf16x8*>(a.in.p + idx);
                if (LDMODE == PA_LD_LIN2) rq[u] = *reinterpret_cast<const bf16x8*>(a.in.q + idx);
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                if ((p0 + u) < NPASS && hp[u] < HP) {
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
                    if (!ok[u]) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) o[j] = (bf16)0.f;
                    }
                    *reinterpret_cast<bf16x8*>(halo + hp[u] * CIN + ((chunk ^ halo_sw<CPP>(hp[u])) << 3)) = o;
                }
            }
        }
    }

    f32x4 acc[NI][MI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int frow = lane & 15, fchk = lane >> 4;
    int pbase[MI], boff[NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) pbase[mi] = (wm * 4 + mi + 1) * PW + frow + 1;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int row = wn * (BN / 2) + ni * 16 + frow;
        boff[ni] = row * 64 + ((fchk ^ (row & 7)) << 3);
    }

    for (int tap = 0; tap < 9; ++tap) {
        const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
        const int toff = dy * PW + dx;
        // element offset of k-chunk `fchk` of the tap-shifted pixel; the other chunks of the pixel are reached by
        // XOR-ing the chunk bits (the swizzle is an XOR on the same bits), so one address per (tap, fragment)
        int aoff[MI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int p = pbase[mi] + toff;
            aoff[mi] = p * CIN + ((fchk ^ halo_sw<CPP>(p)) << 3);
        }
#pragma unroll
        for (int kh = 0; kh < KT; ++kh) {
            const int it = tap * KT + kh;
            if (it + 1 < 9 * KT) issue_w(it + 1, (it + 1) & 1);
            const bf16* Bs = wbuf + (it & 1) * (BN * 64);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                bf16x8 fa[MI], fw[NI];
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
                    fa[mi] = *reinterpret_cast<const bf16x8*>(halo + (aoff[mi] ^ ((kh * 8 + kk * 4) << 3)));
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    fw[ni] = *reinterpret_cast<const bf16x8*>(Bs + (boff[ni] ^ ((kk * 4) << 3)));
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
                        acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[ni], fa[mi], acc[ni][mi], 0, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }

    pa_conv_epilogue<BN, NI, MI>(a, acc, n0, wm, wn,
                                 [&](int mi) { return (b * a.H + y0 + wm * 4 + mi) * a.W + x0 + (lane & 15); },
                                 reinterpret_cast<float*>(lds), (int)blockIdx.x);
}

template <int CIN, int BN>
static void launch_tile_ld(const PaConvArgs& a, dim3 grid, hipStream_t st) {
    switch (a.in.mode) {
        case PA_LD_PLAIN: hipLaunchKernelGGL((conv3x3_tile_kernel<CIN, BN, PA_LD_PLAIN>), grid, dim3(256), 0, st, a); break;
        case PA_LD_BNRELU: hipLaunchKernelGGL((conv3x3_tile_kernel<CIN, BN, PA_LD_BNRELU>), grid, dim3(256), 0, st, a); break;
        default: hipLaunchKernelGGL((conv3x3_tile_kernel<CIN, BN, PA_LD_LIN2>), grid, dim3(256), 0, st, a); break;
    }
}

bool pa_conv3x3_tile_supported(const PaConvArgs& a) {
    return a.taps == 9 && (a.Cin == 64 || a.Cin == 128) && a.Cout % 64 == 0 && a.H % 8 == 0 && a.W % 16 == 0;
}

int pa_launch_conv3x3_tile(const PaConvArgs& a, hipStream_t st, int* stat_rows) {
    if (!pa_conv3x3_tile_supported(a)) { pa_set_error_msg("pa_launch_conv3x3_tile: unsupported shape"); return 1; }
    const int tiles = a.B * (a.H / 8) * (a.W / 16);
    if (stat_rows) *stat_rows = tiles;
    if (a.ep.rows_out) *a.ep.rows_out = tiles;
    static int n64 = -1;
    if (n64 < 0) n64 = getenv("PA_CONV3_BN64") ? 1 : 0;          // experiment: 64-channel halves
    const bool bigN = a.Cout % 128 == 0 && !n64;
    dim3 grid(tiles, a.Cout / (bigN ? 128 : 64));
    if (a.Cin == 128) { if (bigN) launch_tile_ld<128, 128>(a, grid, st); else launch_tile_ld<128, 64>(a, grid, st); }
    else { if (bigN) launch_tile_ld<64, 128>(a, grid, st); else launch_tile_ld<64, 64>(a, grid, st); }
    return (int)hipGetLastError();
}
