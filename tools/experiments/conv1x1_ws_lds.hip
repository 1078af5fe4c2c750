// SHELVED EXPERIMENT of round 6 (not in the build; to try it: copy to pose_adv_aug_amd/csrc/conv1x1_ws.hip, add `conv1x1_ws` to SRCS of build.sh,
// declare pa_conv1x1_ws_supported / pa_launch_conv1x1_ws in kernels.h and dispatch to it in pa_launch_conv in front of the row-tile kernel).
// Parity-green (tests/test_gpu_conv.py incl. three shapes added for it, test_gpu_local.py, test_gpu_net.py); measured (tools/bench_cold.py, 24 x 64 x 64,
// us per launch hot / cold; row-tile kernel -> this file's forward form with 1 tile of prefetch -> with 3):
//   256 -> 128 plain                21.2 / 35.1  ->  18.7 / 34.5  ->  21.3 / 38.3
//   256 -> 128 BatchNorm+ReLU, stats 23.4 / 37.5  ->  21.6 / 36.5  ->  23.9 / 40.2
//   256 -> 128 ... + addend          25.5 / 45.7  ->  21.2 / 38.9  ->  23.7 / 41.0
//   128 -> 256 plain                 21.8 / 32.0  ->  24.5 / 38.5  ->  25.4 / 41.2
//   128 -> 256 ... + addend          29.7 / 51.8  ->  26.0 / 46.5  ->  28.0 / 47.0
//   the generic form (shared epilogues) data gradient with BatchNorm backward: 37.7 / 61.2 -> 78.6 / 95.7
// and in the step: 5.76 -> 5.74 ms (forward form, one box, three pairs: noise).  With ONE workgroup per CU every barrier and every wait of the
// tile loop is exposed (3 barriers per 64-row tile), which costs what the resident weights and the prefetch return; more tiles in flight made
// it slower.  The three-workgroup row-tile kernel stays.
// 1x1 convolution (forward and data gradient) with the WEIGHTS RESIDENT IN LDS and PERSISTENT workgroups, gfx950 (round 6).
//
// Why: cycle stamps of the row-tile kernel (conv1x1_tile.hip, round 5) show a workgroup as a chain of memory round trips -- stage 64 rows,
// then one L2 round trip PER WEIGHT SLICE (4 - 8 of them, 4.5 K cycles for 1 K cycles of MFMA), then the epilogue -- and the forward 256 -> 128
// layer at 64 x 64 moving its 75 MB at 3.4 TB/s where the part streams 5.5 - 6.2.  Every workgroup of a launch streams the SAME 64 KB of weights
// through its ring.  Here a workgroup
//   * loads the whole weight matrix [Cout][Cin] (<= 128 KB bf16: the 128 <-> 256 and 256 -> 256 layers of the residual blocks, reference
//     models/asn_stacked_hg.py:17,23,25) into LDS ONCE with global_load_lds,
//   * then walks over its row tiles (one workgroup per CU, tile = blockIdx.x + i * gridDim.x): the K loop of a tile has no memory access and no
//     barrier at all, so the 16-byte loads of the NEXT tile's rows -- requested into registers right before it -- stay in flight through the K
//     loop and the epilogue of this tile (vmcnt is in-order: with a weight ring every slice wait would force them to land first; that is what
//     bounded the persistent ring form of round 3), and the epilogue's stores drain under the next tile's K loop instead of at a kernel end.
//   * 512 threads = 8 waves = 4 (rows: 16 each) x 2 (channels: 64 each) per 64-row x 128-channel block, two waves per SIMD.
// LDS: weights (64 - 128 KB) + ONE region that holds the staged activation tile during the K loop and the fp32 epilogue tile after it.
// The pending BatchNorm+ReLU / BatchNorm backward of the input is applied when the prefetched rows are written to LDS; epilogues are those
// of conv_epilogue.h (all modes), one partial-statistics row per TILE like the row-tile kernel (the finalize kernels see the same rows).
#include "common.h"
#include "kernels.h"
#include "conv_epilogue.h"
#include <stdlib.h>
#include <type_traits>

#define PA_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define PA_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define PA_CONV1_WS_DEFAULT 1

// CIN: input channels (128 / 256); NB: 128-channel output blocks (1 / 2); LDMODE: transform of the input on load
template <int CIN, int NB, int LDMODE>
__global__ __launch_bounds__(512, 1) void conv1x1_ws_kernel(PaConvArgs a, int ntiles) {
    constexpr int BM = 64, BN = 128, NT = 512;
    constexpr int CPP = CIN / 8;                     // 16-byte chunks per pixel row
    constexpr int NI = 4, MI = 1;                    // wave tile: 16 rows x 64 channels
    constexpr int KH = CIN / 64;                     // 64-channel weight slices per output block
    constexpr int PSTEP = NT / CPP, NPASS = BM / PSTEP;      // rows staged per pass (16 / 32), passes (4 / 2)
    constexpr int TREG = (BM * CIN * 2 > BM * BN * 4) ? BM * CIN * 2 : BM * BN * 4;      // bytes of the activation / epilogue region (32 KB)
    // ONE shared object: [weights: NB x KH slices of [128][64]] [activation tile | fp32 epilogue tile]
    __shared__ __attribute__((aligned(16))) bf16 lds[NB * KH * BN * 64 + TREG / 2];
    bf16* wres = lds;
    bf16* As = lds + NB * KH * BN * 64;
    float* T = reinterpret_cast<float*>(As);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 3, wn = wave >> 2;
    const int M = a.B * a.H * a.W;
    PA_SET_MAIN_PRIO();

    // ---- the whole weight matrix, once: slice (nb, kh) = rows nb*128 .. +127, k = kh*64 .. +63, 128-byte rows, 16-byte slot ^ (row & 7)
    {
#pragma unroll
        for (int s = 0; s < NB * KH; ++s) {
            const int nb = s / KH, kh = s % KH;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int lr = wave * 16 + i * 8 + (lane >> 3);
                const int wr = pa_weight_row_of_lds_row<BN, NI>(lr);
                const int wc = ((lane & 7) ^ (lr & 7)) << 3;
                __builtin_amdgcn_global_load_lds(PA_GLOBAL_PTR(a.w + (size_t)(nb * BN + wr) * CIN + kh * 64 + wc),
                                                 PA_LDS_PTR(wres + s * (BN * 64) + (wave * 16 + i * 8) * 64), 16, 0, 0);
            }
        }
    }

    // ---- per-thread constants of the input transform (the same chunk in every pass: NT % CPP == 0)
    const int chunk = tid % CPP, c = chunk * 8, prow = tid / CPP;
    float k0[8], k1[8], k2[8];
    if (LDMODE != PA_LD_PLAIN) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            k0[j] = a.in.k0[c + j]; k1[j] = a.in.k1[c + j];
            if (LDMODE == PA_LD_LIN2) k2[j] = a.in.k2[c + j];
        }
    }
    bf16x8 ra[NPASS], rq[NPASS];
    // the tile's rows -> registers.  NO condition anywhere near the loads (a uniform `tile < ntiles` became a branch around them and hipcc
    // then waits vmcnt(0) per load): the tile index is clamped with a scalar min -- the last workgroups re-request the last tile, harmlessly --
    // and M is a multiple of 64 (pa_conv1x1_ws_supported), so every row exists
    auto request = [&](int tile) {
        const int m0 = min(tile, ntiles - 1) * BM;
#pragma unroll
        for (int u = 0; u < NPASS; ++u) {
            const size_t idx = (size_t)(m0 + u * PSTEP + prow) * CIN + c;
            ra[u] = *reinterpret_cast<const bf16x8*>(a.in.p + idx);
            if (LDMODE == PA_LD_LIN2) rq[u] = *reinterpret_cast<const bf16x8*>(a.in.q + idx);
        }
    };
    int tile = (int)blockIdx.x;
    request(tile);

    const int frow = lane & 15, fchk = lane >> 4;
    bool first = true;
    for (; tile < ntiles; tile += (int)gridDim.x) {
        const int m0 = tile * BM;
        // ---- (a) transform the prefetched rows and write the activation tile (the region is free: the previous tile's epilogue is behind a barrier)
#pragma unroll
        for (int u = 0; u < NPASS; ++u) {
            const int row = u * PSTEP + prow;
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
            const int sw = CPP >= 16 ? (row & 15) : ((row >> 1) & 7);
            *reinterpret_cast<bf16x8*>(As + row * CIN + ((chunk ^ sw) << 3)) = o;
            if (LDMODE == PA_LD_LIN2 && a.dz_out)
                *reinterpret_cast<bf16x8*>(a.dz_out + (size_t)(m0 + row) * CIN + c) = o;
        }
        if (first) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); first = false; }      // (the weights have landed too)
        pa_lds_barrier();                            // (LDS-only: __syncthreads() would also wait for the previous tile's output stores)
        // ---- (b) the NEXT tile's rows: in flight through this tile's K loop and epilogue
        request(tile + (int)gridDim.x);
        // ---- (c) K loop: no memory access, no barrier
        f32x4 acc[NB][NI][MI];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) acc[nb][ni][0] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kh = 0; kh < KH; ++kh) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int ch = kh * 8 + kk * 4 + fchk;
                const int row = wm * 16 + frow;
                const int sw = CPP >= 16 ? (row & 15) : ((row >> 1) & 7);
                const bf16x8 fa = *reinterpret_cast<const bf16x8*>(As + row * CIN + ((ch ^ sw) << 3));
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const bf16* Bs = wres + (nb * KH + kh) * (BN * 64);
                    bf16x8 fw[NI];
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) {
                        const int wrow = wn * (BN / 2) + ni * 16 + frow;
                        fw[ni] = *reinterpret_cast<const bf16x8*>(Bs + wrow * 64 + (((fchk + 4 * kk) ^ (wrow & 7)) << 3));
                    }
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) acc[nb][ni][0] = PA_MFMA_16x16x32(fw[ni], fa, acc[nb][ni][0]);
                }
            }
        }
        pa_lds_barrier();                            // every wave is done with the activation tile: the region becomes the epilogue tile
        // ---- (d) epilogues (one per 128-channel block; the statistics row of this TILE)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            pa_conv_epilogue_auto<BN, NI, MI, true, false, NT>(a, acc[nb], nb * BN, wm, wn,
                                                               [&](int wr, int mi, int p) { const int m = m0 + wr * 16 + p; return m < M ? m : -1; },
                                                               T, tile);
            pa_lds_barrier();                        // T is read: the next block's pass / the next tile's rows may overwrite it
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------------------
// FORWARD specialisation (the first form above measured on par with the row-tile kernel: with ONE workgroup per CU every wait inside the tile
// loop is exposed, and the shared epilogues wait vmcnt(0) -- for their bias / addend loads and, at the loop head, for the previous tile's
// STORES -- which also drains the prefetch).  Here the tile loop has NO vm wait except the counted ones the compiler derives for the
// prefetched registers (no branch near a memory instruction):
//   * two LDS regions R[0] / R[1]: K loop(t) reads R[b]; the rows of tile t+1 (requested before K loop(t)) are transformed into R[b^1] right
//     after it; R[b] then serves as the fp32 epilogue tile of tile t
//   * bias and the constants of both transforms are loaded ONCE; the addend rows of tile t+1 are requested at the end of tile t's epilogue
//   * the per-channel statistics are accumulated in registers over ALL tiles of the workgroup: ONE partial row per workgroup at the end
//     (256 rows instead of 1536 for the finalize kernel), two barriers per 128-channel block and tile
// Modes: input PLAIN / BNRELU; output PLAIN / STATS; addend none / PLAIN / BNRELU (conv3's shortcut).
template <int CIN, int NB, int LDMODE>
__global__ __launch_bounds__(512, 1) void conv1x1_wsf_kernel(PaConvArgs a, int ntiles) {
    constexpr int BM = 64, BN = 128, NT = 512;
    constexpr int CPP = CIN / 8;
    constexpr int NI = 4;
    constexpr int KH = CIN / 64;
    constexpr int PSTEP = NT / CPP, NPASS = BM / PSTEP;
    constexpr int RB = 32768;                        // bytes of THE region: the activation tile (16 / 32 KB), then the fp32 epilogue tile (32 KB)
    constexpr int CPR = BN / 8;                      // 16-byte output chunks per pixel row of a block
    __shared__ __attribute__((aligned(16))) bf16 lds[NB * KH * BN * 64 + RB / 2 + NB * BN * 8];
    bf16* wres = lds;
    bf16* R0 = lds + NB * KH * BN * 64;
    float4* ctab = reinterpret_cast<float4*>(R0 + RB / 2);      // [NB][128] {bias, addend k0, addend k1, -}: entry j * 16 + chunk (conflict-free)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 3, wn = wave >> 2;
    const int N = a.Cout;
    PA_SET_MAIN_PRIO();
#pragma unroll
    for (int s = 0; s < NB * KH; ++s) {
        const int nb = s / KH, kh = s % KH;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int lr = wave * 16 + i * 8 + (lane >> 3);
            const int wr = pa_weight_row_of_lds_row<BN, NI>(lr);
            const int wc = ((lane & 7) ^ (lr & 7)) << 3;
            __builtin_amdgcn_global_load_lds(PA_GLOBAL_PTR(a.w + (size_t)(nb * BN + wr) * CIN + kh * 64 + wc),
                                             PA_LDS_PTR(wres + s * (BN * 64) + (wave * 16 + i * 8) * 64), 16, 0, 0);
        }
    }
    // ---- constants, once
    const int chunk = tid % CPP, c = chunk * 8, prow = tid / CPP;            // staging side: 8 input channels of row prow (+ pass * PSTEP)
    const int ochunk = tid % CPR, orow = tid / CPR;                          // epilogue side: 8 output channels of rows orow, orow + 32
    float k0[8], k1[8];
    if (LDMODE == PA_LD_BNRELU) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { k0[j] = a.in.k0[c + j]; k1[j] = a.in.k1[c + j]; }
    }
    const int amode = a.add1.mode;
    for (int i = tid; i < NB * BN; i += NT) {
        const int nb = i / BN, l = i % BN;
        ctab[nb * BN + (l & 7) * CPR + (l >> 3)] = make_float4(a.bias ? a.bias[i] : 0.f, amode == PA_LD_BNRELU ? a.add1.k0[i] : 1.f,
                                                                amode == PA_LD_BNRELU ? a.add1.k1[i] : 0.f, 0.f);
    }
    // (mode PLAIN is the BNRELU arithmetic without the clamp: v = 1 * p + 0 exactly; NONE: the pointer below reads the OUTPUT tensor's own
    // rows -- any readable address -- and the value is multiplied away)
    const bf16* addp = amode == PA_LD_NONE ? a.out : a.add1.p;
    const float again = amode == PA_LD_NONE ? 0.f : 1.f;
    const float alo = amode == PA_LD_BNRELU ? 0.f : -3.0e38f;               // lower clamp of the addend: ReLU or none

    // D register sets: the rows (and addend rows) of the next D - 1 tiles are in flight at any time -- a CU needs ~100 KB in flight to draw its
    // share of the HBM bandwidth at the loaded latency (one tile ahead: 32 KB per CU, 2.2 TB/s cold; Little's law)
    constexpr int D = 3;
    bf16x8 ra[D][NPASS];
    auto request = [&](auto SC, int tile) {
        constexpr int S = decltype(SC)::value;
        const int m0 = min(tile, ntiles - 1) * BM;
#pragma unroll
        for (int u = 0; u < NPASS; ++u) ra[S][u] = *reinterpret_cast<const bf16x8*>(a.in.p + (size_t)(m0 + u * PSTEP + prow) * CIN + c);
    };
    auto stage = [&](auto SC, bf16* As) {
        constexpr int S = decltype(SC)::value;
#pragma unroll
        for (int u = 0; u < NPASS; ++u) {
            const int row = u * PSTEP + prow;
            bf16x8 o;
            if (LDMODE == PA_LD_PLAIN) o = ra[S][u];
            else {
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (bf16)fmaxf(fmaf(k0[j], (float)ra[S][u][j], k1[j]), 0.f);
            }
            const int sw = CPP >= 16 ? (row & 15) : ((row >> 1) & 7);
            *reinterpret_cast<bf16x8*>(As + row * CIN + ((chunk ^ sw) << 3)) = o;
        }
    };
    bf16x8 ea[D][NB][2];                             // addend rows orow, orow + 32 of each block, per set
    auto request_add = [&](auto SC, int tile) {
        constexpr int S = decltype(SC)::value;
        const int m0 = min(tile, ntiles - 1) * BM;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int sw = 0; sw < 2; ++sw)
                ea[S][nb][sw] = *reinterpret_cast<const bf16x8*>(addp + (size_t)(m0 + sw * 32 + orow) * N + nb * BN + ochunk * 8);
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;

    int tile = (int)blockIdx.x;
    const int step = (int)gridDim.x;
    // iteration i works on tile t_i = blockIdx.x + i * step with addend set i % D; the rows of t_(i+1) sit in set (i+1) % D
    request(I0{}, tile);
    stage(I0{}, R0);                                 // (waits for the rows)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // ... and for the weights
    pa_lds_barrier();                                // (also publishes the constant table)
    // requests in the order they are consumed (vmcnt is in-order): addend t0, rows t1, addend t1, rows t2, addend t2, rows t3 (-> set 0)
    request_add(I0{}, tile);
    request(I1{}, tile + step);
    request_add(I1{}, tile + step);
    request(I2{}, tile + 2 * step);
    request_add(I2{}, tile + 2 * step);
    request(I0{}, tile + 3 * step);

    float s1[NB][8], s2[NB][8];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int j = 0; j < 8; ++j) { s1[nb][j] = 0.f; s2[nb][j] = 0.f; }
    const int frow = lane & 15, fchk = lane >> 4;
    const int wrow = wm * 16 + frow;
    const int wslot0 = (wn * (BN / 2)) / 4 + 2 * (lane >> 4);
    // TWO regions would need 64 KB next to 64 KB of weights; ONE region serves both roles when the next tile's rows are staged AFTER the
    // epilogue has read it (a third barrier per tile) -- kept simple: region = R0 for the K loop and the epilogue, rows staged behind the last read
    auto iteration = [&](auto SC) {
        constexpr int S = decltype(SC)::value, S1 = (S + 1) % D;
        const int m0 = tile * BM;
        f32x4 acc[NB][NI];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) acc[nb][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kh = 0; kh < KH; ++kh) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int ch = kh * 8 + kk * 4 + fchk;
                const int sw = CPP >= 16 ? (wrow & 15) : ((wrow >> 1) & 7);
                const bf16x8 fa = *reinterpret_cast<const bf16x8*>(R0 + wrow * CIN + ((ch ^ sw) << 3));
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const bf16* Bs = wres + (nb * KH + kh) * (BN * 64);
                    bf16x8 fw[NI];
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) {
                        const int r = wn * (BN / 2) + ni * 16 + frow;
                        fw[ni] = *reinterpret_cast<const bf16x8*>(Bs + r * 64 + (((fchk + 4 * kk) ^ (r & 7)) << 3));
                    }
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) acc[nb][ni] = PA_MFMA_16x16x32(fw[ni], fa, acc[nb][ni]);
                }
            }
        }
        float* T = reinterpret_cast<float*>(R0);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            pa_lds_barrier();                        // the region is free (K loop / the previous block's reads are done)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
                *reinterpret_cast<f32x4*>(T + wrow * BN + (((wslot0 + 8 * (ni >> 1) + (ni & 1)) ^ (wrow & 7)) << 2)) = acc[nb][ni];
            pa_lds_barrier();
#pragma unroll
            for (int sw = 0; sw < 2; ++sw) {
                const int r = sw * 32 + orow;
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(T + r * BN + (((2 * ochunk) ^ (r & 7)) << 2));
                const f32x4 v1 = *reinterpret_cast<const f32x4*>(T + r * BN + (((2 * ochunk + 1) ^ (r & 7)) << 2));
                bf16x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float4 k = ctab[nb * BN + j * CPR + ochunk];
                    const float e = again * fmaxf(fmaf(k.y, (float)ea[S][nb][sw][j], k.z), alo);
                    const float v = (j < 4 ? v0[j & 3] : v1[j & 3]) + k.x + e;
                    o[j] = (bf16)v;
                    const float rv = (float)o[j];
                    s1[nb][j] += rv;
                    s2[nb][j] += rv * rv;
                }
                *reinterpret_cast<bf16x8*>(a.out + (size_t)(m0 + r) * N + nb * BN + ochunk * 8) = o;
            }
        }
        pa_lds_barrier();                            // the last block's tile is read: the region takes the next tile's rows
        stage(std::integral_constant<int, S1>{}, R0);            // (counted wait for the rows of t_(i+1), requested D - 1 tiles ago)
        pa_lds_barrier();
        // this iteration's addend set and the rows' set just consumed are free: t_(i+D)'s addend, t_(i+1+D)'s rows
        request_add(SC, tile + D * step);
        request(std::integral_constant<int, S1>{}, tile + (D + 1) * step);
        tile += step;
    };
    while (true) {
        iteration(I0{}); if (tile >= ntiles) break;
        iteration(I1{}); if (tile >= ntiles) break;
        iteration(I2{}); if (tile >= ntiles) break;
    }
    // ---- ONE partial-statistics row per workgroup
    if (a.ep.mode == PA_OUT_STATS) {
        float* T = reinterpret_cast<float*>(R0);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            pa_lds_barrier();
            const f32x2 v = pa_stats_reduce<BN, NT, false>(s1[nb], s2[nb], T);
            if (tid < BN) *reinterpret_cast<f32x2*>(a.ep.stats + ((size_t)blockIdx.x * N + nb * BN + tid) * 2) = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------ host side
static int ws_on() {
    static int on = -1;
    if (on < 0) { const char* e = pa_getenv("PA_CONV1_WS"); on = e ? atoi(e) : PA_CONV1_WS_DEFAULT; }
    return on;
}

static int ws_cus() {
    static int cus = -1;
    if (cus < 0) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    }
    return cus;
}

// bit 1: forward launches, bit 2: data gradients (tuning builds: PA_CONV1_WS=<mask>)
bool pa_conv1x1_ws_supported(const PaConvArgs& a) {
    const int on = ws_on();
    if (!on || a.taps != 1 || a.fin.rows > 0 || a.out2) return false;
    // (256 -> 256 -- linear / forth_conv: 128 KB of weights leave no room; its instances spill 4 - 54 registers: the row-tile kernel keeps them)
    if (!((a.Cin == 128 && (a.Cout == 128 || a.Cout == 256)) || (a.Cin == 256 && a.Cout == 128))) return false;
    if (a.in.mode != PA_LD_PLAIN && a.in.mode != PA_LD_BNRELU && a.in.mode != PA_LD_LIN2) return false;
    if (a.in.mode == PA_LD_LIN2 ? !(on & 2) : !(on & 1)) return false;
    // the epilogue instances of a 512-thread kernel: forward modes, or the LDS form of the BatchNorm-backward epilogue
    if (a.ep.mode == PA_OUT_BWD && !pa_bwd_epilogue_lds_ok(a)) return false;
    const long M = (long)a.B * a.H * a.W;
    if ((size_t)M * (size_t)(a.Cin > a.Cout ? a.Cin : a.Cout) >= ((size_t)1 << 31)) return false;
    // persistent workgroups pay where every CU gets several tiles (the 64 x 64 maps at batch 24: 6 per CU); below that the three-workgroup row tiles
    static int mint = -1;
    if (mint < 0) { const char* e = pa_getenv("PA_CONV1_WS_MINTILES"); mint = e ? atoi(e) : 4; }
    return M % 64 == 0 && M / 64 >= (long)mint * ws_cus();
}

static bool wsf_takes(const PaConvArgs& a) {
    static int on = -1;
    if (on < 0) { const char* e = pa_getenv("PA_CONV1_WSF"); on = e ? atoi(e) : 1; }
    return on && (a.in.mode == PA_LD_PLAIN || a.in.mode == PA_LD_BNRELU) && (a.ep.mode == PA_OUT_PLAIN || a.ep.mode == PA_OUT_STATS) &&
           (a.add1.mode == PA_LD_NONE || a.add1.mode == PA_LD_PLAIN || a.add1.mode == PA_LD_BNRELU) && a.add2.mode == PA_LD_NONE && !a.dz_out;
}

template <int CIN, int NB>
static void launch_wsf_ld(const PaConvArgs& a, int grid, int ntiles, hipStream_t st) {
    if (a.in.mode == PA_LD_PLAIN) hipLaunchKernelGGL((conv1x1_wsf_kernel<CIN, NB, PA_LD_PLAIN>), dim3(grid), dim3(512), 0, st, a, ntiles);
    else hipLaunchKernelGGL((conv1x1_wsf_kernel<CIN, NB, PA_LD_BNRELU>), dim3(grid), dim3(512), 0, st, a, ntiles);
}

template <int CIN, int NB>
static void launch_ws_ld(const PaConvArgs& a, int grid, int ntiles, hipStream_t st) {
    switch (a.in.mode) {
        case PA_LD_PLAIN: hipLaunchKernelGGL((conv1x1_ws_kernel<CIN, NB, PA_LD_PLAIN>), dim3(grid), dim3(512), 0, st, a, ntiles); break;
        case PA_LD_BNRELU: hipLaunchKernelGGL((conv1x1_ws_kernel<CIN, NB, PA_LD_BNRELU>), dim3(grid), dim3(512), 0, st, a, ntiles); break;
        default: hipLaunchKernelGGL((conv1x1_ws_kernel<CIN, NB, PA_LD_LIN2>), dim3(grid), dim3(512), 0, st, a, ntiles); break;
    }
}

int pa_launch_conv1x1_ws(const PaConvArgs& a, hipStream_t st, int* stat_rows) {
    if (!pa_conv1x1_ws_supported(a)) { pa_set_error_msg("pa_launch_conv1x1_ws: unsupported launch"); return 1; }
    const int M = a.B * a.H * a.W, ntiles = M / 64;
    if (stat_rows) *stat_rows = ntiles;
    if (a.ep.rows_out) *a.ep.rows_out = ntiles;
    const int grid = ntiles < ws_cus() ? ntiles : ws_cus();
    if (wsf_takes(a)) {              // forward: the specialised form, ONE statistics row per workgroup
        if (stat_rows) *stat_rows = grid;
        if (a.ep.rows_out) *a.ep.rows_out = grid;
        if (a.Cin == 256) launch_wsf_ld<256, 1>(a, grid, ntiles, st);
        else if (a.Cout == 128) launch_wsf_ld<128, 1>(a, grid, ntiles, st);
        else launch_wsf_ld<128, 2>(a, grid, ntiles, st);
        return (int)hipGetLastError();
    }
    if (a.Cin == 256) launch_ws_ld<256, 1>(a, grid, ntiles, st);
    else { if (a.Cout == 128) launch_ws_ld<128, 1>(a, grid, ntiles, st); else launch_ws_ld<128, 2>(a, grid, ntiles, st); }
    return (int)hipGetLastError();
}
