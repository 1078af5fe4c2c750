// 3x3 'same' convolution (forward and data gradient) as a HALO-TILE implicit GEMM on bf16 MFMA, gfx950.
//
// The generic kernel (conv_igemm.hip) re-loads and re-transforms the activation tile for each of the 9
// taps and spends ~10 VALU instructions per MFMA doing so.  Here a workgroup owns an 8 x 16 block of
// output pixels of one image and
//   * stages the (8+2) x (16+2) input halo ONCE into LDS -- the pending BatchNorm+ReLU (PA_LD_BNRELU,
//     forward: reference models/asn_stacked_hg.py:36-41) or BatchNorm backward (PA_LD_LIN2, dgrad) is
//     applied during this single pass, zero padding is written as zeros, 1.4x read amplification
//     instead of 9x;
//   * streams the weight slice [BN][32] of every (tap, 32-channel slice) with global_load_lds (16 B per
//     lane, no VGPR staging, no ds_write) through a ring of 4 LDS buffers: slices t+1, t+2 are in flight
//     while slice t feeds the MFMAs (counted s_waitcnt vmcnt + raw s_barrier); the 16-byte slot swizzle
//     is realised on the per-lane SOURCE address (the LDS side of global_load_lds is lane-linear);
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

// TW x TH = spatial block of one image; a workgroup always owns 128 output pixels = IMG blocks.  16 x 8: one block of
// a big map.  8 x 8 / 4 x 4: the low-resolution levels, where the block IS the image and a workgroup takes 2 / 8
// images (the generic kernel needs 20 us for these 384..1536-pixel problems: 18-36 serial K-steps with two barriers).
// 16 x 4 (BM = 64 pixels): smaller halo and a ring of 3 slices = 52 KB of LDS -> THREE workgroups per CU, and 1536
// instead of 768 workgroups for a 24 x 64 x 64 map (2 full rounds): staging, K loop and epilogue of different
// workgroups overlap instead of running in lockstep.
// SPS = weight slices consumed per K-loop step (per barrier).  The small problems -- the 8x8 / 4x4 maps (3..12 workgroups on
// the whole chip) and the 16 x 4 tiles of the 32x32 level (1.5 workgroups per CU) -- are bound by the LATENCY of their 36
// serial steps (barrier, LDS-DMA wait, fragment reads: ~0.35 us each for 8-16 MFMAs per wave), not by MFMA or memory
// throughput: with a whole tap (SPS = 4) or half a tap (SPS = 2) per barrier the chain is 9 / 18 steps long.
template <int N> __device__ __forceinline__ void pa_wait_vmcnt() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else static_assert(N == 0, "unsupported vmcnt");
}

// TRI (NT = 512 threads, 16 x 8 tiles): THREE pixel tiles per workgroup = 384 pixels = exactly one CU's share of a 24 x 64 x 64 map.
// The three tiles share ONE weight ring, so the L2 -> LDS weight stream per pixel is a third (it is what bounds the K loop of the
// 128-pixel kernel: 295 KB per tile against 3.9 us of MFMA work, two workgroups per CU streaming at once), every K step has 24 MFMAs
// per wave between two barriers instead of 16, and the launch is ONE full round of 256 workgroups instead of 768 on 512 slots.
// 8 waves = 4 (pixels: 6 fragments of 16 each) x 2 (channels: 4 fragments); LDS 135 KB of halos + a ring of 3 slices: one workgroup
// per CU, whose memory phases (staging, epilogue) run as chip-wide bursts.
// FIN: the instance that can carry the input's BatchNorm finalize in its prologue (bn_fin.h): instantiated for the tilings the maps with
// <= 128 statistics rows use (launch_tile_shape); all other instances are the plain kernel.
// PA_TUNING builds only: a.dbg -- phase ablation (1 no K loop, 2 no epilogue, 4 no staging: wrong results, timing) and bit 8, per-workgroup tap rotation
// NT = 512 / 1024 on the maps that give at most one workgroup per CU (8 x 8, 4 x 4, 16 x 16): with ONE wave per SIMD nothing overlaps -- cycle
// stamps (tools/conv3_clocks.py) show a K-loop step as the plain SUM of its parts (barrier ~100 cycles, each LDS-DMA issue 100-150, the ds_read
// round trip ~150, 17 per MFMA: 45-60 cycles per MFMA instead of 17), and the staging / epilogue are load / store round trips of 4 waves.  Two
// or four waves per SIMD (the same 128 pixels x 64 channels, MI = 2 or 1 fragments per wave) let one wave's MFMAs run under another's waits.
// cycle stamps (tuning builds, a.dbg & 64): entry / weight ring issued / own halo part staged / K loop / barrier / epilogue
PA_STAMP_DECL(pa_conv3_clk, pa_debug_conv3_clocks)
#define PA_STAMP(i) PA_STAMP_AT(pa_conv3_clk, a.dbg & 64, i)
template <int CIN, int BN, int LDMODE, int TW = 16, int TH = 8, int SPS = 1, bool PF = false, int NT = 256, bool FIN = false>
__global__ __launch_bounds__(NT, (NT >= 512 ? 1 : (TW == 16 ? (TH == 4 ? (SPS == 1 ? 3 : 2) : 2) : 1))) void conv3x3_tile_kernel(PaConvArgs a) {
    constexpr bool TRI = NT == 512 && TW == 16 && TH == 8;
    PA_SET_MAIN_PRIO();
    PA_STAMP(0);
#if defined(PA_TUNING) && !defined(PA_CONV3_CONSTDBG)      // (PA_EXTRA=-DPA_CONV3_CONSTDBG: the release code with the cycle stamps only)
    const int dbg = a.dbg;
#else
    constexpr int dbg = 0;
#endif
#ifdef PA_TUNING
    // a.dbg bits 8.. (tuning builds): stagger -- the workgroups of the SECOND slot of every CU (dispatch order: workgroup 256 .. 511 of a launch that
    // fills 256 CUs twice) start (dbg >> 8) x 64 x 16 cycles late, so that co-resident workgroups are not in the same phase (staging / K loop / epilogue)
    if ((a.dbg >> 8) > 0 && gridDim.x * gridDim.y > 256 && ((blockIdx.x + blockIdx.y * gridDim.x) / 256) % 2 == 1)
        for (int i = 0; i < (a.dbg >> 8); ++i) __builtin_amdgcn_s_sleep(16);
#endif
    constexpr int NW = NT / 64, WM = NW / 2;                             // waves; waves along the pixels (2 along the channels)
    constexpr int BM = TRI ? 384 : ((TW == 16 && TH == 4) ? 64 : 128);
    constexpr int IMG = BM / (TW * TH), PW = TW + 2, PHh = TH + 2, HP = IMG * PHh * PW;   // 180 / 108 / 200 / 288 / 540 halo pixels
    constexpr int CPP = CIN / 8;                                         // 16-byte chunks per pixel
    constexpr int NI = BN / 32, MI = BM / (16 * WM);
    constexpr int NSL = CIN / 32;                                        // 32-channel weight slices per tap
    constexpr int NIT = 9 * NSL;                                         // slices [BN][32]
    // LDS-DMA pieces (16 rows x 64 B of a slice, one wave-instruction) per wave: NIW per slice when a slice has at least one piece for every wave,
    // else (8 / 16 waves) the SPS * BN / 16 pieces of a STEP are dealt out, PPS per wave
    constexpr bool DEAL = 16 * NW > BN;
    constexpr int NIW = DEAL ? 0 : BN / (16 * NW), PPS = SPS * BN / (16 * NW);
    static_assert(PPS >= 1 && PPS * 16 * NW == SPS * BN, "every wave issues the same number of weight pieces per step");
    constexpr int NST = NIT / SPS, GPT = NSL / SPS;                      // K-loop steps, steps per tap
    constexpr int NBUF = TRI ? 3 : (SPS == 1 ? (BM == 64 ? 3 : 4) : 3);  // ring of NBUF step buffers [SPS][BN][32]
    constexpr int AHEAD = NBUF - 2;                                      // steps still in flight while one is consumed
    static_assert(NSL % SPS == 0, "a step stays inside one tap");
    static_assert(!TRI || (TW == 16 && TH == 8 && PF && SPS == 1), "three-tile workgroups: 16 x 8 tiles, pipelined K loop");
    constexpr int PSTEP = NT / CPP;                                      // halo pixels staged per pass
    constexpr int NPASS = (HP + PSTEP - 1) / PSTEP;
    // ONE shared object (a second one makes hipcc drain vmcnt(0) before every ds_read of the pipeline)
    __shared__ __attribute__((aligned(16))) bf16 lds[HP * CIN + NBUF * SPS * BN * 32];
    bf16* halo = lds;
    bf16* wbuf = lds + HP * CIN;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wn = wave / WM;
    const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + TH - 1) / TH;      // (ragged only for the small tiles: maps up to 8 x 8 that are not 8 x 8 / 4 x 4)
    // consecutive workgroups go to different XCDs (round robin), each with its own L2: give every XCD one contiguous range
    // of tiles so that the halo rows / columns shared by neighbouring tiles are found in that L2
    int t = (a.xcd & 1) ? (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int t0 = t * IMG;                            // TRI: first of the workgroup's three consecutive tiles
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    const int b = (t / tiles_y) * IMG;                 // first image of this workgroup (small maps: tile = image)
    const int y0 = ty * TH, x0 = tx * TW;
    const int n0 = blockIdx.y * BN;
    const int K = 9 * CIN;
    // origin of sub-tile `im` of the workgroup: small maps -- image b + im at (0, 0); TRI -- tile t0 + im of the tile grid
    // (three wave-uniform origins, selected per lane)
    int ob[3], oy[3], ox[3];
    if constexpr (TRI) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            int tt = t0 + j;
            ox[j] = (tt % tiles_x) * TW; tt /= tiles_x;
            oy[j] = (tt % tiles_y) * TH;
            ob[j] = tt / tiles_y;
        }
    }
    auto origin = [&](int im, int& bb, int& yy, int& xx) {
        if constexpr (TRI) {
            bb = im == 0 ? ob[0] : (im == 1 ? ob[1] : ob[2]);
            yy = im == 0 ? oy[0] : (im == 1 ? oy[1] : oy[2]);
            xx = im == 0 ? ox[0] : (im == 1 ? ox[1] : ox[2]);
        } else if constexpr (IMG > 1) {
            // small tiles (8 x 8 / 4 x 4): the tile IS the image (8 x 8 / 4 x 4 maps) -- or, round 6, sub-tile t0 + im of the tile grid of a map
            // whose sides are multiples of the tile (24 x 24, 12 x 12 of the 384 x 384 configuration); a workgroup's sub-tiles may lie in two images
            if (tiles_x * tiles_y == 1) { bb = b + im; yy = 0; xx = 0; }
            else {
                int tt = t0 + im;
                xx = (tt % tiles_x) * TW; tt /= tiles_x;
                yy = (tt % tiles_y) * TH;
                bb = tt / tiles_y;
            }
        } else { bb = b + im; yy = y0; xx = x0; }
    };

    // ---- weight slices [BN][32] (64-byte rows): wave w streams LDS rows [w*BN/4, (w+1)*BN/4), one instruction =
    // 16 rows x 4 slots.  Slot swizzle wsw(row) = (-(row >> 2)) & 3: conflict-free for ds_read_b128's lane groups
    // (rows 0-3/12-15 at chunk c with rows 4-11 at chunk c^1).  Slice `it` covers k = 32*it .. 32*it+31.
    const bf16* wsrc[DEAL ? PPS : NIW];
    int wj[DEAL ? PPS : 1], wrow[DEAL ? PPS : 1];      // (DEAL) slice of the step and first LDS row of the wave's piece i
#pragma unroll
    for (int i = 0; i < (DEAL ? PPS : NIW); ++i) {
        int lr;
        if constexpr (DEAL) {
            const int q = wave * PPS + i;
            wj[i] = q / (BN / 16); wrow[i] = (q % (BN / 16)) * 16;
            lr = wrow[i] + (lane >> 2);
        } else lr = wave * (BN / NW) + i * 16 + (lane >> 2);
        const int slot = lane & 3;
        wsrc[i] = a.w + (size_t)(n0 + pa_weight_row_of_lds_row<BN, NI>(lr)) * K + ((slot ^ ((-(lr >> 2)) & 3)) << 3);
    }
    // tap rotation (a.dbg & 8): workgroup i walks the taps in the order rot, rot + 1, ... (mod 9), rot = i % 9 -- every workgroup of a launch
    // streams the SAME 295 KB of weights; started together they all pull the same slice from the same L2 channels at the same time
    const int rot = (dbg & 8) ? (int)(blockIdx.x % 9) : 0;
    auto ptap = [&](int ltap) { const int p = ltap + rot; return p >= 9 ? p - 9 : p; };
    auto issue_w = [&](int st) {                       // step st = slices st*SPS .. st*SPS+SPS-1
        if constexpr (DEAL) {
#pragma unroll
            for (int i = 0; i < PPS; ++i) {
                bf16* dst = wbuf + ((st % NBUF) * SPS + wj[i]) * (BN * 32) + wrow[i] * 32;
                const int it = st * SPS + wj[i], lt = it / NSL;
                const int pit = ptap(lt) * NSL + (it - lt * NSL);
                __builtin_amdgcn_global_load_lds(PA_GLOBAL_PTR(wsrc[i] + pit * 32), PA_LDS_PTR(dst), 16, 0, 0);
            }
        } else
#pragma unroll
        for (int j = 0; j < SPS; ++j) {
            bf16* dst = wbuf + ((st % NBUF) * SPS + j) * (BN * 32) + wave * (BN / NW) * 32;
            const int it = st * SPS + j, lt = it / NSL;
            const int pit = ptap(lt) * NSL + (it - lt * NSL);
#pragma unroll
            for (int i = 0; i < NIW; ++i)
                __builtin_amdgcn_global_load_lds(PA_GLOBAL_PTR(wsrc[i] + pit * 32), PA_LDS_PTR(dst + i * 16 * 32), 16, 0, 0);
        }
    };
    issue_w(0); issue_w(1);
    if (NBUF == 4) issue_w(2);
    if (PF) issue_w(NBUF - 1);                         // (pipelined K loop: the whole ring is in flight during the staging)
    PA_STAMP(1);

    // ---- halo staging (single pass over the input, transform applied here)
    if (!(dbg & 4)) {
        const int chunk = tid % CPP;                 // the same for every pass of a thread (256 % CPP == 0)
        const int c = chunk * 8;
        float k0[8], k1[8], k2[8];
        if (LDMODE != PA_LD_PLAIN) {
            bool done = false;
            if constexpr (FIN && !TRI) {
                if (a.fin.rows > 0) {
                    // the input's BatchNorm finalize from the producer's <= 128 partial rows (bn_fin.h) instead of a launch of its own in
                    // front of this kernel; table + scratch sit in the halo region, which nobody writes before the barrier below
                    float* ktab = reinterpret_cast<float*>(halo);
                    pa_bn_fin_prologue<NT, CIN>(a.fin, CIN, ktab, ktab + 3 * CIN, blockIdx.x == 0 && blockIdx.y == 0);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        k0[j] = ktab[c + j]; k1[j] = ktab[CIN + c + j];
                        if (LDMODE == PA_LD_LIN2) k2[j] = ktab[2 * CIN + c + j];
                    }
                    __syncthreads();
                    done = true;
                }
            }
            if (!done) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    k0[j] = a.in.k0[c + j]; k1[j] = a.in.k1[c + j];
                    if (LDMODE == PA_LD_LIN2) k2[j] = a.in.k2[c + j];
                }
            }
        }
        constexpr int UN = LDMODE == PA_LD_LIN2 ? 6 : (TRI ? 9 : 12);      // loads in flight per thread before the first transform
#pragma unroll
        for (int p0 = 0; p0 < NPASS; p0 += UN) {
            bf16x8 ra[UN], rq[UN];
            bool ok[UN];
            int hp[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                hp[u] = (p0 + u) * PSTEP + tid / CPP;
                const int im = IMG == 1 ? 0 : hp[u] / (PHh * PW);
                const int hr = hp[u] - im * (PHh * PW);
                const int hy = hr / PW, hx = hr - hy * PW;
                int bb, yy, xx;
                origin(im, bb, yy, xx);
                const int y = yy + hy - 1, x = xx + hx - 1;
                ok[u] = hp[u] < HP && bb < a.B && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
                // unconditional (clamped) loads: a branch around a load makes hipcc wait vmcnt(0) per element
                const size_t idx = ok[u] ? (((size_t)bb * a.H + y) * a.W + x) * CIN + c : 0;
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
                    if (LDMODE == PA_LD_LIN2 && a.dz_out && blockIdx.y == 0 && ok[u]) {       // interior pixels: this tile owns them
                        const int im = IMG == 1 ? 0 : hp[u] / (PHh * PW);
                        const int hr = hp[u] - im * (PHh * PW);
                        const int hy = hr / PW, hx = hr - hy * PW;
                        int bb, yy, xx;
                        origin(im, bb, yy, xx);
                        if (hy >= 1 && hy < PHh - 1 && hx >= 1 && hx < PW - 1)
                            *reinterpret_cast<bf16x8*>(a.dz_out + (((size_t)bb * a.H + yy + hy - 1) * a.W + xx + hx - 1) * CIN + c) = o;
                    }
                }
            }
        }
    }

    f32x4 acc[NI][MI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};

    PA_STAMP(2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int frow = lane & 15, fchk = lane >> 4;
    int pbase[MI], boff[NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int l = (wm * MI + mi) * 16 + frow;                        // pixel 0..BM-1 of the workgroup
        const int im = l / (TW * TH), r = l - im * (TW * TH);
        pbase[mi] = im * (PHh * PW) + (r / TW + 1) * PW + r % TW + 1;
    }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int row = wn * (BN / 2) + ni * 16 + frow;
        boff[ni] = row * 32 + ((fchk ^ ((-(row >> 2)) & 3)) << 3);
    }

    if constexpr (PF) {
        // SOFTWARE-PIPELINED K loop (SPS == 1): the fragments of slice st + 1 are read from LDS into a second register set while
        // the MFMAs of slice st (read one step earlier) issue -- the plain loop below starts every step with a ds_read round trip
        // in front of its 16 MFMAs.  Barrier of step st: slice st + 1 has landed for every wave (its DMA was issued three steps
        // earlier; two younger slices may stay in flight), and every wave holds slice st's fragments in registers (lgkmcnt(0)),
        // so buffer st % NBUF is free for slice st + NBUF.
        static_assert(!PF || (SPS == 1 && (NBUF == 4 || NBUF == 3) && GPT % 2 == 0), "pipelined K loop: one slice per step, ring of 3 or 4, even steps per tap");
        bf16x8 fa[2][MI], fw[2][NI];
        int aoff[MI];
        auto tap_offsets = [&](int tap) {
            const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
            const int toff = dy * PW + dx;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int p = pbase[mi] + toff;
                aoff[mi] = p * CIN + ((fchk ^ halo_sw<CPP>(p)) << 3);
            }
        };
        tap_offsets(ptap(0));
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) fa[0][mi] = *reinterpret_cast<const bf16x8*>(halo + aoff[mi]);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) fw[0][ni] = *reinterpret_cast<const bf16x8*>(wbuf + boff[ni]);
        for (int tap = 0; tap < ((dbg & 1) ? 0 : 9); ++tap) {
#pragma unroll
            for (int g = 0; g < GPT; ++g) {
                const int st = tap * GPT + g;
                constexpr int dummy = 0; (void)dummy;
                const int cur = g & 1, nxt = cur ^ 1;                  // (GPT is even: the register set of a step is a compile-time index)
                if (st + 1 < NST) {
                    const int younger = NST - 2 - st;               // slices issued after st + 1 (at most NBUF - 2 of them are in flight)
                    if (younger >= 2 && NBUF == 4) pa_wait_vmcnt<2 * PPS>();
                    else if (younger >= 1) pa_wait_vmcnt<PPS>();
                    else pa_wait_vmcnt<0>();
                    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                    if (st + NBUF < NST) issue_w(st + NBUF);
                    if (g == GPT - 1) tap_offsets(ptap(tap + 1));
                    const int sub = (g + 1) % GPT;
                    const bf16* Bs = wbuf + ((st + 1) % NBUF) * (BN * 32);
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) fa[nxt][mi] = *reinterpret_cast<const bf16x8*>(halo + (aoff[mi] ^ ((sub * 4) << 3)));
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) fw[nxt][ni] = *reinterpret_cast<const bf16x8*>(Bs + boff[ni]);
                }
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
                        acc[ni][mi] = PA_MFMA_16x16x32(fw[cur][ni], fa[cur][mi], acc[ni][mi]);
            }
        }
    } else
    // K loop: step `st` is consumed while steps st+1 (, st+2) are in flight and step st+NBUF-1 is issued right after the
    // barrier into the buffer that was read in step st-1.  Counted vmcnt + raw s_barrier: __syncthreads() would drain
    // the LDS-DMA queue (vmcnt(0)) and expose one L2 round trip per slice, which is what bounded the first version
    // of this kernel (0.7 us per 64-channel slice = 30 % MFMA utilisation).
    for (int tap = 0; tap < ((dbg & 1) ? 0 : 9); ++tap) {
        const int pt = ptap(tap);
        const int dy = pt / 3 - 1, dx = pt - (pt / 3) * 3 - 1;
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
        for (int g = 0; g < GPT; ++g) {
            const int st = tap * GPT + g;
            const int rem = NST - 1 - st;                       // steps issued after this one
            const int fly = rem < AHEAD ? rem : AHEAD;          // ... that may stay in flight
            if (fly == 2) pa_wait_vmcnt<2 * PPS>();
            else if (fly == 1) pa_wait_vmcnt<PPS>();
            else pa_wait_vmcnt<0>();
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (st + NBUF - 1 < NST) issue_w(st + NBUF - 1);
#pragma unroll
            for (int j = 0; j < SPS; ++j) {
                const int sub = g * SPS + j;
                const bf16* Bs = wbuf + ((st % NBUF) * SPS + j) * (BN * 32);
                bf16x8 fa[MI], fw[NI];
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
                    fa[mi] = *reinterpret_cast<const bf16x8*>(halo + (aoff[mi] ^ ((sub * 4) << 3)));
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    fw[ni] = *reinterpret_cast<const bf16x8*>(Bs + boff[ni]);
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
                        acc[ni][mi] = PA_MFMA_16x16x32(fw[ni], fa[mi], acc[ni][mi]);
            }
        }
    }
    PA_STAMP(3);
    __syncthreads();            // every wave is done with the halo and the ring before the epilogue reuses the LDS
    PA_STAMP(4);
    if (dbg & 2) {            // (timing ablation: no epilogue; the accumulators stay alive)
        float sacc = 0.f;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) sacc += acc[ni][mi][0] + acc[ni][mi][1] + acc[ni][mi][2] + acc[ni][mi][3];
        if (sacc == 123.456f) a.out[tid] = (bf16)sacc;
        return;
    }

    pa_conv_epilogue_auto<BN, NI, MI, true, false, NT>(a, acc, n0, wm, wn,
                                     [&](int wr, int mi, int p) {
                                         const int l = (wr * MI + mi) * 16 + p;
                                         const int im = l / (TW * TH), r = l - im * (TW * TH);
                                         int bb, yy, xx;
                                         origin(im, bb, yy, xx);
                                         return (bb < a.B && yy + r / TW < a.H && xx + r % TW < a.W) ? (bb * a.H + yy + r / TW) * a.W + xx + r % TW : -1;
                                     },
                                     reinterpret_cast<float*>(lds), (int)blockIdx.x);
    PA_STAMP(5);
}

// WITHFIN: this tiling also exists as a finalize-carrying instance (BatchNorm-on-load modes only), used when the launch brings one
template <int CIN, int BN, int TW, int TH, int SPS, bool PF, bool WITHFIN = false, int NT = 256>
static void launch_tile_ld(const PaConvArgs& a, dim3 grid, hipStream_t st) {
    if constexpr (WITHFIN) {
        if (a.fin.rows > 0) {
            if (a.in.mode == PA_LD_BNRELU) hipLaunchKernelGGL((conv3x3_tile_kernel<CIN, BN, PA_LD_BNRELU, TW, TH, SPS, PF, NT, true>), grid, dim3(NT), 0, st, a);
            else hipLaunchKernelGGL((conv3x3_tile_kernel<CIN, BN, PA_LD_LIN2, TW, TH, SPS, PF, NT, true>), grid, dim3(NT), 0, st, a);
            return;
        }
    }
    switch (a.in.mode) {
        case PA_LD_PLAIN: hipLaunchKernelGGL((conv3x3_tile_kernel<CIN, BN, PA_LD_PLAIN, TW, TH, SPS, PF, NT>), grid, dim3(NT), 0, st, a); break;
        case PA_LD_BNRELU: hipLaunchKernelGGL((conv3x3_tile_kernel<CIN, BN, PA_LD_BNRELU, TW, TH, SPS, PF, NT>), grid, dim3(NT), 0, st, a); break;
        default: hipLaunchKernelGGL((conv3x3_tile_kernel<CIN, BN, PA_LD_LIN2, TW, TH, SPS, PF, NT>), grid, dim3(NT), 0, st, a); break;
    }
}

// SPS: slices per K-loop step of the latency-bound variants (128 input channels): a whole tap for the 8x8 / 4x4 maps, half a
// tap for the 16 x 4 tiles; 1 everywhere else (the 16 x 8 tiles of the big maps are throughput-bound and need their LDS for 2
// workgroups per CU).  The 16 x 8 tiles run the software-pipelined K loop (PA_CONV3_PF=0: the plain one).
template <int TW, int TH, int SPS>
static void launch_tile_shape(const PaConvArgs& a, dim3 grid, bool bigN, hipStream_t st) {
    if constexpr (TW != 16) {                      // the small maps always run 64-channel halves (bigN is false for them)
        if (a.Cin == 128) {
            if constexpr (SPS == 4) {
                static int nt = -1;
                if (nt < 0) { const char* e = pa_getenv("PA_CONV3_NT"); nt = e ? atoi(e) : 512; }      // measured 8x8: 14.1 / 10.6 / 12.8 us at 256 / 512 / 1024 threads
                if (nt == 512) { launch_tile_ld<128, 64, TW, TH, SPS, false, true, 512>(a, grid, st); return; }
                if (nt == 1024) { launch_tile_ld<128, 64, TW, TH, SPS, false, true, 1024>(a, grid, st); return; }
            }
            launch_tile_ld<128, 64, TW, TH, SPS, false, true>(a, grid, st);
        } else launch_tile_ld<64, 64, TW, TH, 1, false, true>(a, grid, st);
    } else if constexpr (TH == 8 && SPS == 1) {
        static int pf = -1;
        if (pf < 0) { const char* e = pa_getenv("PA_CONV3_PF"); pf = e ? atoi(e) : 1; }
        if (pf) {
            if (a.Cin == 128) { if (bigN) launch_tile_ld<128, 128, TW, TH, 1, true>(a, grid, st); else launch_tile_ld<128, 64, TW, TH, 1, true>(a, grid, st); }
            else { if (bigN) launch_tile_ld<64, 128, TW, TH, 1, true>(a, grid, st); else launch_tile_ld<64, 64, TW, TH, 1, true, true>(a, grid, st); }
        } else {
            if (a.Cin == 128) { if (bigN) launch_tile_ld<128, 128, TW, TH, 1, false>(a, grid, st); else launch_tile_ld<128, 64, TW, TH, 1, false>(a, grid, st); }
            else { if (bigN) launch_tile_ld<64, 128, TW, TH, 1, false>(a, grid, st); else launch_tile_ld<64, 64, TW, TH, 1, false>(a, grid, st); }
        }
    } else {
        if (a.Cin == 128) {
            if constexpr (SPS == 2) {
                // 512 threads when the launch is at most one workgroup per CU (the 16 x 16 maps in 64-channel halves: 192 workgroups, 12.0 -> 8.6 us);
                // with two workgroups per CU (32 x 32) the 256-thread workgroups already overlap each other and 512 measured slower (16.1 vs 18.2)
                static int nt = -1, cus = 0;
                if (nt < 0) {
                    int dev = 0;
                    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 0;
                    const char* e = pa_getenv("PA_CONV3_NT16"); nt = e ? atoi(e) : 512;
                }
                if (nt == 512 && (int)(grid.x * grid.y) <= cus) { if (bigN) launch_tile_ld<128, 128, TW, TH, SPS, false, true, 512>(a, grid, st); else launch_tile_ld<128, 64, TW, TH, SPS, false, true, 512>(a, grid, st); return; }
            }
            if (bigN) launch_tile_ld<128, 128, TW, TH, SPS, false, (SPS == 2)>(a, grid, st); else launch_tile_ld<128, 64, TW, TH, SPS, false, (SPS == 2)>(a, grid, st);
        }
        else { if (bigN) launch_tile_ld<64, 128, TW, TH, 1, false>(a, grid, st); else launch_tile_ld<64, 64, TW, TH, 1, false>(a, grid, st); }
    }
}

// does the instance the launcher below picks for this shape exist with the finalize prologue?  (pa_conv_takes_fin)
bool pa_conv3x3_tile_takes_fin(const PaConvArgs& a) {
    if (!pa_conv3x3_tile_supported(a)) return false;
    if (pa_getenv("PA_CONV3_SPS") || pa_getenv("PA_CONV3_PF") || pa_getenv("PA_CONV3_BM64") || pa_getenv("PA_CONV3_BN64")) return false;      // (tuning switches select other instances)
    const bool small = !(a.H % 8 == 0 && a.W % 16 == 0);
    if (small) return true;                                                    // 8 x 8 / 4 x 4 maps: <128 | 64, 64, .., SPS 4 | 1>
    const int tiles128 = a.B * (a.H / 8) * (a.W / 16);
    if (a.Cin == 128) return tiles128 < 512;                                   // 16 x 4 tiles, half a tap per step
    return a.Cout % 128 != 0;                                                  // 64 input channels: the 16 x 8 <64, 64> instance
}

// maps tiled by the 8 x 8 / 4 x 4 variants: sides multiples of 8 (that the 16 x 8 tiles do not take: W % 16 != 0) or of 4
// (any map up to 8 x 8 -- the 6 x 6 necks of the 384 x 384 configuration -- is ONE 8 x 8 tile whose pixels outside the map are masked)
static int small_tile(const PaConvArgs& a) { return (a.H % 8 == 0 && a.W % 8 == 0) ? 8 : ((a.H % 4 == 0 && a.W % 4 == 0) ? 4 : ((a.H <= 8 && a.W <= 8) ? 8 : 0)); }
static bool small_map(const PaConvArgs& a) { return small_tile(a) != 0; }

bool pa_conv3x3_tile_supported(const PaConvArgs& a) {
    static int nosmall = -1;
    if (nosmall < 0) nosmall = pa_getenv("PA_CONV3_NOSMALL") ? 1 : 0;
    if (a.taps != 9 || (a.Cin != 64 && a.Cin != 128) || a.Cout % 64 != 0) return false;
    if ((size_t)a.B * a.H * a.W * (size_t)(a.Cin > a.Cout ? a.Cin : a.Cout) >= ((size_t)1 << 31)) return false;      // 32-bit element offsets in the epilogue
    return (a.H % 8 == 0 && a.W % 16 == 0) || (!nosmall && small_map(a));
}

int pa_launch_conv3x3_tile(const PaConvArgs& a, hipStream_t st, int* stat_rows) {
    if (!pa_conv3x3_tile_supported(a)) { pa_set_error_msg("pa_launch_conv3x3_tile: unsupported shape"); return 1; }
    const bool small = !(a.H % 8 == 0 && a.W % 16 == 0);
    const int st_ = small ? small_tile(a) : 0;                      // side of the small tile
    const int img = small ? 128 / (st_ * st_) : 1;                  // sub-tiles per workgroup
    // 16 x 4 pixel tiles (3 workgroups per CU) where 16 x 8 tiles would leave CUs idle: measured 64x64 maps (768 tiles)
    // 39.8 vs 42.8 us in favour of 16 x 8, 32x32 maps (192 tiles) 17.6 vs 14.6 us in favour of 16 x 4
    static int bm64 = -1;
    if (bm64 < 0) { const char* e = pa_getenv("PA_CONV3_BM64"); bm64 = e ? atoi(e) : -2; }
    const int tiles128 = small ? 0 : a.B * (a.H / 8) * (a.W / 16);
    const bool half = !small && a.Cin == 128 && (bm64 == -2 ? tiles128 < 512 : bm64 != 0);
    const int tiles = small ? (a.B * ((a.H + st_ - 1) / st_) * ((a.W + st_ - 1) / st_) + img - 1) / img : a.B * (a.H / (half ? 4 : 8)) * (a.W / 16);
    if (stat_rows) *stat_rows = tiles;
    if (a.ep.rows_out) *a.ep.rows_out = tiles;
    static int n64 = -1;
    if (n64 < 0) n64 = pa_getenv("PA_CONV3_BN64") ? 1 : 0;          // experiment: 64-channel halves
    // the low-resolution levels have 3..12 pixel tiles: 64-channel halves double the number of workgroups
    // ... and so do the 16 x 4 tiles (32 x 32 / 16 x 16 maps): 64-channel blocks halve the serial chain of a workgroup (weight pieces, MFMAs,
    // epilogue) -- 32 x 32: 16.5 -> 16.1 / 19.3 -> 17.8 us (forward / data gradient), 16 x 16: see launch_tile_shape
    static int h64 = -1;
    if (h64 < 0) { const char* e = pa_getenv("PA_CONV3_HALF64"); h64 = e ? atoi(e) : 1; }
    const bool bigN = a.Cout % 128 == 0 && !n64 && !small && !(half && h64);
    dim3 grid(tiles, a.Cout / (bigN ? 128 : 64));
    static int xcd = -1;
    if (xcd < 0) xcd = pa_getenv("PA_CONV3_NOXCD") ? 0 : 1;          // +0.3 % on the step (halo re-reads served by the XCD's own L2)
    PaConvArgs b = a;
    static int dbg = -1;
    if (dbg < 0) { const char* e = pa_getenv("PA_CONV3_DBG"); dbg = e ? atoi(e) : 0; }      // tuning builds: phase ablation / tap rotation
    b.dbg = dbg;
    b.xcd = (a.xcd & 2) | ((xcd && !small && tiles % 8 == 0) ? 1 : 0);
    static int sps = -1;
    if (sps < 0) { const char* e = pa_getenv("PA_CONV3_SPS"); sps = e ? atoi(e) : 1; }      // 0: one slice per step everywhere (round 1)
    // three 16 x 8 tiles per 512-thread workgroup (one shared weight ring) when that is exactly one round of workgroups on the chip:
    // the 64 x 64 maps of the hourglass at batch 24 (768 tiles = 256 workgroups)
#ifdef PA_CONV3_TRI_BUILD          // (measured a wash, DESIGN.md round 4: compiled only on request -- PA_EXTRA=-DPA_CONV3_TRI_BUILD)
    static int tri = -1, cus = 0;
    if (tri < 0) {
        const char* e = pa_getenv("PA_CONV3_TRI"); tri = e ? atoi(e) : 0;
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 0;
    }
    const int wg3 = tiles128 / 3;
    if (tri && a.fin.rows <= 0 && !small && !half && a.Cin == 128 && bigN && tiles128 % 3 == 0 && wg3 % 8 == 0 && wg3 <= cus && 2 * wg3 > cus &&
        (a.ep.mode != PA_OUT_BWD || pa_bwd_epilogue_lds_ok(a))) {
        if (stat_rows) *stat_rows = wg3;
        if (a.ep.rows_out) *a.ep.rows_out = wg3;
        b.xcd = (a.xcd & 2) | (xcd ? 1 : 0);
        dim3 g3(wg3, a.Cout / 128);
        switch (a.in.mode) {
            case PA_LD_PLAIN: hipLaunchKernelGGL((conv3x3_tile_kernel<128, 128, PA_LD_PLAIN, 16, 8, 1, true, 512>), g3, dim3(512), 0, st, b); break;
            case PA_LD_BNRELU: hipLaunchKernelGGL((conv3x3_tile_kernel<128, 128, PA_LD_BNRELU, 16, 8, 1, true, 512>), g3, dim3(512), 0, st, b); break;
            default: hipLaunchKernelGGL((conv3x3_tile_kernel<128, 128, PA_LD_LIN2, 16, 8, 1, true, 512>), g3, dim3(512), 0, st, b); break;
        }
        return (int)hipGetLastError();
    }
#endif
    if (half) { if (sps) launch_tile_shape<16, 4, 2>(b, grid, bigN, st); else launch_tile_shape<16, 4, 1>(b, grid, bigN, st); }
    else if (!small) launch_tile_shape<16, 8, 1>(b, grid, bigN, st);
    else if (st_ == 8) { if (sps) launch_tile_shape<8, 8, 4>(b, grid, bigN, st); else launch_tile_shape<8, 8, 1>(b, grid, bigN, st); }
    else { if (sps) launch_tile_shape<4, 4, 4>(b, grid, bigN, st); else launch_tile_shape<4, 4, 1>(b, grid, bigN, st); }
    return (int)hipGetLastError();
}
