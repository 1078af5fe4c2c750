// Weight gradients of the 1x1 / 3x3 convolutions on bf16 MFMA (gfx950).
//
//   dw[n][tap][c] = sum_m dy[m][n] * x[pixel(m)+tap][c]
//
// The reduction runs over the pixel index m, which is the SLOW index of both NHWC operands, while
// v_mfma_f32_16x16x32_bf16 wants 8 consecutive reduction elements per lane.  Each staging thread
// therefore owns an 8(pixel) x 8(channel) block: it issues 8 coalesced 16-byte loads, applies the
// operand transform in fp32 (BatchNorm backward on dy: PA_LD_LIN2; BatchNorm+ReLU on x:
// PA_LD_BNRELU) and packs the block TRANSPOSED, so the LDS tiles are [channel][64 pixels] and the
// fragment reads are the same swizzled ds_read_b128 pattern as the forward kernel.
// The pixel range is split over `splits` workgroups per output tile; each writes a deterministic
// fp32 partial slab that pa_launch_wgrad_reduce sums into the PyTorch-layout gradient.
#include "common.h"
#include "kernels.h"
#include "wgrad_reduce.h"
#include <stdlib.h>

// STEM: x is the 4-channel-padded image and the 'channels' are the 256 (ky*32+kx*4+c) patch
// elements of the 7x7 stride-2 stem conv (one 16-byte chunk = 2 adjacent input pixels); a.H/a.W
// are the OUTPUT dims.
template <int TN, int TK, int PMODE, int QMODE, int TAPS, bool STEM = false>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(PaWgradArgs a) {
    constexpr int NI = TN / 32, KI = TK / 32;          // fragments per wave (wave tile TN/2 x TK/2)
    // pixels per step: 64, or 128 for the 64x64 tile so that all 256 threads stage an 8x8 block each
    constexpr int MS = (TN + TK <= 128) ? 128 : 64;
    constexpr int MG = MS / 8;                         // 8-pixel groups per step
    constexpr int NP = MG * TN / 8, NQ = MG * TK / 8;  // staging threads for P / Q
    __shared__ __attribute__((aligned(16))) bf16 lds[(TN + TK) * MS];
    bf16* Pt = lds;                 // [TN][MS pixels]
    bf16* Qt = lds + TN * MS;       // [TK][MS pixels]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave & 1, wk = wave >> 1;
    const int M = a.B * a.H * a.W, HW = a.H * a.W;
    const int Kfull = TAPS * a.Cin;
    const int ktiles_per_tap = a.Cin / TK;
    const int split = blockIdx.x;
    const int tap = blockIdx.y / ktiles_per_tap;
    const int cbase = (blockIdx.y - tap * ktiles_per_tap) * TK;     // channel offset inside the tap
    const int nbase = blockIdx.z * TN;
    const int dy = (TAPS == 9) ? tap / 3 - 1 : 0, dx = (TAPS == 9) ? tap - (tap / 3) * 3 - 1 : 0;

    const int steps_total = (M + MS - 1) / MS;
    const int steps_per = (steps_total + a.splits - 1) / a.splits;
    const int step0 = split * steps_per;
    const int step1 = min(step0 + steps_per, steps_total);

    // staging role of this thread: one 8x8 block of P (dy) or of Q (x)
    const bool isP = tid < NP, isQ = !isP && tid < NP + NQ;
    const int bid = isP ? tid : tid - NP;
    const int mg = bid % MG, cg = bid / MG;
    const int chan = isP ? nbase + cg * 8 : cbase + cg * 8;
    float k0[8], k1[8], k2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { k0[j] = 1.f; k1[j] = 0.f; k2[j] = 0.f; }
    if (isP && PMODE == PA_LD_LIN2) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { k0[j] = a.dy.k0[chan + j]; k1[j] = a.dy.k1[chan + j]; k2[j] = a.dy.k2[chan + j]; }
    }
    if (isQ && QMODE == PA_LD_BNRELU) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { k0[j] = a.x.k0[chan + j]; k1[j] = a.x.k1[chan + j]; }
    }

    bf16x8 rp[8], rq[8];
    unsigned okmask = 0;
    float colsum[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) colsum[j] = 0.f;
    const bool want_db = (a.dbpart != nullptr) && blockIdx.y == 0;

    auto gload = [&](int step) {
        okmask = 0;
        const int mfirst = step * MS + mg * 8;
        if (isP) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                int m = mfirst + j;
                if (m < M) {
                    size_t idx = (size_t)m * a.Cout + chan;
                    rp[j] = *reinterpret_cast<const bf16x8*>(a.dy.p + idx);
                    if (PMODE == PA_LD_LIN2) rq[j] = *reinterpret_cast<const bf16x8*>(a.dy.q + idx);
                    okmask |= 1u << j;
                }
            }
        } else if (isQ && STEM) {
            const int cidx = chan >> 3, ky = cidx >> 2, q = cidx & 3;
            const int Hin = 2 * a.H, Win = 2 * a.W;
            int b = mfirst / HW;
            int rem = mfirst - b * HW;
            int y = rem / a.W, x = rem - y * a.W;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                bf16x4 lo = {(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f}, hi = lo;
                const int yi = 2 * y + ky - 3, xi = 2 * x + 2 * q - 3;
                if (mfirst + j < M && ky < 7 && (unsigned)yi < (unsigned)Hin) {
                    const bf16* rowp = a.x.p + ((size_t)b * Hin + yi) * Win * 4;
                    if ((unsigned)xi < (unsigned)Win) lo = *reinterpret_cast<const bf16x4*>(rowp + (size_t)xi * 4);
                    if ((unsigned)(xi + 1) < (unsigned)Win) hi = *reinterpret_cast<const bf16x4*>(rowp + (size_t)(xi + 1) * 4);
                }
                rp[j] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                okmask |= 1u << j;
                if (++x == a.W) { x = 0; if (++y == a.H) { y = 0; ++b; } }
            }
        } else if (isQ) {
            int y = 0, x = 0;
            if (TAPS == 9) { int rem = mfirst % HW; y = rem / a.W; x = rem - y * a.W; }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                int m = mfirst + j;
                bool ok = m < M;
                if (TAPS == 9) ok = ok && (unsigned)(y + dy) < (unsigned)a.H && (unsigned)(x + dx) < (unsigned)a.W;
                if (ok) {
                    size_t idx = (size_t)(m + dy * a.W + dx) * a.Cin + chan;
                    rp[j] = *reinterpret_cast<const bf16x8*>(a.x.p + idx);
                    okmask |= 1u << j;
                }
                if (TAPS == 9) { if (++x == a.W) { x = 0; if (++y == a.H) y = 0; } }
            }
        }
    };

    auto lstore = [&]() {
        if (!(isP || isQ)) return;
        bf16x8 o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {                  // j = pixel inside the block
            float v[8];
            if (okmask & (1u << j)) {
                if (isP) {
                    if (PMODE == PA_LD_LIN2) {
#pragma unroll
                        for (int c = 0; c < 8; ++c) v[c] = fmaf(k0[c], (float)rp[j][c], fmaf(k1[c], (float)rq[j][c], k2[c]));
                    } else {
#pragma unroll
                        for (int c = 0; c < 8; ++c) v[c] = (float)rp[j][c];
                    }
                } else {
                    if (QMODE == PA_LD_BNRELU) {
#pragma unroll
                        for (int c = 0; c < 8; ++c) v[c] = fmaxf(fmaf(k0[c], (float)rp[j][c], k1[c]), 0.f);
                    } else {
#pragma unroll
                        for (int c = 0; c < 8; ++c) v[c] = (float)rp[j][c];
                    }
                }
            } else {
#pragma unroll
                for (int c = 0; c < 8; ++c) v[c] = 0.f;
            }
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                o[c][j] = (bf16)v[c];
                if (want_db && isP) colsum[c] += (float)o[c][j];
            }
        }
        bf16* T = isP ? Pt : Qt;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            int row = cg * 8 + c;
            *reinterpret_cast<bf16x8*>(T + row * MS + ((mg ^ (row & (MG - 1))) << 3)) = o[c];
        }
    };

    f32x4 acc[NI][KI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int ki = 0; ki < KI; ++ki) acc[ni][ki] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fchk = lane >> 4;
    if (step0 < step1) {
        gload(step0);
        lstore();
        __syncthreads();
        for (int step = step0; step < step1; ++step) {
            if (step + 1 < step1) gload(step + 1);
#pragma unroll
            for (int kk = 0; kk < MS / 32; ++kk) {
                bf16x8 fp[NI], fq[KI];
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    int row = wn * (TN / 2) + ni * 16 + frow;
                    fp[ni] = *reinterpret_cast<const bf16x8*>(Pt + row * MS + (((fchk + 4 * kk) ^ (row & (MG - 1))) << 3));
                }
#pragma unroll
                for (int ki = 0; ki < KI; ++ki) {
                    int row = wk * (TK / 2) + ki * 16 + frow;
                    fq[ki] = *reinterpret_cast<const bf16x8*>(Qt + row * MS + (((fchk + 4 * kk) ^ (row & (MG - 1))) << 3));
                }
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int ki = 0; ki < KI; ++ki)
                        acc[ni][ki] = PA_MFMA_16x16x32(fq[ki], fp[ni], acc[ni][ki]);
            }
            __syncthreads();
            if (step + 1 < step1) {
                lstore();
                __syncthreads();
            }
        }
    }

    // partial slab: part[split][n][tap*Cin + c]; lane holds 4 consecutive c for one n
    float* slab = a.part + (size_t)split * a.Cout * Kfull;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int n = nbase + wn * (TN / 2) + ni * 16 + (lane & 15);
#pragma unroll
        for (int ki = 0; ki < KI; ++ki) {
            const int c = cbase + wk * (TK / 2) + ki * 16 + (lane >> 4) * 4;
            *reinterpret_cast<f32x4*>(slab + (size_t)n * Kfull + tap * a.Cin + c) = acc[ni][ki];
        }
    }
    if (want_db && isP) {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float s = colsum[c];
            s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
            if (MG == 16) s += __shfl_xor(s, 8, 64);
            if (mg == 0) a.dbpart[(size_t)split * a.Cout + chan + c] = s;
        }
    }
}

static int round_up8(int v) { return (v + 7) & ~7; }

static void tile_of(int Cin, int Cout, int taps, int& TN, int& TK) {
    TN = (Cout % 128 == 0) ? 128 : 64;
    TK = (Cin % 128 == 0) ? 128 : 64;
    static int t1 = -1, t9 = -1;
    if (t1 < 0) { const char* e = pa_getenv("PA_WGRAD_TILE1"); t1 = e ? atoi(e) : 0; }
    if (t9 < 0) { const char* e = pa_getenv("PA_WGRAD_TILE9"); t9 = e ? atoi(e) : 0; }
    const int f = taps == 1 ? t1 : t9;       // experiments: 1 = 64x64, 2 = 128x64, 3 = 64x128
    if (f == 1) { TN = 64; TK = 64; }
    if (f == 2) { TK = 64; }
    if (f == 3) { TN = 64; }
}

int pa_wgrad_splits(int M, int H, int W, int Cin, int Cout, int taps) {
    if (H > 0 && W > 0 && M % (H * W) == 0) {
        const int ts = pa_wgrad_tile_splits(M / (H * W), H, W, Cin, Cout, taps);
        if (ts > 0) return ts;
    }
    int TN, TK;
    tile_of(Cin, Cout, taps, TN, TK);
    const int tiles = (Cout / TN) * (taps * Cin / TK);
    const int ms = (TN + TK <= 128) ? 128 : 64;
    const int steps_total = (M + ms - 1) / ms;
    // enough workgroups to keep every CU busy with several of them (measured on MI355X: 512 for the
    // 1x1 layers, 1024 for the 3x3 layers; PA_WGRAD_BLOCKS overrides for experiments)
    static int forced = -1;
    if (forced < 0) { const char* e = pa_getenv("PA_WGRAD_BLOCKS"); forced = e ? atoi(e) : 0; }
    const int target = forced >= 8 ? forced : (taps == 9 ? 1024 : 512);
    int s = round_up8((target + tiles - 1) / tiles);
    if (s > steps_total) s = steps_total;
    if (s < 1) s = 1;
    return s;
}

template <int TN, int TK, int PMODE, int QMODE>
static void launch_w_taps(const PaWgradArgs& a, dim3 grid, hipStream_t st) {
    if (a.taps == 1) hipLaunchKernelGGL((conv_wgrad_kernel<TN, TK, PMODE, QMODE, 1>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((conv_wgrad_kernel<TN, TK, PMODE, QMODE, 9>), grid, dim3(256), 0, st, a);
}

template <int TN, int TK>
static void launch_w_modes(const PaWgradArgs& a, dim3 grid, hipStream_t st) {
    const bool lin2 = a.dy.mode == PA_LD_LIN2, bnrelu = a.x.mode == PA_LD_BNRELU;
    if (lin2 && bnrelu) launch_w_taps<TN, TK, PA_LD_LIN2, PA_LD_BNRELU>(a, grid, st);
    else if (lin2) launch_w_taps<TN, TK, PA_LD_LIN2, PA_LD_PLAIN>(a, grid, st);
    else if (bnrelu) launch_w_taps<TN, TK, PA_LD_PLAIN, PA_LD_BNRELU>(a, grid, st);
    else launch_w_taps<TN, TK, PA_LD_PLAIN, PA_LD_PLAIN>(a, grid, st);
}

int pa_launch_stem_wgrad(const PaWgradArgs& a, hipStream_t st) {
    // a.x.p = img4, a.Cin = 256 (virtual patch length), a.Cout = 64, a.H/a.W output dims
    if (a.Cin != 256 || a.Cout != 64 || a.taps != 1 || a.x.mode != PA_LD_PLAIN || a.splits < 1) {
        pa_set_error_msg("pa_launch_stem_wgrad: bad arguments");
        return 1;
    }
    { const int rc = pa_launch_stem_wgrad_tile(a, st); if (rc >= 0) return rc; }
    dim3 grid(a.splits, 2, 1);
    if (a.dy.mode == PA_LD_LIN2)
        hipLaunchKernelGGL((conv_wgrad_kernel<64, 128, PA_LD_LIN2, PA_LD_PLAIN, 1, true>), grid, dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL((conv_wgrad_kernel<64, 128, PA_LD_PLAIN, PA_LD_PLAIN, 1, true>), grid, dim3(256), 0, st, a);
    return (int)hipGetLastError();
}

int pa_launch_wgrad(const PaWgradArgs& a, hipStream_t st) {
    if ((a.taps != 1 && a.taps != 9) || a.Cin % 64 != 0 || a.Cout % 64 != 0 || a.splits < 1) {
        pa_set_error_msg("pa_launch_wgrad: channel counts must be multiples of 64, taps 1 or 9, splits >= 1");
        return 1;
    }
    const int rc = pa_launch_wgrad_tile(a, st);
    if (rc >= 0) return rc;
    int TN, TK;
    tile_of(a.Cin, a.Cout, a.taps, TN, TK);
    dim3 grid(a.splits, a.taps * a.Cin / TK, a.Cout / TN);
    if (TN == 128 && TK == 128) launch_w_modes<128, 128>(a, grid, st);
    else if (TN == 128) launch_w_modes<128, 64>(a, grid, st);
    else if (TK == 128) launch_w_modes<64, 128>(a, grid, st);
    else launch_w_modes<64, 64>(a, grid, st);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// sum the split slabs and scatter into PyTorch layout  dst[n][c][tap]   (one job per conv layer): wgrad_reduce.h
__global__ void wgrad_reduce_kernel(const PaWgradReduceJob* jobs) {
    wgrad_reduce_body(jobs[blockIdx.y], (int)blockIdx.x, (int)gridDim.x);
}

// the same reduction for an arbitrary set of <= PA_RED_LIST_MAX layers (job indices by value): the slabs of two or three consecutive
// flushes of weight gradients in one launch (Net::flush_wgrads)
__global__ void wgrad_reduce_list_kernel(const PaWgradReduceJob* jobs, PaRedList list) {
    wgrad_reduce_body(jobs[list.idx[blockIdx.y]], (int)blockIdx.x, (int)gridDim.x);
}

int pa_launch_wgrad_reduce_list(const PaWgradReduceJob* jobs_dev, const int* idx, int n, int max_elems, hipStream_t st) {
    if (n <= 0) return 0;
    if (n > PA_RED_LIST_MAX) { pa_set_error_msg("pa_launch_wgrad_reduce_list: too many jobs"); return 1; }
    PaRedList list;
    for (int i = 0; i < PA_RED_LIST_MAX; ++i) list.idx[i] = idx[i < n ? i : 0];
    int bx = (max_elems + 255) / 256;
    if (bx > 576) bx = 576;
    if (bx < 1) bx = 1;
    hipLaunchKernelGGL(wgrad_reduce_list_kernel, dim3(bx, n), dim3(256), 0, st, jobs_dev, list);
    return (int)hipGetLastError();
}

int pa_launch_wgrad_reduce(const PaWgradReduceJob* jobs_dev, int njobs, int max_elems, hipStream_t st) {
    if (njobs <= 0) return 0;
    int bx = (max_elems + 255) / 256;
    if (bx > 576) bx = 576;          // the largest layer (3x3, 128x128: 147456 weights) gets one element per thread
    if (bx < 1) bx = 1;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(bx, njobs), dim3(256), 0, st, jobs_dev);
    return (int)hipGetLastError();
}

// one workgroup per output channel n: 1024 threads = 256 patch elements k (coalesced 1 KB rows) x 4 slices of the
// split range, LDS tree over the slices, scatter into dst[n][c][ky][kx]
__global__ __launch_bounds__(1024) void stem_wgrad_reduce_kernel(const float* part, int splits, float* dst, float* zero64) {
    __shared__ float red[4][256];
    const int n = blockIdx.x, k = threadIdx.x & 255, sl = threadIdx.x >> 8;
    if (zero64 && n == 0 && threadIdx.x < 64) zero64[threadIdx.x] = 0.f;          // the stem's bias gradient (a BatchNorm follows: exactly zero) -- was a 6 us memset launch at the very end of the step
    const float* src = part + (size_t)n * 256 + k;
    float s = 0.f;
#pragma unroll 8
    for (int sp = sl; sp < splits; sp += 4) s += src[(size_t)sp * 64 * 256];
    red[sl][k] = s;
    __syncthreads();
    if (sl == 0) {
        const int ky = k >> 5, kx = (k & 31) >> 2, c = k & 3;
        if (ky < 7 && kx < 7 && c < 3) dst[((size_t)(n * 3 + c) * 7 + ky) * 7 + kx] = red[0][k] + red[1][k] + red[2][k] + red[3][k];
    }
}

int pa_launch_stem_wgrad_reduce(const float* part, int splits, float* dst, hipStream_t st, float* zero64) {
    hipLaunchKernelGGL(stem_wgrad_reduce_kernel, dim3(64), dim3(1024), 0, st, part, splits, dst, zero64);
    return (int)hipGetLastError();
}
