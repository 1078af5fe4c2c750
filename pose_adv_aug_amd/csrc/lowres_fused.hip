// The low-resolution sub-hourglass in ONE persistent launch (forward pass).
//
// Below 32 x 32 the hourglass (reference models/asn_stacked_hg.py:139-157 down path, :192-203 up path; blocks :30-49) is a
// chain of ~45 dependent launches per stack -- 16 x 16, 8 x 8 and 4 x 4 maps, 6 ... 96 workgroups each -- that leaves the
// chip almost empty for 0.45 ms per stack and direction (rocprofv3 trace): every kernel is a few microseconds of
// launch, staging and epilogue latency around almost no work.  Here one workgroup owns one IMAGE for the whole stretch
//
//     pool -> down[1] -> skip[2] -> pool -> down[2] -> skip[3] -> pool -> down[3] -> neck -> up[3] -> up+add -> up[2] -> up+add -> up[1]
//
// (9 residual blocks = 27 convolutions, 3 max pools, 2 upsample-adds).  Convolution, pooling and upsampling never cross an
// image, so all activations stay with their workgroup (written to HBM once, because the backward pass reads them, and read
// back through the CU's own L1 / L2); ONLY the training-mode BatchNorm statistics couple the images: after each convolution
// every workgroup publishes its per-channel partial sums (device-scope write-through stores), meets the others at a counter
// barrier (one device-scope atomic per workgroup, relaxed polling by one lane), sums the G partial rows in a fixed order --
// every workgroup gets the same bits, run to run -- and finalizes scale / shift itself.  27 barriers of ~3 us replace ~45
// launches + 27 finalize launches.
//
// A convolution is an implicit GEMM on v_mfma_f32_16x16x32 (A = 16 output channels x 32 k of the weights, straight from
// global memory / L2 with a one-step register prefetch: no weight ring, no barrier in the K loop; B = 32 k x 16 pixels of the
// image staged ONCE in LDS with the pending BatchNorm + ReLU applied, 16-byte slots XOR-swizzled by the row).  8 waves; a
// wave owns blocks of PB pixel fragments x NB channel fragments so that its LDS and weight traffic per MFMA stays below the
// CU's rates.  Same arithmetic and rounding points as the stand-alone kernels (conv + bias + shortcut -> bf16 -> statistics of
// the stored values).
#include "common.h"
#include "kernels.h"

#define LR_THREADS 512
#define LR_WAVES 8

__device__ __forceinline__ void lr_store_sc1(float2* p, float2 v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), *reinterpret_cast<unsigned long long*>(&v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// four device-scope (L2-bypassing) 8-byte loads in flight at once (the builtin atomic loads are issued one by one)
__device__ __forceinline__ void lr_load_sc1_x4(const float2* p0, const float2* p1, const float2* p2, const float2* p3, float2& a, float2& b,
                                               float2& c, float2& d) {
    asm volatile("global_load_dwordx2 %0, %4, off sc1\n\tglobal_load_dwordx2 %1, %5, off sc1\n\tglobal_load_dwordx2 %2, %6, off sc1\n\t"
                 "global_load_dwordx2 %3, %7, off sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(p0), "v"(p1), "v"(p2), "v"(p3) : "memory");
}
__device__ __forceinline__ void lr_load_sc1_x6(const float2* base, size_t stride, float2 (&v)[6]) {
    const float2 *p0 = base, *p1 = base + stride, *p2 = base + 2 * stride, *p3 = base + 3 * stride, *p4 = base + 4 * stride, *p5 = base + 5 * stride;
    asm volatile("global_load_dwordx2 %0, %6, off sc1\n\tglobal_load_dwordx2 %1, %7, off sc1\n\tglobal_load_dwordx2 %2, %8, off sc1\n\t"
                 "global_load_dwordx2 %3, %9, off sc1\n\tglobal_load_dwordx2 %4, %10, off sc1\n\tglobal_load_dwordx2 %5, %11, off sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5])
                 : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(p4), "v"(p5) : "memory");
}
__device__ __forceinline__ void lr_load_sc1_x6x2(const float2* base, size_t stride, float2 (&v)[6], float2 (&w)[6]) {
    const float2* p[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) p[i] = base + i * stride;
    asm volatile("global_load_dwordx2 %0, %12, off sc1\n\tglobal_load_dwordx2 %1, %13, off sc1\n\tglobal_load_dwordx2 %2, %14, off sc1\n\t"
                 "global_load_dwordx2 %3, %15, off sc1\n\tglobal_load_dwordx2 %4, %16, off sc1\n\tglobal_load_dwordx2 %5, %17, off sc1\n\t"
                 "global_load_dwordx2 %6, %18, off sc1\n\tglobal_load_dwordx2 %7, %19, off sc1\n\tglobal_load_dwordx2 %8, %20, off sc1\n\t"
                 "global_load_dwordx2 %9, %21, off sc1\n\tglobal_load_dwordx2 %10, %22, off sc1\n\tglobal_load_dwordx2 %11, %23, off sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3]),
                   "=&v"(w[4]), "=&v"(w[5])
                 : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7]), "v"(p[8]), "v"(p[9]), "v"(p[10]), "v"(p[11])
                 : "memory");
}
__device__ __forceinline__ float2 lr_load_sc1(const float2* p) {
    unsigned long long u = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *reinterpret_cast<float2*>(&u);
}

// sum over the 16 lanes of a DPP row (= the 16 pixels of an MFMA fragment column group); every lane gets the total
__device__ __forceinline__ float lr_row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));      // row_ror:8
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));      // row_ror:4
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));      // row_ror:2
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));      // row_ror:1
    return v;
}

struct LrSmem {
    bf16 act[65536];                        // 128 KB: the staged input image of the current convolution
    float2 cin[256];                        // {k0, k1} of the input operand (k0 == 0 and k1 == 0: plain)
    float2 cad[256];                        // ... of the addend
    float stat[4][256][2];                  // partial statistics of the current convolution per pixel block pb (a channel has <= 4 of them);
                                            // after the publish: collect's row-group partials
    bf16 tile[LR_WAVES][16][40];            // per wave: 16 pixels x 32 channels (80-byte rows: conflict-free), output / addend transposition
    LrOp prog[48];                          // the whole program (a descriptor fetched from global memory per step is a ~1 us scalar round trip)
};

// 16-byte slot of channel chunk `ch` of LDS row r (XOR swizzle inside the row's own slots)
__device__ __forceinline__ int lr_slot(int ch, int r, int CPP) { return ch ^ (r & (CPP < 16 ? CPP - 1 : 15)); }

// ---- per-channel constants of an operand into LDS
__device__ __forceinline__ void lr_load_consts(float2* dst, const float* k0, const float* k1, int C) {
    for (int c = threadIdx.x; c < C; c += LR_THREADS) dst[c] = k0 ? make_float2(k0[c], k1[c]) : make_float2(0.f, 0.f);
}
__device__ __forceinline__ float lr_value(float x, float2 k, bool bn) { return bn ? fmaxf(fmaf(k.x, x, k.y), 0.f) : x; }

// ---- stage one image [P][Cin] (taps == 1) or its zero-padded halo image [(H+2)(W+2)][Cin] (taps == 9) into LDS
__device__ __forceinline__ void lr_stage(LrSmem& sm, const LrOp& op, int b) {
    const int Cin = op.Cin, CPP = Cin >> 3, P = op.H * op.W;
    const bool bn = op.in_k0 != nullptr;
    const bf16* src = op.in + (size_t)b * P * Cin;
    const int W2 = op.W + 2;
    const int rows = op.taps == 9 ? (op.H + 2) * W2 : P;
    const int shift = Cin == 256 ? 5 : (Cin == 128 ? 4 : 3);
    constexpr int U = 8;                               // loads in flight per thread (the image comes back from L2: ~1 us a round trip)
    const int total = rows * CPP;
    float2 kc[8];                                      // a thread always stages the same channel chunk (LR_THREADS % CPP == 0)
    if (bn) {
#pragma unroll
        for (int j = 0; j < 8; ++j) kc[j] = sm.cin[(threadIdx.x & (CPP - 1)) * 8 + j];
    }
    for (int base = 0; base < total; base += LR_THREADS * U) {
        bf16x8 v[U];
        int rr[U], cc[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = base + u * LR_THREADS + threadIdx.x;
            const int r = idx >> shift, ch = idx & (CPP - 1);
            int p = r;
            bool valid = idx < total;
            if (op.taps == 9) {
                const int py = r / W2 - 1, px = r - (py + 1) * W2 - 1;
                valid = valid && py >= 0 && py < op.H && px >= 0 && px < op.W;
                p = py * op.W + px;
            }
            rr[u] = idx < total ? r : -1; cc[u] = ch; ok[u] = valid;
            if (valid) v[u] = *reinterpret_cast<const bf16x8*>(src + (size_t)p * Cin + ch * 8);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (rr[u] < 0) continue;
            bf16x8 o;
            if (ok[u]) {
                if (bn) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] = (bf16)fmaxf(fmaf(kc[j].x, (float)v[u][j], kc[j].y), 0.f);
                } else o = v[u];
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (bf16)0.f;
            }
            *reinterpret_cast<bf16x8*>(sm.act + rr[u] * Cin + (lr_slot(cc[u], rr[u], CPP) << 3)) = o;
        }
    }
}

// ---- one convolution of one image: blocks of PB pixel fragments x NB channel fragments per wave
// Weights come PACKED per MFMA fragment -- wp[channel fragment][K step][lane][8] (weight_prep_kernel) -- so that a wave's A
// operand of a step is ONE contiguous kilobyte; they are requested LR_WD (2 ... 4) steps ahead (an L2 round trip is ~3 K steps of MFMA work).  The B fragments of the next step are read from LDS while the current step's MFMAs issue.
// LR_WD is even: the LDS double buffer is then indexed by d & 1, a compile-time constant
template <int PB, int NB, int LR_WD>
__device__ __forceinline__ void lr_conv_compute(LrSmem& sm, const LrOp& op, int b, long long* timing, long long& t0, int lvl) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int Cin = op.Cin, Cout = op.Cout, P = op.H * op.W, W2 = op.W + 2, CPP = Cin >> 3;
    const int PBK = (P >> 4) / PB, NBK = (Cout >> 4) / NB;
    const int kper = Cin >> 5;                       // K steps (of 32) per tap
    const int ksteps = op.taps * kper;
    const int frow = lane & 15, fq = lane >> 4;
    const bool add_bn = op.add_k0 != nullptr;
    const bf16* addp = op.add ? op.add + (size_t)b * P * Cout : nullptr;
    bf16* outp = op.out + (size_t)b * P * Cout;
    for (int blk = wave; blk < PBK * NBK; blk += LR_WAVES) {
        const int pb = blk % PBK, nb = blk / PBK;
        f32x4 acc[NB][PB];
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
            for (int j = 0; j < PB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        int brow[PB];                                 // LDS row of this lane's pixel (centre tap) per pixel fragment
#pragma unroll
        for (int j = 0; j < PB; ++j) {
            const int p = (pb * PB + j) * 16 + frow;
            if (op.taps == 9) { const int py = p / op.W, px = p - py * op.W; brow[j] = (py + 1) * W2 + px + 1; }
            else brow[j] = p;
        }
        // fragment i of step s: wp + (((nb * NB + i) * ksteps + s) * 64 + lane) * 8
        const bf16* wl = op.w + ((size_t)(nb * NB) * ksteps * 64 + lane) * 8;
        const size_t fstride = (size_t)ksteps * 512;
        bf16x8 fa[LR_WD][NB];
#pragma unroll
        for (int d = 0; d < LR_WD; ++d)
            if (d < ksteps) {
#pragma unroll
                for (int i = 0; i < NB; ++i) fa[d][i] = *reinterpret_cast<const bf16x8*>(wl + i * fstride + (size_t)d * 512);
            }
        auto load_b = [&](int s, bf16x8 (&fb)[PB]) {
            const int tap = s / kper, kk = s - tap * kper;
            int toff = 0;
            if (op.taps == 9) { const int dy = tap / 3, dx = tap - dy * 3; toff = (dy - 1) * W2 + (dx - 1); }
#pragma unroll
            for (int j = 0; j < PB; ++j) {
                const int r = brow[j] + toff;
                fb[j] = *reinterpret_cast<const bf16x8*>(sm.act + r * Cin + (lr_slot(kk * 4 + fq, r, CPP) << 3));
            }
        };
        bf16x8 fb[2][PB];
        load_b(0, fb[0]);
        // steps in groups of LR_WD so that the weight ring is indexed with compile-time constants
        for (int s0 = 0; s0 < ksteps; s0 += LR_WD) {
#pragma unroll
            for (int d = 0; d < LR_WD; ++d) {
                const int s = s0 + d;
                if (s < ksteps) {
                    if (s + 1 < ksteps) load_b(s + 1, fb[(d + 1) & 1]);
#pragma unroll
                    for (int i = 0; i < NB; ++i)
#pragma unroll
                        for (int j = 0; j < PB; ++j) acc[i][j] = PA_MFMA_16x16x32(fa[d][i], fb[d & 1][j], acc[i][j]);
                    if (s + LR_WD < ksteps) {
#pragma unroll
                        for (int i = 0; i < NB; ++i) fa[d][i] = *reinterpret_cast<const bf16x8*>(wl + i * fstride + (size_t)(s + LR_WD) * 512);
                    }
                }
            }
        }
        if (timing && b == 0 && threadIdx.x == 0) { const long long t = clock64(); atomicAdd(reinterpret_cast<unsigned long long*>(timing) + lvl * 8 + 2, (unsigned long long)(t - t0)); t0 = t; }
        // ---- epilogue of the block: lane = 4 consecutive channels of one pixel per fragment.  Every global operand of the
        // block (biases, shortcut addends) is requested before the first use: one L2 round trip instead of one per fragment row
        f32x4 bias[NB];
        bf16x4 av[NB][PB];
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int n = (nb * NB + i) * 16 + fq * 4;
            bias[i] = *reinterpret_cast<const f32x4*>(op.bias + n);
            if (addp) {
#pragma unroll
                for (int j = 0; j < PB; ++j) av[i][j] = *reinterpret_cast<const bf16x4*>(addp + (size_t)((pb * PB + j) * 16 + frow) * Cout + n);
            }
        }
        float s1[NB][4], s2[NB][4];
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) { s1[i][q] = 0.f; s2[i][q] = 0.f; }
        bf16 (*tile)[40] = sm.tile[wave];
#pragma unroll
        for (int j = 0; j < PB; ++j) {
            const int p0 = (pb * PB + j) * 16;
#pragma unroll
            for (int i0 = 0; i0 < NB; i0 += 2) {             // 32 channels (two fragments; one when NB == 1) of 16 pixels at a time
#pragma unroll
                for (int h = 0; h < 2 && i0 + h < NB; ++h) {
                    const int i = i0 + h;
                    const int n = (nb * NB + i) * 16 + fq * 4;
                    float v[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = acc[i][j][q] + bias[i][q];
                    if (addp) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) v[q] += lr_value((float)av[i][j][q], sm.cad[n + q], add_bn);
                    }
                    bf16x4 o;
#pragma unroll
                    for (int q = 0; q < 4; ++q) { o[q] = (bf16)v[q]; const float rv = (float)o[q]; s1[i][q] += rv; s2[i][q] += rv * rv; }
                    *reinterpret_cast<bf16x4*>(&tile[frow][h * 16 + fq * 4]) = o;
                }
                // read back row-wise: lane = (pixel lane / 4, 16-byte chunk lane % 4): 64 contiguous bytes per pixel
                const int n0 = (nb * NB + i0) * 16;
                if (NB >= 2) {
                    const bf16x8 r = *reinterpret_cast<const bf16x8*>(&tile[lane >> 2][(lane & 3) * 8]);
                    *reinterpret_cast<bf16x8*>(outp + (size_t)(p0 + (lane >> 2)) * Cout + n0 + (lane & 3) * 8) = r;
                } else if (lane < 32) {                      // one fragment: 16 pixels x 32 bytes
                    const bf16x8 r = *reinterpret_cast<const bf16x8*>(&tile[lane >> 1][(lane & 1) * 8]);
                    *reinterpret_cast<bf16x8*>(outp + (size_t)(p0 + (lane >> 1)) * Cout + n0 + (lane & 1) * 8) = r;
                }
            }
        }
        if (op.has_bn) {
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int n = (nb * NB + i) * 16 + fq * 4;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float a = lr_row16_sum(s1[i][q]), c = lr_row16_sum(s2[i][q]);
                    if (frow == 0) { sm.stat[pb][n + q][0] = a; sm.stat[pb][n + q][1] = c; }
                }
            }
        }
        if (timing && b == 0 && threadIdx.x == 0) { const long long t = clock64(); atomicAdd(reinterpret_cast<unsigned long long*>(timing) + lvl * 8 + 7, (unsigned long long)(t - t0)); t0 = t; }
    }
}

// ---- BatchNorm of the convolution just computed: publish, barrier, collect, finalize (every workgroup; workgroup 0 also moves
// the running estimates)
__device__ __forceinline__ void lr_bn_sync(LrSmem& sm, const LrOp& op, const LrLaunch& L, int b, unsigned epoch, long long& t0, int lvl);
#define LR_TS(ph) do { if (L.timing && b == 0 && threadIdx.x == 0) { const long long t = clock64(); atomicAdd(reinterpret_cast<unsigned long long*>(L.timing) + lvl * 8 + (ph), (unsigned long long)(t - t0)); t0 = t; } } while (0)
__device__ __forceinline__ void lr_bn_sync(LrSmem& sm, const LrOp& op, const LrLaunch& L, int b, unsigned epoch, long long& t0, int lvl) {
    const int G = gridDim.x, C = op.Cout, tid = threadIdx.x;
    __syncthreads();                                               // the per-wave statistics are complete
    float2* rows = L.rows + (size_t)(epoch & 1u) * G * 256;        // two row sets: a fast workgroup's next publish cannot overtake a slow reader
    // a channel's partial sums: one per pixel block pb = 0 .. PBK-1 (in that order: reproducible)
    const int PF = (op.H * op.W) >> 4, PB = PF >= 4 ? 4 : 1, PBK = PF / PB;
    for (int c = tid; c < C; c += LR_THREADS) {
        float a = 0.f, q = 0.f;
        for (int pb = 0; pb < PBK; ++pb) { a += sm.stat[pb][c][0]; q += sm.stat[pb][c][1]; }
        lr_store_sc1(rows + (size_t)b * 256 + c, make_float2(a, q));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // this workgroup's activations AND its row have left the CU
    __syncthreads();
    LR_TS(3);
    if (tid == 0) {
        __hip_atomic_fetch_add(L.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = epoch * (unsigned)G;
        for (unsigned spin = 0; spin < (1u << 26); ++spin) {       // (bounded: a lost workgroup must not hang the device)
            if (__hip_atomic_load(L.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) break;
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
    LR_TS(4);
    // collect: Q threads per channel, each sums every Q-th row in increasing order; then the Q partials in order
    float2* part = reinterpret_cast<float2*>(&sm.stat[0][0][0]);      // (the per-wave statistics are dead: everybody is past the publish barrier)
    const int Q = LR_THREADS / C < 4 ? LR_THREADS / C : 4;           // 2 (C = 256) or 4 (C <= 128)
    if (tid < Q * C) {
        const int c = tid % C, q = tid / C;
        float a = 0.f, s = 0.f;
        int r = q;
        for (; r + 11 * Q < G; r += 12 * Q) {
            float2 v[6], w[6];
            lr_load_sc1_x6x2(rows + (size_t)r * 256 + c, (size_t)Q * 256, v, w);
#pragma unroll
            for (int u = 0; u < 6; ++u) { a += v[u].x; s += v[u].y; }
#pragma unroll
            for (int u = 0; u < 6; ++u) { a += w[u].x; s += w[u].y; }
        }
        for (; r + 5 * Q < G; r += 6 * Q) {
            float2 v[6];
            lr_load_sc1_x6(rows + (size_t)r * 256 + c, (size_t)Q * 256, v);
#pragma unroll
            for (int u = 0; u < 6; ++u) { a += v[u].x; s += v[u].y; }
        }
        for (; r < G; r += Q) { const float2 v = lr_load_sc1(rows + (size_t)r * 256 + c); a += v.x; s += v.y; }
        part[q * 256 + c] = make_float2(a, s);
    }
    __syncthreads();
    for (int c = tid; c < C; c += LR_THREADS) {
        float S1 = 0.f, S2 = 0.f;
        for (int q = 0; q < Q; ++q) { S1 += part[q * 256 + c].x; S2 += part[q * 256 + c].y; }
        const float cnt = L.batch * (float)(op.H * op.W);
        const float mu = S1 / cnt;
        const float var = fmaxf(S2 / cnt - mu * mu, 0.f);
        const float is = rsqrtf(var + L.eps);
        const float sc = op.bn.gamma[c] * is;
        // every workgroup stores the same bits; its own later reads (next convolution, shortcut) come back through its own CU
        const float sh = op.bn.beta[c] - mu * sc;
        op.bn.scale[c] = sc;
        op.bn.shift[c] = sh;
        sm.cin[c] = make_float2(sc, sh);           // the next step usually consumes exactly this tensor: its constants are in place
        if (b == 0) {
            op.bn.mean[c] = mu;
            op.bn.invstd[c] = is;
            if (L.update_running) {
                const float unb = cnt > 1.f ? var * cnt / (cnt - 1.f) : var;
                op.bn.rmean[c] = (1.f - L.momentum) * op.bn.rmean[c] + L.momentum * mu;
                op.bn.rvar[c] = (1.f - L.momentum) * op.bn.rvar[c] + L.momentum * unb;
            }
        }
    }
    LR_TS(5);
}

// ---- 2 x 2 max pool of value(in) / nearest-upsample(value(low)) + value(skip), one image; all loads of two items in flight
__device__ __forceinline__ void lr_pool(LrSmem& sm, const LrOp& op, int b) {
    const int C = op.Cout, CG = C >> 3, Wo = op.W, Ho = op.H, Wi = 2 * Wo;
    const bool bn = op.in_k0 != nullptr;
    const bf16* src = op.in + (size_t)b * (4 * Ho * Wo) * C;
    bf16* dst = op.out + (size_t)b * (Ho * Wo) * C;
    const int total = Ho * Wo * CG;
    constexpr int U = 2;
    float2 kc[8];                                      // (LR_THREADS % CG == 0: a thread keeps its channel group)
#pragma unroll
    for (int j = 0; j < 8; ++j) kc[j] = sm.cin[(threadIdx.x % CG) * 8 + j];
    for (int base = 0; base < total; base += LR_THREADS * U) {
        bf16x8 v[U][4];
        int pp[U], cc[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = base + u * LR_THREADS + threadIdx.x;
            const int cg = idx % CG, p = idx / CG, yo = p / Wo, xo = p - yo * Wo;
            pp[u] = idx < total ? p : -1; cc[u] = cg * 8;
            if (idx < total) {
#pragma unroll
                for (int k = 0; k < 4; ++k) v[u][k] = *reinterpret_cast<const bf16x8*>(src + ((size_t)(2 * yo + (k >> 1)) * Wi + 2 * xo + (k & 1)) * C + cg * 8);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (pp[u] < 0) continue;
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float m = lr_value((float)v[u][0][j], kc[j], bn);
#pragma unroll
                for (int k = 1; k < 4; ++k) m = fmaxf(m, lr_value((float)v[u][k][j], kc[j], bn));
                o[j] = (bf16)m;
            }
            *reinterpret_cast<bf16x8*>(dst + (size_t)pp[u] * C + cc[u]) = o;
        }
    }
}

__device__ __forceinline__ void lr_upadd(LrSmem& sm, const LrOp& op, int b) {
    const int C = op.Cout, CG = C >> 3, W = op.W, H = op.H, Wl = W / 2;
    const bool bnl = op.in_k0 != nullptr, bns = op.add_k0 != nullptr;
    const bf16* low = op.in + (size_t)b * (H / 2 * Wl) * C;
    const bf16* skip = op.add + (size_t)b * (H * W) * C;
    bf16* dst = op.out + (size_t)b * (H * W) * C;
    const int total = H * W * CG;
    constexpr int U = 4;
    float2 kl[8], ks[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { kl[j] = sm.cin[(threadIdx.x % CG) * 8 + j]; ks[j] = sm.cad[(threadIdx.x % CG) * 8 + j]; }
    for (int base = 0; base < total; base += LR_THREADS * U) {
        bf16x8 l[U], s[U];
        int pp[U], cc[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = base + u * LR_THREADS + threadIdx.x;
            const int cg = idx % CG, p = idx / CG, y = p / W, x = p - y * W;
            pp[u] = idx < total ? p : -1; cc[u] = cg * 8;
            if (idx < total) {
                l[u] = *reinterpret_cast<const bf16x8*>(low + ((size_t)(y >> 1) * Wl + (x >> 1)) * C + cg * 8);
                s[u] = *reinterpret_cast<const bf16x8*>(skip + (size_t)p * C + cg * 8);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (pp[u] < 0) continue;
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (bf16)(lr_value((float)l[u][j], kl[j], bnl) + lr_value((float)s[u][j], ks[j], bns));
            *reinterpret_cast<bf16x8*>(dst + (size_t)pp[u] * C + cc[u]) = o;
        }
    }
}

// L.timing (tuning aid, normally null): workgroup 0 adds the shader-clock cycles of every phase, per map size:
// timing[(level * 8) + phase], level 0 / 1 / 2 = 16 / 8 / 4 pixel maps, phase 0 constants + trailing drain, 1 staging, 2 K loop
// (wave 0), 7 epilogue (wave 0), 3 wait for the other waves + publish + store drain, 4 barrier wait, 5 collect + finalize,
// 6 pool / upsample-add
#define LR_T(ph) do { if (L.timing && b == 0 && threadIdx.x == 0) { const long long t = clock64(); atomicAdd(reinterpret_cast<unsigned long long*>(L.timing) + lvl * 8 + (ph), (unsigned long long)(t - t0)); t0 = t; } } while (0)
__global__ __launch_bounds__(LR_THREADS, 1) void lowres_fwd_kernel(const LrOp* ops, int nops, LrLaunch L) {
    __shared__ __attribute__((aligned(16))) LrSmem sm;
    const int b = blockIdx.x;
    unsigned epoch = 0;
    const float* cin_of = reinterpret_cast<const float*>(~(size_t)0);      // whose {scale, shift} sm.cin holds (null = plain operand)
    long long t0 = L.timing ? clock64() : 0;
    for (int i = threadIdx.x; i < nops * (int)(sizeof(LrOp) / 4); i += LR_THREADS) reinterpret_cast<int*>(sm.prog)[i] = reinterpret_cast<const int*>(ops)[i];
    __syncthreads();
    for (int oi = 0; oi < nops; ++oi) {
        const LrOp& op = sm.prog[oi];
        const int lvl = op.W >= 16 ? 0 : (op.W >= 8 ? 1 : 2);
        // constants of the operands (written by THIS workgroup's own finalize steps, or by earlier launches); the finalize of
        // the previous convolution has left its own in sm.cin already
        if (op.in_k0 != cin_of) { lr_load_consts(sm.cin, op.in_k0, op.in_k1, op.type == LR_CONV ? op.Cin : op.Cout); cin_of = op.in_k0; }
        if (op.add) lr_load_consts(sm.cad, op.add_k0, op.add_k1, op.Cout);
        __syncthreads();
        LR_T(0);
        if (op.type == LR_POOL) { lr_pool(sm, op, b); __syncthreads(); LR_T(6); }
        else if (op.type == LR_UPADD) { lr_upadd(sm, op, b); __syncthreads(); LR_T(6); }
        else {
            lr_stage(sm, op, b);
            __syncthreads();
            LR_T(1);
            const int PF = (op.H * op.W) >> 4;
            if (PF >= 16) lr_conv_compute<4, 4, 2>(sm, op, b, L.timing, t0, lvl);
            else if (PF >= 4) lr_conv_compute<4, 1, 4>(sm, op, b, L.timing, t0, lvl);
            else lr_conv_compute<1, 1, 4>(sm, op, b, L.timing, t0, lvl);
            if (op.has_bn) {
                ++epoch;
                __syncthreads();
                LR_T(3);
                lr_bn_sync(sm, op, L, b, epoch, t0, lvl);
                cin_of = op.bn.scale;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // stores of this step before the next one reads them back
        __syncthreads();
        LR_T(0);
    }
}

int pa_launch_lowres_fwd(const LrOp* ops_dev, int nops, const LrLaunch& L, int B, hipStream_t st) {
    if (B < 1 || B > 256) { pa_set_error_msg("pa_launch_lowres_fwd: one workgroup per image, 1 <= B <= 256"); return 1; }
    hipLaunchKernelGGL(lowres_fwd_kernel, dim3(B), dim3(LR_THREADS), 0, st, ops_dev, nops, L);
    return (int)hipGetLastError();
}
