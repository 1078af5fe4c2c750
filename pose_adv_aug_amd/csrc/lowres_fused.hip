// The low-resolution sub-hourglass in ONE persistent launch (forward pass).
//
// Below 32 x 32 the hourglass (reference models/asn_stacked_hg.py:139-157 down path, :192-203 up path; blocks :30-49) is a
// chain of ~45 dependent launches per stack -- 16 x 16, 8 x 8 and 4 x 4 maps, 6 ... 96 workgroups each -- that leaves the
// chip almost empty for 0.45 ms per stack and direction (rocprofv3 trace): every kernel is a few microseconds of
// launch, staging and epilogue latency around almost no work.  Here one workgroup owns one IMAGE for the whole stretch
//
//     down[1] -> skip[2] -> pool -> down[2] -> skip[3] -> pool -> down[3] -> neck -> up[3] -> up+add -> up[2] -> up+add -> up[1]
//
// (9 residual blocks = 27 convolutions, 2 max pools, 2 upsample-adds).  Convolution, pooling and upsampling never cross an
// image, so all activations stay with their workgroup; ONLY the training-mode BatchNorm statistics couple the images: after
// each convolution every workgroup publishes its per-channel partial sums as tagged 16-byte granules {sum, tag, sum of squares,
// tag} (one device-scope write-through store each; tag = launch * 64 + epoch, two row sets alternate), polls the granules of
// the whole batch with device-scope loads until every tag matches -- the data IS the barrier: no arrival counter, no atomics --
// sums the G partial rows in a fixed order (every workgroup gets the same bits, run to run) and finalizes scale / shift itself.
// The exchange needs every workgroup of the launch RESIDENT at once: the launcher refuses a batch larger than what the device
// can hold (occupancy query x compute units), and a poll that does not complete traps instead of continuing with stale rows.
//
// Inside a residual block the two inner tensors never come back from memory: conv1's raw output x1 stays in LDS (buffer V),
// is normalised IN PLACE once its statistics are final and is conv2's B operand; the same for x2 (buffer U) and conv3.  Both
// are also written to HBM (the backward pass reads them), by a fully coalesced pass that runs WHILE the workgroup waits at the
// barrier.  conv1's 256-channel input is staged from L2 in two 128-channel halves (buffer U), conv3's 256-channel output and
// its shortcut addend go through small per-wave transposition tiles.
//
// A convolution is an implicit GEMM on v_mfma_f32_16x16x32 (A = 16 output channels x 32 k of the weights, PACKED per fragment
// by weight_prep_kernel and read straight from L2 two to four K steps ahead: no weight ring, no barrier in the K loop; B = 32 k
// x 16 pixels from the LDS image, 16-byte slots XOR-swizzled by the pixel; the 3x3 taps read the un-padded image with the border
// masked per lane).  8 waves; a wave owns blocks of PB pixel fragments x NB channel fragments.  Same arithmetic and rounding
// points as the stand-alone kernels (conv + bias + shortcut -> bf16 -> statistics of the stored values).
#include "common.h"
#include "kernels.h"

#define LR_THREADS 512
#define LR_WAVES 8
#define LR_HALF 128                                  // channels of a staged K half

typedef const __attribute__((address_space(1))) bf16x8* g_bf16x8c;
typedef __attribute__((address_space(1))) bf16x8* g_bf16x8;
typedef const __attribute__((address_space(1))) bf16x4* g_bf16x4c;
typedef const __attribute__((address_space(1))) f32x4* g_f32x4c;
typedef const __attribute__((address_space(1))) float* g_f32c;
typedef __attribute__((address_space(1))) float* g_f32;
#define LR_G(T, p) ((T)(p))                          // the descriptors live in LDS: tell the compiler these pointers are global

__device__ __forceinline__ void lr_store_sc1(float2* p, float2 v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), *reinterpret_cast<unsigned long long*>(&v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// device-scope (L2-bypassing) 8-byte loads, 6 / 12 in flight at once (the builtin atomic loads are issued one by one)
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

// ---- the barrier IS the data: a workgroup's partial row is C 16-byte granules {sum, tag, sum of squares, tag} written with ONE
// device-scope (write-through) store each; tag = (launch id, epoch).  A reader polls the granules it needs until both tags of each
// match (8-byte halves arrive whole): no arrival counter, no store drain in front of it, one fabric round trip instead of three.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void lr_store_granule(u32x4* p, u32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void lr_load_granules6(const u32x4* p0, const u32x4* p1, const u32x4* p2, const u32x4* p3, const u32x4* p4, const u32x4* p5,
                                                  u32x4 (&g)[6]) {
    asm volatile("global_load_dwordx4 %0, %6, off sc1\n\tglobal_load_dwordx4 %1, %7, off sc1\n\tglobal_load_dwordx4 %2, %8, off sc1\n\t"
                 "global_load_dwordx4 %3, %9, off sc1\n\tglobal_load_dwordx4 %4, %10, off sc1\n\tglobal_load_dwordx4 %5, %11, off sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(g[0]), "=&v"(g[1]), "=&v"(g[2]), "=&v"(g[3]), "=&v"(g[4]), "=&v"(g[5])
                 : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(p4), "v"(p5) : "memory");
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
    bf16 buf[2][32768];                     // U, V: 64 KB each, an image of <= 256 pixels x <= 128 channels (16-byte slots swizzled by the pixel)
    float2 cin[256];                        // {k0, k1} of the input operand (k0 == 0 and k1 == 0: plain)
    float2 cad[256];                        // ... of the addend
    float stat[4][256][2];                  // partial statistics of the current convolution per pixel block pb (a channel has <= 4 of them);
                                            // after the publish: collect's row-group partials
    bf16 tile[LR_WAVES][16][40];            // per wave: 16 pixels x 32 channels (80-byte rows: conflict-free), output transposition
    LrOp prog[48];                          // the whole program (a descriptor fetched from global memory per step is a ~1 us scalar round trip)
};

struct LrTimer {                            // tuning aid: workgroup 0, thread 0 adds the shader-clock cycles of the phases (kernels.h LrLaunch::timing)
    long long* t; long long t0; int lvl; bool on;
    __device__ __forceinline__ void mark(int phase) {
        if (on) { const long long now = clock64(); atomicAdd(reinterpret_cast<unsigned long long*>(t) + lvl * 8 + phase, (unsigned long long)(now - t0)); t0 = now; }
    }
};

// 16-byte slot of channel chunk `ch` of LDS row r (XOR swizzle inside the row's own slots)
__device__ __forceinline__ int lr_slot(int ch, int r, int CPP) { return ch ^ (r & (CPP < 16 ? CPP - 1 : 15)); }

// ---- per-channel constants of an operand into LDS
__device__ __forceinline__ void lr_load_consts(float2* dst, const float* k0, const float* k1, int C) {
    g_f32c a = LR_G(g_f32c, k0), c = LR_G(g_f32c, k1);
    for (int i = threadIdx.x; i < C; i += LR_THREADS) dst[i] = k0 ? make_float2(a[i], c[i]) : make_float2(0.f, 0.f);
}
__device__ __forceinline__ float lr_value(float x, float2 k, bool bn) { return bn ? fmaxf(fmaf(k.x, x, k.y), 0.f) : x; }

// ---- stage channels [c0, c0 + CW) of one image [P][Cin] from global memory into an LDS buffer [P][CW], BatchNorm + ReLU applied
__device__ __forceinline__ void lr_stage(bf16* dst, const LrSmem& sm, const LrOp& op, int b, int c0, int CW) {
    const int Cin = op.Cin, CPP = CW >> 3, P = op.H * op.W;
    const bool bn = op.in_k0 != nullptr;
    g_bf16x8c src = LR_G(g_bf16x8c, op.in + (size_t)b * P * Cin + c0);
    const int shift = CW == 128 ? 4 : (CW == 64 ? 3 : 2);
    constexpr int U = 8;                               // loads in flight per thread
    const int total = P * CPP;
    float2 kc[8];                                      // a thread always stages the same channel chunk (LR_THREADS % CPP == 0)
    if (bn) {
#pragma unroll
        for (int j = 0; j < 8; ++j) kc[j] = sm.cin[c0 + (threadIdx.x & (CPP - 1)) * 8 + j];
    }
    for (int base = 0; base < total; base += LR_THREADS * U) {
        bf16x8 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = base + u * LR_THREADS + threadIdx.x;
            if (idx < total) v[u] = src[((idx >> shift) * Cin >> 3) + (idx & (CPP - 1))];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = base + u * LR_THREADS + threadIdx.x;
            if (idx >= total) continue;
            const int r = idx >> shift, ch = idx & (CPP - 1);
            bf16x8 o = v[u];
            if (bn) {
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (bf16)fmaxf(fmaf(kc[j].x, (float)v[u][j], kc[j].y), 0.f);
            }
            *reinterpret_cast<bf16x8*>(dst + r * CW + (lr_slot(ch, r, CPP) << 3)) = o;
        }
    }
}

// ---- the raw image an earlier convolution of this block left in an LDS buffer: BatchNorm + ReLU in place
// (a thread keeps ONE channel chunk -- LR_THREADS % CPP == 0 -- so its constants sit in registers)
__device__ __forceinline__ void lr_transform_inplace(bf16* buf, const LrSmem& sm, int P, int C) {
    const int CPP = C >> 3, total = P * CPP;
    const int shift = C == 128 ? 4 : (C == 64 ? 3 : 2);
    const int ch = threadIdx.x & (CPP - 1);
    float2 kc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) kc[j] = sm.cin[ch * 8 + j];
    for (int idx = threadIdx.x; idx < total; idx += LR_THREADS) {
        const int r = idx >> shift;
        bf16x8* p = reinterpret_cast<bf16x8*>(buf + r * C + (lr_slot(ch, r, CPP) << 3));
        const bf16x8 v = *p;
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (bf16)fmaxf(fmaf(kc[j].x, (float)v[j], kc[j].y), 0.f);
        *p = o;
    }
}

// ---- LDS image [P][C] -> global [b][P][C], coalesced (runs while the workgroup waits at the barrier)
__device__ __forceinline__ void lr_writeout(const bf16* buf, bf16* out, int b, int P, int C) {
    const int CPP = C >> 3, total = P * CPP;
    const int shift = C == 128 ? 4 : (C == 64 ? 3 : 2);
    g_bf16x8 dst = LR_G(g_bf16x8, out + (size_t)b * P * C);
    for (int idx = threadIdx.x; idx < total; idx += LR_THREADS) {
        const int r = idx >> shift, ch = idx & (CPP - 1);
        dst[idx] = *reinterpret_cast<const bf16x8*>(buf + r * C + (lr_slot(ch, r, CPP) << 3));
    }
}

// ---- acc += W * image over the taps x KPER K steps [tap * kper_all + k_lo, ... + KPER) of one block.  Weights packed per MFMA
// fragment, wp[channel fragment][K step][lane][8] (weight_prep_kernel), so a wave's A operand of a step is ONE contiguous kilobyte;
// requested WD steps ahead (an L2 round trip is ~3 steps of MFMA work).  The B fragments of the next step are read from LDS while the
// current step's MFMAs issue.  Per tap the border test, the row and the swizzle of a lane's pixel are computed ONCE; a step only
// moves the channel chunk.  KPER (steps of this call per tap) is a multiple of WD, WD is even: every ring index is a constant.
// k_lo: first step of the call within a tap's Cin / 32 (the K halves of conv1); the LDS image holds channels [32 k_lo, 32 (k_lo + KPER)).
template <int PB, int NB, int WD, int KPER>
__device__ __forceinline__ void lr_kloop(f32x4 (&acc)[NB][PB], const bf16* img, const LrOp& op, int pb, int nb, int k_lo) {
    const int lane = threadIdx.x & 63, frow = lane & 15, fq = lane >> 4;
    const int W = op.W, H = op.H;
    constexpr int CW = KPER * 32, CPP = CW >> 3;
    const int kper_all = op.Cin >> 5, ksteps = op.taps * kper_all, taps = op.taps;
    int py[PB], px[PB];
#pragma unroll
    for (int j = 0; j < PB; ++j) {
        const int p = (pb * PB + j) * 16 + frow;
        py[j] = p / W; px[j] = p - py[j] * W;
    }
    g_bf16x8c wl = LR_G(g_bf16x8c, op.w) + ((size_t)(nb * NB) * ksteps * 64 + lane);      // fragment i, step s: + (i * ksteps + s) * 64
    const int fstride = ksteps * 64;
    // global K step of (tap, kk): tap * kper_all + k_lo + kk; this call covers taps * KPER of them, numbered t = tap * KPER + kk
    auto gstep = [&](int t) { const int tap = t / KPER; return tap * kper_all + k_lo + (t - tap * KPER); };
    const int nsteps = taps * KPER;
    bf16x8 fa[WD][NB];
#pragma unroll
    for (int d = 0; d < WD; ++d)
        if (d < nsteps) {
#pragma unroll
            for (int i = 0; i < NB; ++i) fa[d][i] = wl[i * fstride + gstep(d) * 64];
        }
    int rbase[PB], rsw[PB];
    bool ok[PB];
    auto setup = [&](int tap) {
        const int dy = taps == 9 ? tap / 3 - 1 : 0, dx = taps == 9 ? tap % 3 - 1 : 0;
#pragma unroll
        for (int j = 0; j < PB; ++j) {
            const int yy = py[j] + dy, xx = px[j] + dx;
            ok[j] = yy >= 0 && yy < H && xx >= 0 && xx < W;
            const int r = ok[j] ? yy * W + xx : py[j] * W + px[j];
            rbase[j] = r * CW; rsw[j] = r & (CPP < 16 ? CPP - 1 : 15);
        }
    };
    auto load_b = [&](int kk, bf16x8 (&fb)[PB]) {
#pragma unroll
        for (int j = 0; j < PB; ++j) {
            bf16x8 v = *reinterpret_cast<const bf16x8*>(img + rbase[j] + (((kk * 4 + fq) ^ rsw[j]) << 3));
            if (!ok[j]) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (bf16)0.f;
            }
            fb[j] = v;
        }
    };
    bf16x8 fb[2][PB];
    setup(0);
    load_b(0, fb[0]);
    for (int tap = 0; tap < taps; ++tap) {
#pragma unroll
        for (int kk = 0; kk < KPER; ++kk) {
            const int t = tap * KPER + kk;
            if (kk + 1 < KPER) load_b(kk + 1, fb[(kk + 1) & 1]);
            else if (tap + 1 < taps) { setup(tap + 1); load_b(0, fb[0]); }
#pragma unroll
            for (int i = 0; i < NB; ++i)
#pragma unroll
                for (int j = 0; j < PB; ++j) acc[i][j] = PA_MFMA_16x16x32(fa[kk % WD][i], fb[kk & 1][j], acc[i][j]);
            if (t + WD < nsteps) {
                const int g = gstep(t + WD);
#pragma unroll
                for (int i = 0; i < NB; ++i) fa[kk % WD][i] = wl[i * fstride + g * 64];
            }
        }
    }
}

// ---- epilogue of one block: + bias (+ shortcut addend) -> bf16 -> statistics; the values go to an LDS image (dst != null: x1 / x2,
// written to global memory later by lr_writeout) or through the wave's transposition tile straight to global memory (x3)
template <int PB, int NB>
__device__ __forceinline__ void lr_epilogue(LrSmem& sm, f32x4 (&acc)[NB][PB], const LrOp& op, int b, int pb, int nb, bf16* dst) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, frow = lane & 15, fq = lane >> 4;
    const int Cout = op.Cout, P = op.H * op.W, CPP = Cout >> 3;
    const bool add_bn = op.add_k0 != nullptr;
    const bool has_add = op.add != nullptr;
    g_bf16x8 outp = LR_G(g_bf16x8, op.out + (size_t)b * P * Cout);
    // every global operand of the block (biases, shortcut addends) is requested before the first use
    f32x4 bias[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) bias[i] = *LR_G(g_f32x4c, op.bias + (nb * NB + i) * 16 + fq * 4);
    // shortcut addends: one coalesced 16-byte load per lane and 32-channel pair of a pixel fragment (lane = pixel lane / 4, chunk
    // lane % 4), turned into the MFMA layout through the wave's tile; the loads of the next pixel fragment are in flight meanwhile
    constexpr int NP = (NB + 1) / 2;                  // 32-channel pairs per block
    g_bf16x8c add8 = LR_G(g_bf16x8c, op.add + (size_t)b * P * Cout);
    bf16x8 ar[2][NP];
    auto load_add = [&](int j, bf16x8 (&a)[NP]) {
        const int p0 = (pb * PB + j) * 16;
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int n0 = (nb * NB + 2 * k) * 16;
            if (NB >= 2) a[k] = add8[((p0 + (lane >> 2)) * Cout + n0 + (lane & 3) * 8) >> 3];
            else a[k] = add8[((p0 + ((lane & 31) >> 1)) * Cout + n0 + (lane & 1) * 8) >> 3];
        }
    };
    if (has_add) load_add(0, ar[0]);
    float s1[NB][4], s2[NB][4];
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) { s1[i][q] = 0.f; s2[i][q] = 0.f; }
    bf16 (*tile)[40] = sm.tile[wave];
#pragma unroll
    for (int j = 0; j < PB; ++j) {
        const int p0 = (pb * PB + j) * 16;
        if (has_add && j + 1 < PB) load_add(j + 1, ar[(j + 1) & 1]);
#pragma unroll
        for (int i0 = 0; i0 < NB; i0 += 2) {             // 32 channels (two fragments; one when NB == 1) of 16 pixels at a time
            bf16x4 av[2];
            if (has_add) {
                if (NB >= 2) *reinterpret_cast<bf16x8*>(&tile[lane >> 2][(lane & 3) * 8]) = ar[j & 1][i0 >> 1];
                else if (lane < 32) *reinterpret_cast<bf16x8*>(&tile[lane >> 1][(lane & 1) * 8]) = ar[j & 1][0];
#pragma unroll
                for (int h = 0; h < 2 && i0 + h < NB; ++h) av[h] = *reinterpret_cast<const bf16x4*>(&tile[frow][h * 16 + fq * 4]);
            }
#pragma unroll
            for (int h = 0; h < 2 && i0 + h < NB; ++h) {
                const int i = i0 + h;
                const int n = (nb * NB + i) * 16 + fq * 4;
                float v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = acc[i][j][q] + bias[i][q];
                if (has_add) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] += lr_value((float)av[h][q], sm.cad[n + q], add_bn);
                }
                bf16x4 o;
#pragma unroll
                for (int q = 0; q < 4; ++q) { o[q] = (bf16)v[q]; const float rv = (float)o[q]; s1[i][q] += rv; s2[i][q] += rv * rv; }
                if (dst) {
                    const int p = p0 + frow;
                    *reinterpret_cast<bf16x4*>(dst + p * Cout + (lr_slot(n >> 3, p, CPP) << 3) + (n & 4)) = o;
                } else {
                    *reinterpret_cast<bf16x4*>(&tile[frow][h * 16 + fq * 4]) = o;
                }
            }
            if (!dst) {
                // read back row-wise: lane = (pixel lane / 4, 16-byte chunk lane % 4): 64 contiguous bytes per pixel
                const int n0 = (nb * NB + i0) * 16;
                if (NB >= 2) {
                    const bf16x8 r = *reinterpret_cast<const bf16x8*>(&tile[lane >> 2][(lane & 3) * 8]);
                    outp[((p0 + (lane >> 2)) * Cout + n0 + (lane & 3) * 8) >> 3] = r;
                } else if (lane < 32) {                      // one fragment: 16 pixels x 32 bytes
                    const bf16x8 r = *reinterpret_cast<const bf16x8*>(&tile[lane >> 1][(lane & 1) * 8]);
                    outp[((p0 + (lane >> 1)) * Cout + n0 + (lane & 1) * 8) >> 3] = r;
                }
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
}

// ---- one convolution of one image.  Input: op.src_lds >= 0: the raw tensor a previous convolution left in that LDS buffer
// (normalised in place first); else staged from global memory in K halves of <= 128 channels into buffer U.
template <int PB, int NB, int WD, int KPER>
__device__ __forceinline__ void lr_conv(LrSmem& sm, const LrOp& op, int b, LrTimer& tm) {
    const int wave = threadIdx.x >> 6;
    const int Cin = op.Cin, Cout = op.Cout, P = op.H * op.W;
    const int PBK = (P >> 4) / PB, NBK = (Cout >> 4) / NB, nblocks = PBK * NBK;
    bf16* dst = op.dst_lds >= 0 ? sm.buf[op.dst_lds] : nullptr;
    if (op.src_lds >= 0) {
        bf16* img = sm.buf[op.src_lds];
        lr_transform_inplace(img, sm, P, Cin);
        __syncthreads();
        tm.mark(1);
        for (int blk = wave; blk < nblocks; blk += LR_WAVES) {
            const int pb = blk % PBK, nb = blk / PBK;
            f32x4 acc[NB][PB];
#pragma unroll
            for (int i = 0; i < NB; ++i)
#pragma unroll
                for (int j = 0; j < PB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            lr_kloop<PB, NB, WD, KPER>(acc, img, op, pb, nb, 0);
            tm.mark(2);
            lr_epilogue<PB, NB>(sm, acc, op, b, pb, nb, dst);
            tm.mark(3);
        }
    } else {
        // (taps == 1 here; a wave has at most one block when there are two halves: its accumulators live across them)
        constexpr int CW = KPER * 32;                   // channels of a staged K half (128, or the whole 64 ... 128-channel input)
        const int halves = Cin / CW;
        const int blk = wave, pb = blk % PBK, nb = blk / PBK;
        f32x4 acc[NB][PB];
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
            for (int j = 0; j < PB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int h = 0; h < halves; ++h) {
            if (h) __syncthreads();                       // every wave is done with the previous half
            lr_stage(sm.buf[0], sm, op, b, h * CW, CW);
            __syncthreads();
            tm.mark(1);
            if (blk < nblocks) lr_kloop<PB, NB, WD, KPER>(acc, sm.buf[0], op, pb, nb, h * KPER);
            tm.mark(2);
        }
        if (blk < nblocks) lr_epilogue<PB, NB>(sm, acc, op, b, pb, nb, dst);
        tm.mark(3);
        // (more blocks than waves with a single half: the remaining ones)
        if (halves == 1)
            for (int bl = wave + LR_WAVES; bl < nblocks; bl += LR_WAVES) {
                f32x4 ac2[NB][PB];
#pragma unroll
                for (int i = 0; i < NB; ++i)
#pragma unroll
                    for (int j = 0; j < PB; ++j) ac2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                lr_kloop<PB, NB, WD, KPER>(ac2, sm.buf[0], op, bl % PBK, bl / PBK, 0);
                lr_epilogue<PB, NB>(sm, ac2, op, b, bl % PBK, bl / PBK, dst);
            }
    }
}

// ---- BatchNorm of the convolution just computed: publish the partial row (tagged granules), write the LDS image out, poll +
// sum the G rows in a fixed order, finalize (every workgroup; workgroup 0 also moves the running estimates)
__device__ __forceinline__ void lr_bn_sync(LrSmem& sm, const LrOp& op, const LrLaunch& L, int b, unsigned epoch, LrTimer& tm) {
    const int G = gridDim.x, C = op.Cout, tid = threadIdx.x, P = op.H * op.W;
    __syncthreads();                                               // statistics and LDS image complete
    // two row sets: a workgroup publishes epoch e + 2 only after it has read every epoch e + 1 row, i.e. after everybody is done with e
    u32x4* rows = reinterpret_cast<u32x4*>(L.rows) + (size_t)(epoch & 1u) * G * 256;
    const unsigned tag = L.launch_id * 64u + epoch;
    // a channel's partial sums: one per pixel block pb = 0 .. PBK-1 (in that order: reproducible)
    const int PF = P >> 4, PB = PF >= 4 ? 4 : 1, PBK = PF / PB;
    for (int c = tid; c < C; c += LR_THREADS) {
        float a = 0.f, q = 0.f;
        for (int pb = 0; pb < PBK; ++pb) { a += sm.stat[pb][c][0]; q += sm.stat[pb][c][1]; }
        lr_store_granule(rows + (size_t)b * 256 + c, u32x4{__float_as_uint(a), tag, __float_as_uint(q), tag});
    }
    tm.mark(4);
    // the finalize's own operands (gamma, beta; workgroup 0: the running estimates) are requested now, used after the poll
    g_f32c gamma = LR_G(g_f32c, op.bn.gamma), beta = LR_G(g_f32c, op.bn.beta);
    g_f32 rmean = LR_G(g_f32, op.bn.rmean), rvar = LR_G(g_f32, op.bn.rvar);
    float pg = 0.f, pbeta = 0.f, prm = 0.f, prv = 0.f;
    if (tid < C) {
        pg = gamma[tid]; pbeta = beta[tid];
        if (b == 0 && L.update_running) { prm = rmean[tid]; prv = rvar[tid]; }
    }
    if (op.dst_lds >= 0) lr_writeout(sm.buf[op.dst_lds], op.out, b, P, C);      // x1 / x2 -> HBM for the backward pass, while the rows travel
    __syncthreads();                                               // (the statistics area is reused below)
    // collect: Q threads per channel, each polls + sums every Q-th row in increasing order; then the Q partials in order
    float2* part = reinterpret_cast<float2*>(&sm.stat[0][0][0]);
    const int Q = LR_THREADS / C < 4 ? LR_THREADS / C : 4;           // 2 (C = 256) or 4 (C <= 128)
    if (tid < Q * C) {
        const int c = tid % C, q = tid / C;
        float a = 0.f, s = 0.f;
        for (int r0 = q; r0 < G; r0 += 6 * Q) {
            const u32x4* p[6];
            int n = 0;
#pragma unroll
            for (int u = 0; u < 6; ++u) { const int r = r0 + u * Q; p[u] = rows + (size_t)(r < G ? r : r0) * 256 + c; n += r < G; }
            u32x4 g[6];
            bool all = false;
            for (unsigned spin = 0; spin < (1u << 22) && !all; ++spin) {   // (bounded: a lost workgroup must not hang the device)
                lr_load_granules6(p[0], p[1], p[2], p[3], p[4], p[5], g);
                all = true;
#pragma unroll
                for (int u = 0; u < 6; ++u) all = all && g[u][1] == tag && g[u][3] == tag;
                if (!all) __builtin_amdgcn_s_sleep(2);
            }
            // a row that never arrived (a workgroup of the launch is not resident, or died): abort the launch -- the host sees a HIP
            // error at its next synchronisation -- rather than finalize statistics from granules of another epoch
            if (!all) __builtin_trap();
#pragma unroll
            for (int u = 0; u < 6; ++u) if (u < n) { a += __uint_as_float(g[u][0]); s += __uint_as_float(g[u][2]); }
        }
        part[q * 256 + c] = make_float2(a, s);
    }
    __syncthreads();
    tm.mark(5);
    g_f32 scale = LR_G(g_f32, op.bn.scale), shiftp = LR_G(g_f32, op.bn.shift), mean = LR_G(g_f32, op.bn.mean), invstd = LR_G(g_f32, op.bn.invstd);
    if (tid < C) {                                     // (C <= 256 < LR_THREADS: one channel per thread)
        const int c = tid;
        float S1 = 0.f, S2 = 0.f;
        for (int q = 0; q < Q; ++q) { S1 += part[q * 256 + c].x; S2 += part[q * 256 + c].y; }
        const float cnt = L.batch * (float)P;
        const float mu = S1 / cnt;
        const float var = fmaxf(S2 / cnt - mu * mu, 0.f);
        const float is = rsqrtf(var + L.eps);
        const float sc = pg * is;
        const float sh = pbeta - mu * sc;
        // every workgroup stores the same bits; its own later reads (the shortcut of conv3, later blocks) come back through its own CU
        scale[c] = sc;
        shiftp[c] = sh;
        sm.cin[c] = make_float2(sc, sh);           // the next step usually consumes exactly this tensor: its constants are in place
        if (b == 0) {
            mean[c] = mu;
            invstd[c] = is;
            if (L.update_running) {
                const float unb = cnt > 1.f ? var * cnt / (cnt - 1.f) : var;
                rmean[c] = (1.f - L.momentum) * prm + L.momentum * mu;
                rvar[c] = (1.f - L.momentum) * prv + L.momentum * unb;
            }
        }
    }
    tm.mark(6);
}

// ---- 2 x 2 max pool of value(in) / nearest-upsample(value(low)) + value(skip), one image; all loads of a few items in flight
__device__ __forceinline__ void lr_pool(LrSmem& sm, const LrOp& op, int b) {
    const int C = op.Cout, CG = C >> 3, Wo = op.W, Ho = op.H, Wi = 2 * Wo;
    const bool bn = op.in_k0 != nullptr;
    g_bf16x8c src = LR_G(g_bf16x8c, op.in + (size_t)b * (4 * Ho * Wo) * C);
    g_bf16x8 dst = LR_G(g_bf16x8, op.out + (size_t)b * (Ho * Wo) * C);
    const int total = Ho * Wo * CG;
    constexpr int U = 2;
    float2 kc[8];                                      // (LR_THREADS % CG == 0: a thread keeps its channel group)
#pragma unroll
    for (int j = 0; j < 8; ++j) kc[j] = sm.cin[(threadIdx.x % CG) * 8 + j];
    for (int base = 0; base < total; base += LR_THREADS * U) {
        bf16x8 v[U][4];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = base + u * LR_THREADS + threadIdx.x;
            const int cg = idx % CG, p = idx / CG, yo = p / Wo, xo = p - yo * Wo;
            if (idx < total) {
#pragma unroll
                for (int k = 0; k < 4; ++k) v[u][k] = src[((2 * yo + (k >> 1)) * Wi + 2 * xo + (k & 1)) * CG + cg];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = base + u * LR_THREADS + threadIdx.x;
            if (idx >= total) continue;
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float m = lr_value((float)v[u][0][j], kc[j], bn);
#pragma unroll
                for (int k = 1; k < 4; ++k) m = fmaxf(m, lr_value((float)v[u][k][j], kc[j], bn));
                o[j] = (bf16)m;
            }
            dst[idx] = o;
        }
    }
}

__device__ __forceinline__ void lr_upadd(LrSmem& sm, const LrOp& op, int b) {
    const int C = op.Cout, CG = C >> 3, W = op.W, H = op.H, Wl = W / 2;
    const bool bnl = op.in_k0 != nullptr, bns = op.add_k0 != nullptr;
    g_bf16x8c low = LR_G(g_bf16x8c, op.in + (size_t)b * (H / 2 * Wl) * C);
    g_bf16x8c skip = LR_G(g_bf16x8c, op.add + (size_t)b * (H * W) * C);
    g_bf16x8 dst = LR_G(g_bf16x8, op.out + (size_t)b * (H * W) * C);
    const int total = H * W * CG;
    constexpr int U = 4;
    float2 kl[8], ks[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { kl[j] = sm.cin[(threadIdx.x % CG) * 8 + j]; ks[j] = sm.cad[(threadIdx.x % CG) * 8 + j]; }
    for (int base = 0; base < total; base += LR_THREADS * U) {
        bf16x8 l[U], s[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = base + u * LR_THREADS + threadIdx.x;
            const int cg = idx % CG, p = idx / CG, y = p / W, x = p - y * W;
            if (idx < total) { l[u] = low[((y >> 1) * Wl + (x >> 1)) * CG + cg]; s[u] = skip[idx]; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = base + u * LR_THREADS + threadIdx.x;
            if (idx >= total) continue;
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (bf16)(lr_value((float)l[u][j], kl[j], bnl) + lr_value((float)s[u][j], ks[j], bns));
            dst[idx] = o;
        }
    }
}

// L.timing (tuning aid, normally null): workgroup 0 adds the shader-clock cycles of every phase, per map size:
// timing[level * 8 + phase], level 0 / 1 / 2 = 16 / 8 / 4 pixel maps; phase 0 descriptor + constants + trailing drain, 1 staging /
// in-place normalisation, 2 K loop (wave 0), 3 epilogue (wave 0), 4 wait for the other waves + publish, 5 write-out + barrier
// wait, 6 collect + finalize, 7 pool / upsample-add
// KPER = 4: LDS images of 128 channels (chan 256: the K halves of conv1, the 128-channel inner tensors); 2: 64 channels (chan 128)
template <int KPER>
__global__ __launch_bounds__(LR_THREADS, 1) void lowres_fwd_kernel(const LrOp* ops, int nops, LrLaunch L) {
    __shared__ __attribute__((aligned(16))) LrSmem sm;
    const int b = blockIdx.x;
    unsigned epoch = 0;
    const float* cin_of = reinterpret_cast<const float*>(~(size_t)0);      // whose {scale, shift} sm.cin holds (null = plain operand)
    LrTimer tm; tm.t = L.timing; tm.on = L.timing != nullptr && b == 0 && threadIdx.x == 0; tm.t0 = tm.on ? clock64() : 0; tm.lvl = 0;
    for (int i = threadIdx.x; i < nops * (int)(sizeof(LrOp) / 4); i += LR_THREADS) reinterpret_cast<int*>(sm.prog)[i] = reinterpret_cast<const int*>(ops)[i];
    __syncthreads();
    for (int oi = 0; oi < nops; ++oi) {
        // the descriptor as wave-uniform values (scalar registers)
        LrOp op;
        {
            int* d = reinterpret_cast<int*>(&op);
            const int* s = reinterpret_cast<const int*>(&sm.prog[oi]);
#pragma unroll
            for (int i = 0; i < (int)(sizeof(LrOp) / 4); ++i) d[i] = __builtin_amdgcn_readfirstlane(s[i]);
        }
#ifdef LR_TIME_BY_TYPE                                     // (harness experiment: rows = conv1 / conv2 / conv3 of the 16 x 16 level only)
        tm.lvl = op.taps == 9 ? 1 : (op.add ? 2 : 0);
        tm.on = L.timing != nullptr && b == 0 && threadIdx.x == 0 && op.W >= 16 && op.type == LR_CONV;
        if (tm.on) tm.t0 = clock64();
#else
        tm.lvl = op.W >= 16 ? 0 : (op.W >= 8 ? 1 : 2);
#endif
        // constants of the operands (written by THIS workgroup's own finalize steps, or by earlier launches); the finalize of
        // the previous convolution has left its own in sm.cin already
        bool loaded = false;
        if (op.in_k0 != cin_of) { lr_load_consts(sm.cin, op.in_k0, op.in_k1, op.type == LR_CONV ? op.Cin : op.Cout); cin_of = op.in_k0; loaded = true; }
        if (op.add) { lr_load_consts(sm.cad, op.add_k0, op.add_k1, op.Cout); loaded = true; }
        if (loaded) __syncthreads();
        tm.mark(0);
        if (op.type == LR_POOL) { lr_pool(sm, op, b); tm.mark(7); }
        else if (op.type == LR_UPADD) { lr_upadd(sm, op, b); tm.mark(7); }
        else {
            const int PF = (op.H * op.W) >> 4;
            if (PF >= 16) lr_conv<4, 4, 2, KPER>(sm, op, b, tm);
            else if (PF >= 4) lr_conv<4, 1, KPER, KPER>(sm, op, b, tm);
            else lr_conv<1, 1, KPER, KPER>(sm, op, b, tm);
            if (op.has_bn) {
                ++epoch;
                lr_bn_sync(sm, op, L, b, epoch, tm);
                cin_of = op.bn.scale;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // stores of this step before the next one reads them back
        __syncthreads();
    }
}

// workgroups of the fused kernel the current device can hold at once (0: query failed): the in-kernel statistics exchange polls
// rows of EVERY workgroup of the launch, so all of them must be co-resident (one per CU: 145 KB of LDS, 512 threads)
int pa_lowres_max_batch(int chan) {
    static int cached[2] = {-1, -1};
    const int i = chan == 256 ? 0 : 1;
    if (cached[i] >= 0) return cached[i];
    int dev = 0, per_cu = 0, cus = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (e == hipSuccess)
        e = chan == 256 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, lowres_fwd_kernel<4>, LR_THREADS, 0)
                        : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, lowres_fwd_kernel<2>, LR_THREADS, 0);
    cached[i] = e == hipSuccess ? per_cu * cus : 0;
    return cached[i];
}

int pa_launch_lowres_fwd(const LrOp* ops_dev, int nops, const LrLaunch& L, int B, int chan, hipStream_t st) {
    if (chan != 256 && chan != 128) { pa_set_error_msg("pa_launch_lowres_fwd: chan 128 or 256"); return 1; }
    // one workgroup per image and ALL of them resident (the BatchNorm exchange polls every row); 256 = rows of the granule buffer
    const int cap = pa_lowres_max_batch(chan);
    if (B < 1 || B > 256 || B > cap) { pa_set_error_msg("pa_launch_lowres_fwd: batch exceeds the workgroups this device holds at once (or 256)"); return 1; }
    if (nops > 48) { pa_set_error_msg("pa_launch_lowres_fwd: program too long"); return 1; }
    if (chan == 256) hipLaunchKernelGGL(lowres_fwd_kernel<4>, dim3(B), dim3(LR_THREADS), 0, st, ops_dev, nops, L);
    else hipLaunchKernelGGL(lowres_fwd_kernel<2>, dim3(B), dim3(LR_THREADS), 0, st, ops_dev, nops, L);
    return (int)hipGetLastError();
}
