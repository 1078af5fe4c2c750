// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of the stacked-hourglass
// training engine.  Activations are NHWC bf16; statistics, accumulators, master weights fp32.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// The 16-bit STORAGE / MFMA-operand type of this build of the library (accumulation, statistics, master weights are fp32
// in both): bfloat16 (libposeadv_hip.so, the default: BASELINE configs[1-3]) or IEEE half (libposeadv_hip_fp16.so, built
// from the same sources with -DPA_FP16: BASELINE configs[4] "fp16 MFMA 1x1 convs").  The type keeps the name `bf16` in the
// sources; nothing below depends on its bit layout.  fp16 has 5 exponent bits: the backward pass runs on gradients
// multiplied by PA_GRAD_SCALE at their origin (heat-map / agent-logit gradients), every fp32 gradient the library
// returns carries that factor, and the optimizer divides it out (pa_grad_scale(), gscale of pa_rmsprop_step).
#ifdef PA_FP16
typedef _Float16 bf16;
typedef _Float16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 bf16x2 __attribute__((ext_vector_type(2)));
#define PA_MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)
#define PA_GRAD_SCALE 32768.f
#define PA_DTYPE_ID 1
#else
typedef __bf16 bf16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
#define PA_MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
#define PA_GRAD_SCALE 1.f
#define PA_DTYPE_ID 0
#endif
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define PA_WAVE 64

#define PA_CHECK(expr)                                                                   \
    do {                                                                                 \
        hipError_t _e = (expr);                                                          \
        if (_e != hipSuccess) { pa_set_error(#expr, _e, __FILE__, __LINE__); return (int)_e; } \
    } while (0)

// A/B switches (PA_* environment variables selecting between parity-tested code paths) exist only in a
// tuning build (PA_TUNING=1 bash build.sh -> -DPA_TUNING); the release library reads no environment at all.
#include <stdlib.h>
#ifdef PA_TUNING
inline const char* pa_getenv(const char* name) { return getenv(name); }
#else
inline const char* pa_getenv(const char*) { return nullptr; }
#endif
// Cycle stamps inside a kernel (tuning builds; tools/conv3_clocks.py): PA_STAMP_DECL(sym, fn) defines the device array and its C accessor
// in one translation unit, PA_STAMP_AT(sym, on, i) records {s_memtime (shader cycles), wall_clock64 (100 MHz)} of thread 0 of workgroup (0, 0).
#ifdef PA_TUNING
#define PA_STAMP_DECL(sym, fn)                                                                                              \
    __device__ unsigned long long sym[32];                                                                                  \
    extern "C" int fn(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(sym), sizeof(unsigned long long) * 32); }
#define PA_STAMP_AT(sym, on, i)                                                                                             \
    do {                                                                                                                    \
        if ((on) && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { sym[2 * (i)] = __builtin_amdgcn_s_memtime(); sym[2 * (i) + 1] = wall_clock64(); } \
    } while (0)
#else
#define PA_STAMP_DECL(sym, fn)
#define PA_STAMP_AT(sym, on, i) do { } while (0)
#endif

// wave priority of the main chain's convolution kernels (s_setprio: instruction-issue arbitration between the waves of a SIMD; the
// weight-gradient group kernels that share CUs with them stay at 0).  Measured round 5, 8 interleaved runs each: 5.857 ms at 0, 5.836 at 2,
// 5.838 at 3 (-DPA_MAIN_PRIO=n in PA_EXTRA; 0 = no instruction)
#ifndef PA_MAIN_PRIO
#define PA_MAIN_PRIO 2
#endif
#define PA_SET_MAIN_PRIO() do { if (PA_MAIN_PRIO > 0 && !(PA_SIDE_LOW && a.low_prio)) __builtin_amdgcn_s_setprio(PA_MAIN_PRIO); } while (0)
#ifndef PA_SIDE_LOW
#define PA_SIDE_LOW 0          // 1: launches marked low_prio (side branches) stay at priority 0
#endif

void pa_set_error(const char* what, hipError_t e, const char* file, int line);
void pa_set_error_msg(const char* msg);

// How an NHWC operand is turned into a value when it is loaded (per element, per channel c):
//   PLAIN : v = p[i]
//   BNRELU: v = max(0, k0[c] * p[i] + k1[c])          (train/eval BatchNorm + ReLU applied on load)
//   LIN2  : v = k0[c] * p[i] + k1[c] * q[i] + k2[c]    (BatchNorm backward applied on load:
//                                                       p = masked upstream grad, q = raw conv output)
enum { PA_LD_NONE = -1, PA_LD_PLAIN = 0, PA_LD_BNRELU = 1, PA_LD_LIN2 = 2 };

struct PaOperand {
    const bf16* p;
    const bf16* q;
    const float* k0;
    const float* k1;
    const float* k2;
    int mode;
};

// What the epilogue of a conv / elementwise backward kernel does with its value v:
//   PLAIN: store bf16(v)
//   STATS: store bf16(v) and accumulate per-channel sum / sum of squares of the stored value
//          (forward BatchNorm statistics fused into the producer)
//   BWD  : v is the gradient w.r.t. a = relu(s*x+t); store dz = v*[s*x+t>0] and accumulate
//          per-channel sum(dz), sum(dz*xhat), xhat=(x-mean)*invstd   (BatchNorm backward
//          reductions fused into the producer of the gradient)
enum { PA_OUT_PLAIN = 0, PA_OUT_STATS = 1, PA_OUT_BWD = 2 };

struct PaEpilogue {
    int mode;
    float* stats;          // partial rows [grid.x][C][2]: every workgroup writes its own row (no atomics)
    const bf16* xref;      // BWD: raw conv output of the tensor this gradient belongs to
    const float* scale;    // BWD: s
    const float* shift;    // BWD: t
    const float* mean;     // BWD
    const float* invstd;   // BWD
    int* rows_out;         // HOST pointer (ignored on the device): the launcher stores the number of partial rows here
};

__device__ __forceinline__ float bf2f(bf16 v) { return (float)v; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <int MODE>
__device__ __forceinline__ void pa_load8(const PaOperand& op, size_t idx, int c, float (&v)[8]) {
    bf16x8 a = *reinterpret_cast<const bf16x8*>(op.p + idx);
    if (MODE == PA_LD_PLAIN) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (float)a[j];
    } else if (MODE == PA_LD_BNRELU) {
        f32x4 s0 = *reinterpret_cast<const f32x4*>(op.k0 + c), s1 = *reinterpret_cast<const f32x4*>(op.k0 + c + 4);
        f32x4 t0 = *reinterpret_cast<const f32x4*>(op.k1 + c), t1 = *reinterpret_cast<const f32x4*>(op.k1 + c + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[j] = fmaxf(fmaf(s0[j], (float)a[j], t0[j]), 0.f);
            v[j + 4] = fmaxf(fmaf(s1[j], (float)a[j + 4], t1[j]), 0.f);
        }
    } else {
        bf16x8 b = *reinterpret_cast<const bf16x8*>(op.q + idx);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = fmaf(op.k0[c + j], (float)a[j], fmaf(op.k1[c + j], (float)b[j], op.k2[c + j]));
    }
}

// runtime-mode 8-wide operand read used by epilogues (16-byte accesses)
__device__ __forceinline__ void pa_read8(const PaOperand& op, size_t idx, int c, float (&v)[8]) {
    if (op.mode == PA_LD_NONE) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.f;
    } else if (op.mode == PA_LD_PLAIN) pa_load8<PA_LD_PLAIN>(op, idx, c, v);
    else if (op.mode == PA_LD_BNRELU) pa_load8<PA_LD_BNRELU>(op, idx, c, v);
    else pa_load8<PA_LD_LIN2>(op, idx, c, v);
}

// the transform of pa_read8 applied to values that were loaded earlier (raw p / q chunks of channel c..c+7)
__device__ __forceinline__ void pa_apply8(const PaOperand& op, const bf16x8& p, const bf16x8& q, int c, float (&v)[8]) {
    if (op.mode == PA_LD_NONE) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.f;
    } else if (op.mode == PA_LD_PLAIN) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (float)p[j];
    } else if (op.mode == PA_LD_BNRELU) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = fmaxf(fmaf(op.k0[c + j], (float)p[j], op.k1[c + j]), 0.f);
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = fmaf(op.k0[c + j], (float)p[j], fmaf(op.k1[c + j], (float)q[j], op.k2[c + j]));
    }
}

// runtime-mode 4-wide operand read used by epilogues (8-byte accesses)
__device__ __forceinline__ void pa_read4(const PaOperand& op, size_t idx, int c, float (&v)[4]) {
    if (op.mode == PA_LD_NONE) { v[0] = v[1] = v[2] = v[3] = 0.f; return; }
    bf16x4 a = *reinterpret_cast<const bf16x4*>(op.p + idx);
    if (op.mode == PA_LD_PLAIN) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (float)a[j];
    } else if (op.mode == PA_LD_BNRELU) {
        f32x4 s = *reinterpret_cast<const f32x4*>(op.k0 + c), t = *reinterpret_cast<const f32x4*>(op.k1 + c);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = fmaxf(fmaf(s[j], (float)a[j], t[j]), 0.f);
    } else {
        bf16x4 b = *reinterpret_cast<const bf16x4*>(op.q + idx);
        f32x4 k0 = *reinterpret_cast<const f32x4*>(op.k0 + c), k1 = *reinterpret_cast<const f32x4*>(op.k1 + c),
              k2 = *reinterpret_cast<const f32x4*>(op.k2 + c);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = fmaf(k0[j], (float)a[j], fmaf(k1[j], (float)b[j], k2[j]));
    }
}
