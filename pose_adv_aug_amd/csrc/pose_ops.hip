// Pose-library kernels (reference pylib/): Gaussian heat maps + MSE, the 16-channel heat-map head
// fused with the loss, arg-max / PCKh, similarity transforms, the scale/rotation bilinear warp and the
// augmentation samplers.  All streaming / tiny: coalesced accesses, wavefront reductions (64 lanes).
#include "common.h"
#include "kernels.h"
#include "pose_ops.h"

// ------------------------------------------------------------------------------------------------
// Gaussian target (reference pylib/HumanPts.py:36-46, 82-116): 7x7 patch exp(-(dx^2+dy^2)/9) whose
// top-left corner is int(pt-3) (C-style truncation toward zero), clipped at the borders; joints with
// x<=0 || y<=0 || x>W || y>H give an all-zero map.  Evaluated per pixel, never materialised on the
// training path.
// float32(exp(-d/9)) for the integer squared distances d = 0..18 of a 7x7 patch (exact table, so the
// map is bit-identical to numpy's float64 exp rounded to float32 by the reference's .float())
__constant__ float pa_gauss_tab[19] = {0x1.0000000000000p+0f, 0x1.ca28620000000p-1f, 0x1.99fa400000000p-1f, 0x1.6edd320000000p-1f,
                       0x1.4848cc0000000p-1f, 0x1.25c3020000000p-1f, 0x1.06de9c0000000p-1f, 0x1.d673ba0000000p-2f,
                       0x1.a4fa9e0000000p-2f, 0x1.78b5640000000p-2f, 0x1.5117f80000000p-2f, 0x1.2da5060000000p-2f,
                       0x1.0dec680000000p-2f, 0x1.e313860000000p-3f, 0x1.b046900000000p-3f, 0x1.82d1360000000p-3f,
                       0x1.5a23a80000000p-3f, 0x1.35bd300000000p-3f, 0x1.152aaa0000000p-3f};
struct GaussPatch { int ulx, uly, brx, bry, valid; };

__device__ __forceinline__ GaussPatch gauss_patch(double px, double py, int H, int W) {
    GaussPatch g;
    g.valid = !(px <= 0.0 || py <= 0.0 || px > (double)W || py > (double)H);
    g.ulx = (int)(px - 3.0);
    g.uly = (int)(py - 3.0);
    g.brx = (int)(px + 3.0);
    g.bry = (int)(py + 3.0);
    if (g.ulx >= W || g.uly >= H || g.brx < 0 || g.bry < 0) g.valid = 0;
    return g;
}

__device__ __forceinline__ float gauss_value(const GaussPatch& g, int x, int y) {
    if (!g.valid) return 0.f;
    int gx = x - g.ulx, gy = y - g.uly;
    // the pasted region is [ul, br] inclusive (:102-115); after truncation toward zero br - ul can be 5
    if ((unsigned)gx > 6u || (unsigned)gy > 6u || x > g.brx || y > g.bry) return 0.f;
    return pa_gauss_tab[(gx - 3) * (gx - 3) + (gy - 3) * (gy - 3)];
}

// materialising variant (parity tests / API): out [B][J][H][W] fp32
__global__ void gaussian_heatmap_kernel(const double* pts, float* out, int B, int J, int H, int W) {
    const size_t total = (size_t)B * J * H * W;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        int x = (int)(e % W);
        size_t r = e / W;
        int y = (int)(r % H);
        size_t bj = r / H;
        GaussPatch g = gauss_patch(pts[bj * 2], pts[bj * 2 + 1], H, W);
        out[e] = gauss_value(g, x, y);
    }
}

int pa_launch_gaussian_heatmap(const double* pts, float* out, int B, int J, int H, int W, hipStream_t st) {
    size_t total = (size_t)B * J * H * W;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(gaussian_heatmap_kernel, dim3(blocks), dim3(256), 0, st, pts, out, B, J, H, W);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// weighted L2 (reference pylib/Criterion.py:12-18; with w==1 the inline loss of stack-hg.py:156-159)
__global__ void weighted_l2_kernel(const float* pred, const float* gt, const float* w, size_t n, float inv_numel, float* loss) {
    float acc = 0.f;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        float d = pred[e] - gt[e];
        acc += d * d * (w ? w[e] : 1.f);
    }
    acc = wave_sum(acc);
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(loss, (part[0] + part[1] + part[2] + part[3]) * inv_numel);
}

int pa_launch_weighted_l2(const float* pred, const float* gt, const float* w, size_t n, float* loss, hipStream_t st) {
    int blocks = (int)((n + 255) / 256);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(weighted_l2_kernel, dim3(blocks), dim3(256), 0, st, pred, gt, w, n, 1.f / (float)n, loss);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Heat-map head fused with the loss: heat[m][j] = sum_k relu(bn(x))[m][k] * w[j][k] + b[j]  (J = 16)
// One wave = 16 pixels per MFMA group; weights are the MFMA A operand (rows = 16 joints), the
// activations (transformed on load) the B operand, so a lane holds 4 joints of one pixel.
__global__ __launch_bounds__(256) void head_fwd_kernel(PaOperand in, const bf16* w16, const float* bias, float* heat, bf16* heat64,
                                                       const double* pts, float* loss, int B, int H, int W, int Cin, float inv_numel) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int WS = Cin + 8;                                  // padded weight row (bf16 elements)
    bf16* ws = reinterpret_cast<bf16*>(smem);                // [16][WS]
    float* ks = reinterpret_cast<float*>(smem + 16 * WS * 2);   // [2][Cin] scale, shift
    for (int i = threadIdx.x; i < 16 * Cin / 8; i += blockDim.x) {
        int row = i / (Cin / 8), ch = i - row * (Cin / 8);
        *reinterpret_cast<bf16x8*>(ws + row * WS + ch * 8) = *reinterpret_cast<const bf16x8*>(w16 + row * Cin + ch * 8);
    }
    for (int i = threadIdx.x; i < Cin; i += blockDim.x) {
        ks[i] = in.mode == PA_LD_BNRELU ? in.k0[i] : 1.f;
        ks[Cin + i] = in.mode == PA_LD_BNRELU ? in.k1[i] : 0.f;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int M = B * H * W, HW = H * W;
    const int groups = (M + 15) / 16;
    const int q = lane >> 4, pl = lane & 15;
    float lsum = 0.f;
    f32x4 bj = *reinterpret_cast<const f32x4*>(bias + q * 4);
    for (int g = blockIdx.x * 4 + wave; g < groups; g += gridDim.x * 4) {
        const int m = g * 16 + pl;
        const bool ok = m < M;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        // all of a lane's 16-byte loads of a batch are issued before the first one is used (8 dependent
        // load -> MFMA round trips per 16 pixels made this kernel run at 0.5 TB/s)
        for (int k0 = 0; k0 < Cin / 32; k0 += 8) {
            bf16x8 raw[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int c = (k0 + u) * 32 + q * 8;
                const size_t idx = (ok && c < Cin) ? (size_t)m * Cin + c : 0;
                raw[u] = *reinterpret_cast<const bf16x8*>(in.p + idx);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int c = (k0 + u) * 32 + q * 8;
                if (c < Cin) {
                    bf16x8 fa;
                    if (in.mode == PA_LD_BNRELU) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) fa[j] = (bf16)fmaxf(fmaf(ks[c + j], (float)raw[u][j], ks[Cin + c + j]), 0.f);
                    } else fa = raw[u];
                    if (!ok) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) fa[j] = (bf16)0.f;
                    }
                    bf16x8 fw = *reinterpret_cast<const bf16x8*>(ws + pl * WS + c);
                    acc = PA_MFMA_16x16x32(fw, fa, acc);
                }
            }
        }
        if (ok) {
            const int b = m / HW, rem = m - b * HW, y = rem / W, x = rem - y * W;
            f32x4 hv;
            bf16x4 hb;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                hv[j] = acc[j] + bj[j];
                hb[j] = (bf16)hv[j];
                if (pts) {
                    const int jj = q * 4 + j;
                    GaussPatch gp = gauss_patch(pts[((size_t)b * 16 + jj) * 2], pts[((size_t)b * 16 + jj) * 2 + 1], H, W);
                    float d = hv[j] - gauss_value(gp, x, y);
                    lsum += d * d;
                }
            }
            *reinterpret_cast<f32x4*>(heat + (size_t)m * 16 + q * 4) = hv;
            if (heat64) *reinterpret_cast<bf16x4*>(heat64 + (size_t)m * 64 + q * 4) = hb;
        }
    }
    if (loss) {
        // one atomic per WORKGROUP (one per wave = 6144 float atomics on a single address: ~60 us serialized in L2)
        __shared__ float wsum[4];
        lsum = wave_sum(lsum);
        if (lane == 0) wsum[wave] = lsum;
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(loss, (wsum[0] + wsum[1] + wsum[2] + wsum[3]) * inv_numel);
    }
}

int pa_launch_head_fwd(const PaOperand& in, const bf16* w16, const float* bias, float* heat, bf16* heat64,
                       const double* pts, float* loss, int B, int H, int W, int Cin, hipStream_t st) {
    if (Cin % 32 != 0) { pa_set_error_msg("pa_launch_head_fwd: Cin must be a multiple of 32"); return 1; }
    const int M = B * H * W;
    int blocks = ((M + 15) / 16 + 3) / 4;
    if (blocks > 768) blocks = 768;
    size_t smem = (size_t)16 * (Cin + 8) * 2 + (size_t)2 * Cin * 4;
    hipLaunchKernelGGL(head_fwd_kernel, dim3(blocks), dim3(256), smem, st, in, w16, bias, heat, heat64, pts, loss, B, H, W, Cin,
                       1.f / ((float)M * 16.f));
    return (int)hipGetLastError();
}

// d loss / d heat (stack-hg.py:156-164): 2 (heat - target) / numel, plus the gradient that arrives
// through in_conv (add64), written as the zero-padded 64-channel bf16 operand of the head's dgrad/wgrad
__global__ void heat_grad_kernel(const float* heat, const double* pts, const bf16* add64, bf16* dheat64, float gscale,
                                 int B, int H, int W) {
    const int M = B * H * W, HW = H * W;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < (size_t)M * 4; t += (size_t)gridDim.x * blockDim.x) {
        const int m = (int)(t >> 2), q = (int)(t & 3);
        const int b = m / HW, rem = m - b * HW, y = rem / W, x = rem - y * W;
        f32x4 hv = *reinterpret_cast<const f32x4*>(heat + (size_t)m * 16 + q * 4);
        bf16x4 o;
        bf16x4 ad;
        if (add64) ad = *reinterpret_cast<const bf16x4*>(add64 + (size_t)m * 64 + q * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int jj = q * 4 + j;
            GaussPatch gp = gauss_patch(pts[((size_t)b * 16 + jj) * 2], pts[((size_t)b * 16 + jj) * 2 + 1], H, W);
            float g = 2.f * (hv[j] - gauss_value(gp, x, y)) * gscale;
            if (add64) g += (float)ad[j];
            o[j] = (bf16)g;
        }
        *reinterpret_cast<bf16x4*>(dheat64 + (size_t)m * 64 + q * 4) = o;
    }
}

int pa_launch_heat_grad(const float* heat, const double* pts, const bf16* add64, bf16* dheat64, float gscale,
                        int B, int H, int W, hipStream_t st) {
    size_t total = (size_t)B * H * W * 4;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(heat_grad_kernel, dim3(blocks), dim3(256), 0, st, heat, pts, add64, dheat64, gscale, B, H, W);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// arg-max (reference pylib/Evaluation.py:6-23) + quarter-pixel refinement and back-projection
// (:169-211, :240-248).  One wavefront per (sample, joint) map; element (b,j,p) lives at
// b*sb + j*sj + p*sp so NCHW (sj=HW, sp=1) and the engine's NHWC16 (sj=1, sp=16) are both served.
// one 256-thread workgroup per (sample, joint) map: 16 loads per thread at 64 x 64 (one wave per map took 21 us for 6 MB:
// 64 dependent rounds per lane), wave shuffles, then the four waves through LDS; first maximum in scan order wins ties
__global__ __launch_bounds__(256) void argmax_kernel(const float* maps, long sb, long sj, long sp, int B, int J, int H, int W,
                              float* preds /*[B][J][2] 1-based, 0 if max<=0*/, float* maxval /*[B][J] or null*/) {
    __shared__ float sv[4];
    __shared__ int si[4];
    const int map = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = map / J, j = map - b * J;
    const float* base = maps + b * sb + j * sj;
    const int HW = H * W;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int p0 = threadIdx.x; p0 < HW; p0 += 8 * 256) {          // eight loads in flight (a 64 x 64 map: two round trips instead of sixteen)
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int p = p0 + u * 256; v[u] = base[(long)(p < HW ? p : p0) * sp]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int p = p0 + u * 256;
            if (p < HW && v[u] > best) { best = v[u]; bi = p; }               // p increases per thread: its first maximum is kept
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        float ov = __shfl_xor(best, o, 64);
        int oi = __shfl_xor(bi, o, 64);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if (lane == 0) { sv[wave] = best; si[wave] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 1; w < 4; ++w)
            if (sv[w] > best || (sv[w] == best && si[w] < bi)) { best = sv[w]; bi = si[w]; }
        float x = (float)(bi % W + 1), y = floorf((float)bi / (float)H) + 1.f;   // the reference divides by size(2)
        if (!(best > 0.f)) { x = 0.f; y = 0.f; }
        preds[(size_t)map * 2] = x;
        preds[(size_t)map * 2 + 1] = y;
        if (maxval) maxval[map] = best;
    }
}

// The engine's own heat maps are [B][H*W][16] (joint-minor): the generic kernel reads them with a 64-byte stride per lane
// (26 us for 6 MB).  Here one workgroup takes one sample, a thread reads whole pixels (four 16-byte loads = 16 joints)
// and keeps the 16 running maxima; shuffle + LDS reduction with the same first-maximum tie rule.
__global__ __launch_bounds__(1024) void argmax_nhwc16_kernel(const float* maps, int H, int W, float* preds, float* maxval) {
    __shared__ float sv[16][16];
    __shared__ int si[16][16];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int HW = H * W;
    const f32x4* base = reinterpret_cast<const f32x4*>(maps + (size_t)b * HW * 16);
    float best[16]; int bi[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) { best[j] = -INFINITY; bi[j] = 0x7fffffff; }
    // four pixels (16 loads of 16 bytes) of a thread in flight at once: the plain loop made one memory round trip per pixel, and a 64 x 64
    // map is exactly four pixels per thread -- the kernel is 24 workgroups of pure latency at the end of every step
    for (int p0 = tid; p0 < HW; p0 += 4 * 1024) {
        f32x4 v[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int p = p0 + u * 1024;
            const size_t pc = p < HW ? (size_t)p : (size_t)p0;           // clamped, unconditional loads
#pragma unroll
            for (int q = 0; q < 4; ++q) v[u][q] = base[pc * 4 + q];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int p = p0 + u * 1024;
            if (p < HW) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int j = q * 4 + r;
                        if (v[u][q][r] > best[j]) { best[j] = v[u][q][r]; bi[j] = p; }       // p increases: the first maximum is kept
                    }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            float ov = __shfl_xor(best[j], o, 64);
            int oi = __shfl_xor(bi[j], o, 64);
            if (ov > best[j] || (ov == best[j] && oi < bi[j])) { best[j] = ov; bi[j] = oi; }
        }
        if (lane == 0) { sv[wave][j] = best[j]; si[wave][j] = bi[j]; }
    }
    __syncthreads();
    if (tid < 16) {
        float bv = sv[0][tid]; int bx = si[0][tid];
#pragma unroll
        for (int w = 1; w < 16; ++w) {
            if (sv[w][tid] > bv || (sv[w][tid] == bv && si[w][tid] < bx)) { bv = sv[w][tid]; bx = si[w][tid]; }
        }
        float x = (float)(bx % W + 1), y = floorf((float)bx / (float)H) + 1.f;   // the reference divides by size(2)
        if (!(bv > 0.f)) { x = 0.f; y = 0.f; }
        preds[((size_t)b * 16 + tid) * 2] = x;
        preds[((size_t)b * 16 + tid) * 2 + 1] = y;
        if (maxval) maxval[(size_t)b * 16 + tid] = bv;
    }
}

int pa_launch_argmax(const float* maps, long sb, long sj, long sp, int B, int J, int H, int W, float* preds, float* maxval,
                     hipStream_t st) {
    if (J == 16 && sj == 1 && sp == 16 && sb == (long)H * W * 16 && (reinterpret_cast<uintptr_t>(maps) & 15) == 0) {
        hipLaunchKernelGGL(argmax_nhwc16_kernel, dim3(B), dim3(1024), 0, st, maps, H, W, preds, maxval);
        return (int)hipGetLastError();
    }
    hipLaunchKernelGGL(argmax_kernel, dim3(B * J), dim3(256), 0, st, maps, sb, sj, sp, B, J, H, W, preds, maxval);
    return (int)hipGetLastError();
}

// similarity transform of reference pylib/HumanAug.py:10-35 in double precision
__device__ __forceinline__ void make_transform(double cx, double cy, double scale, double rot, double res, double size, double (&t)[6]) {
    const double h = size * scale;
    double a = res / h, tx = res * (-cx / h + 0.5), ty = res * (-cy / h + 0.5);
    t[0] = a; t[1] = 0.0; t[2] = tx; t[3] = 0.0; t[4] = a; t[5] = ty;
    if (rot != 0.0) {
        const double rr = -rot * 3.14159265358979323846 / 180.0;
        const double sn = sin(rr), cs = cos(rr), hr = res / 2;
        // T <- Tinv * R * Tm * T   (rotation about the crop centre)
        double m00 = cs * t[0], m01 = -sn * t[4], m02 = cs * (t[2] - hr) - sn * (t[5] - hr) + hr;
        double m10 = sn * t[0], m11 = cs * t[4], m12 = sn * (t[2] - hr) + cs * (t[5] - hr) + hr;
        t[0] = m00; t[1] = m01; t[2] = m02; t[3] = m10; t[4] = m11; t[5] = m12;
    }
}

__device__ __forceinline__ void invert_affine(const double (&t)[6], double (&i)[6]) {
    const double det = t[0] * t[4] - t[1] * t[3];
    i[0] = t[4] / det; i[1] = -t[1] / det; i[3] = -t[3] / det; i[4] = t[0] / det;
    i[2] = -(i[0] * t[2] + i[1] * t[5]);
    i[5] = -(i[3] * t[2] + i[4] * t[5]);
}

// final_preds: refine + back-project the 1-based arg-max coordinates to original-image pixels
__global__ void final_preds_kernel(const float* maps, long sb, long sj, long sp, const float* coords, const float* center,
                                   const float* scale, const float* rot, int B, int J, int H, int W, float* out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * J) return;
    const int b = t / J, j = t - b * J;
    float cx = coords[(size_t)t * 2], cy = coords[(size_t)t * 2 + 1];
    const int px = (int)floorf(cx), py = (int)floorf(cy);
    const float* hm = maps + b * sb + j * sj;
    if (px > 1 && px < W && py > 1 && py < H) {
        float dx = hm[(long)((py - 1) * W + px) * sp] - hm[(long)((py - 1) * W + px - 2) * sp];
        float dy = hm[(long)(py * W + px - 1) * sp] - hm[(long)((py - 2) * W + px - 1) * sp];
        cx += 0.25f * (dx > 0.f ? 1.f : (dx < 0.f ? -1.f : 0.f));
        cy += 0.25f * (dy > 0.f ? 1.f : (dy < 0.f ? -1.f : 0.f));
    }
    cx += 0.5f; cy += 0.5f;
    double tf[6], ti[6];
    make_transform((double)center[b * 2], (double)center[b * 2 + 1], (double)scale[b], (double)rot[b], (double)W, 200.0, tf);
    invert_affine(tf, ti);
    const double x0 = (double)(cx - 1.f), y0 = (double)(cy - 1.f);      // the reference subtracts in fp32
    const double ox = ti[0] * x0 + ti[1] * y0 + ti[2], oy = ti[3] * x0 + ti[4] * y0 + ti[5];
    out[(size_t)t * 2] = (float)((int)ox + 1);
    out[(size_t)t * 2 + 1] = (float)((int)oy + 1);
}

int pa_launch_final_preds(const float* maps, long sb, long sj, long sp, const float* coords, const float* center, const float* scale,
                          const float* rot, int B, int J, int H, int W, float* out, hipStream_t st) {
    hipLaunchKernelGGL(final_preds_kernel, dim3((B * J + 63) / 64), dim3(64), 0, st, maps, sb, sj, sp, coords, center, scale, rot,
                       B, J, H, W, out);
    return (int)hipGetLastError();
}

// PCK from point sets (reference pylib/Evaluation.py:25-75, 77-97, 99-167; pylib/HumanAcc.py:7-44).
//   dist[j][b] = |pred-gt| / norm[b] if gt.x > boundary && gt.y > boundary else -1
//   acc[0] = mean over the listed joints that have >=1 valid sample of acc[1+i] = #(d<=thr & valid)/#valid
//   person[b] (optional) = the same ratio per sample over the listed joints, additionally requiring
//   vis[b][j] (1-based arg-max of the augmented GT map) > 1 in both coordinates; 0 when nothing is valid
__global__ void pck_kernel(const float* pred, const float* gt, const float* norm, float boundary, const int* idxs, int nidx,
                           float thr, const float* vis, int B, int J, float* acc, float* person, float* dists_out) {
    extern __shared__ float dist[];            // [J][B]
    for (int t = threadIdx.x; t < B * J; t += blockDim.x) {
        const int b = t / J, j = t - b * J;
        const float gx = gt[(size_t)t * 2], gy = gt[(size_t)t * 2 + 1];
        float d = -1.f;
        if (gx > boundary && gy > boundary) {
            const float dx = pred[(size_t)t * 2] - gx, dy = pred[(size_t)t * 2 + 1] - gy;
            d = sqrtf(dx * dx + dy * dy) / norm[b];
        }
        dist[j * B + b] = d;
        if (dists_out) dists_out[j * B + b] = d;
    }
    __syncthreads();
    if (acc) {
        __shared__ float jacc[64];
        for (int i = threadIdx.x; i < nidx; i += blockDim.x) {
            const int j = idxs[i];
            int valid = 0, good = 0;
            for (int b = 0; b < B; ++b) {
                float d = dist[j * B + b];
                if (d != -1.f) { ++valid; if (d <= thr) ++good; }
            }
            float a = valid > 0 ? (float)good / (float)valid : -1.f;
            jacc[i] = a;
            acc[i + 1] = a;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            float s = 0.f; int cnt = 0;
            for (int i = 0; i < nidx; ++i) if (jacc[i] >= 0.f) { s += jacc[i]; ++cnt; }
            acc[0] = cnt ? s / (float)cnt : 0.f;
        }
    }
    if (person) {
        for (int b = threadIdx.x; b < B; b += blockDim.x) {
            int nd = 0, nv = 0, both = 0, good = 0;
            for (int i = 0; i < nidx; ++i) {
                const int j = idxs[i];
                const float d = dist[j * B + b];
                const bool v = vis ? (vis[((size_t)b * J + j) * 2] > 1.f && vis[((size_t)b * J + j) * 2 + 1] > 1.f) : true;
                if (d != -1.f) ++nd;
                if (v) ++nv;
                if (d != -1.f && v) { ++both; if (d <= thr) ++good; }
            }
            person[b] = (nd > 0 && nv > 0 && both > 0) ? (float)good / (float)both : 0.f;
        }
    }
}

int pa_launch_pck(const float* pred, const float* gt, const float* norm, float boundary, const int* idxs, int nidx, float thr,
                  const float* vis, int B, int J, float* acc, float* person, float* dists_out, hipStream_t st) {
    if (nidx > 64) { pa_set_error_msg("pa_launch_pck: at most 64 joints"); return 1; }
    hipLaunchKernelGGL(pck_kernel, dim3(1), dim3(256), (size_t)B * J * sizeof(float), st, pred, gt, norm, boundary, idxs, nidx, thr,
                       vis, B, J, acc, person, dists_out);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// per-sample geometry: forward transform at heat-map resolution and inverse at input resolution
//   params[b] = {cx, cy, scale, rot, flip, gain_r, gain_g, gain_b}  (cx already mirrored when flip)
__global__ void affine_params_kernel(const double* params, int B, int res_in, int res_out, double* t_out, double* tinv_in) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const double* p = params + (size_t)b * 8;
    double t[6], ti[6];
    make_transform(p[0], p[1], p[2], p[3], (double)res_out, 200.0, t);
#pragma unroll
    for (int i = 0; i < 6; ++i) t_out[(size_t)b * 6 + i] = t[i];
    make_transform(p[0], p[1], p[2], p[3], (double)res_in, 200.0, t);
    invert_affine(t, ti);
#pragma unroll
    for (int i = 0; i < 6; ++i) tinv_in[(size_t)b * 6 + i] = ti[i];
}

__global__ void params_csr_kernel(const double* params, int B, float* csr) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const double* p = params + (size_t)b * 8;
    csr[2 * b] = (float)p[0]; csr[2 * b + 1] = (float)p[1];
    csr[2 * B + b] = (float)p[2];
    csr[3 * B + b] = (float)p[3];
}
int pa_launch_params_csr(const double* params, int B, float* csr, hipStream_t st) {
    hipLaunchKernelGGL(params_csr_kernel, dim3((B + 63) / 64), dim3(64), 0, st, params, B, csr);
    return (int)hipGetLastError();
}

int pa_launch_affine_params(const double* params, int B, int res_in, int res_out, double* t_out, double* tinv_in, hipStream_t st) {
    hipLaunchKernelGGL(affine_params_kernel, dim3((B + 63) / 64), dim3(64), 0, st, params, B, res_in, res_out, t_out, tinv_in);
    return (int)hipGetLastError();
}

// joints -> heat-map coordinates (reference pylib/HumanAug.py:45-54 + data/mpii_for_mpii.py:126-147):
// optional mirror (x <- width - x, left/right joints swapped), transform, invalid (x<=0||y<=0) -> 0
// sizes (optional): int [B][2] = width, height of each sample's own frame (frames of a batch are padded to a common size)
__global__ void transform_pts_kernel(const float* pts, const double* params, const double* t, int B, int J, float width,
                                     const int* sizes, double* out, float* pts_img) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * J) return;
    const int b = i / J, j = i - b * J;
    const bool flip = params[(size_t)b * 8 + 4] != 0.0;
    int src = j;
    if (flip && J == 16) {
        // left/right pairs of pylib/HumanAug.py:241-244: (0,5) (1,4) (2,3) (10,15) (11,14) (12,13)
        src = (j < 6) ? 5 - j : (j >= 10 ? 25 - j : j);
    }
    const float x = pts[((size_t)b * J + src) * 2], y = pts[((size_t)b * J + src) * 2 + 1];
    if (sizes) width = (float)sizes[2 * b];
    const float fxf = flip ? width - x : x;          // the reference mirrors every joint, valid or not
    double ox = 0.0, oy = 0.0;
    if (!(fxf <= 0.f || y <= 0.f)) {
        const double* tt = t + (size_t)b * 6;
        ox = tt[0] * (double)fxf + tt[1] * (double)y + tt[2];
        oy = tt[3] * (double)fxf + tt[4] * (double)y + tt[5];
    }
    out[(size_t)i * 2] = ox;
    out[(size_t)i * 2 + 1] = oy;
    if (pts_img) { pts_img[(size_t)i * 2] = fxf; pts_img[(size_t)i * 2 + 1] = y; }
}

int pa_launch_transform_pts(const float* pts, const double* params, const double* t, int B, int J, float width, const int* sizes,
                            double* out, float* pts_img, hipStream_t st) {
    hipLaunchKernelGGL(transform_pts_kernel, dim3((B * J + 63) / 64), dim3(64), 0, st, pts, params, t, B, J, width, sizes, out, pts_img);
    return (int)hipGetLastError();
}

// scale/rotation warp + crop as ONE inverse-affine bilinear gather (replaces the host PIL path of
// reference pylib/HumanAug.py:117-176): out[b][v][u][c] = clamp(gain_c * bilinear(src_b, Tinv_b (u,v))),
// zero outside the frame, optional mirror of the source frame, output NHWC with 4 channels (4th = 0).
// src: uint8 [B][Hs][Ws][3].  out4: bf16 [B][res][res][4]; outf (optional): fp32 NCHW [B][3][res][res].
// sizes (optional): int [B][2] = width, height of each sample's own frame inside the padded [Hs][Ws] buffer
__global__ void warp_kernel(const unsigned char* src, int Hs, int Ws, const int* sizes, const double* tinv, const double* params, int B,
                            int res, bf16* out4, float* outf) {
    const size_t total = (size_t)B * res * res;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const int u = (int)(t % res);
        const size_t r = t / res;
        const int v = (int)(r % res), b = (int)(r / res);
        const double* ti = tinv + (size_t)b * 6;
        const double* p = params + (size_t)b * 8;
        const bool flip = p[4] != 0.0;
        const int wb = sizes ? sizes[2 * b] : Ws, hb = sizes ? sizes[2 * b + 1] : Hs;
        const double sx = ti[0] * u + ti[1] * v + ti[2], sy = ti[3] * u + ti[4] * v + ti[5];
        const double fx0 = floor(sx), fy0 = floor(sy);
        const int x0 = (int)fx0, y0 = (int)fy0;
        const double ax = sx - fx0, ay = sy - fy0;
        double acc[3] = {0.0, 0.0, 0.0};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int xx = x0 + (k & 1), yy = y0 + (k >> 1);
            const double w = ((k & 1) ? ax : 1.0 - ax) * ((k >> 1) ? ay : 1.0 - ay);
            if ((unsigned)xx < (unsigned)wb && (unsigned)yy < (unsigned)hb) {
                const int xs = flip ? wb - 1 - xx : xx;
                const unsigned char* px = src + (((size_t)b * Hs + yy) * Ws + xs) * 3;
                acc[0] += w * ((double)px[0] / 255.0);
                acc[1] += w * ((double)px[1] / 255.0);
                acc[2] += w * ((double)px[2] / 255.0);
            }
        }
        float o[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) o[c] = (float)fmin(fmax(acc[c] * p[5 + c], 0.0), 1.0);
        if (out4) {
            bf16x4 ob = {(bf16)o[0], (bf16)o[1], (bf16)o[2], (bf16)0.f};
            *reinterpret_cast<bf16x4*>(out4 + t * 4) = ob;
        }
        if (outf) {
            const size_t hw = (size_t)res * res, pix = (size_t)v * res + u;
            outf[((size_t)b * 3 + 0) * hw + pix] = o[0];
            outf[((size_t)b * 3 + 1) * hw + pix] = o[1];
            outf[((size_t)b * 3 + 2) * hw + pix] = o[2];
        }
    }
}

int pa_launch_warp(const unsigned char* src, int Hs, int Ws, const int* sizes, const double* tinv, const double* params, int B, int res,
                   bf16* out4, float* outf, hipStream_t st) {
    size_t total = (size_t)B * res * res;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(warp_kernel, dim3(blocks), dim3(256), 0, st, src, Hs, Ws, sizes, tinv, params, B, res, out4, outf);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Samplers.  Counter-based (stateless) generator: splitmix64 of (seed, step, sample, slot) -> uniform
// in (0,1); normals by Box-Muller.  The LAWS are the reference's (data/mpii_for_mpii.py:12-13,119-135;
// data/joint_train_s_r_agent.py:15-16,33-36,134-139); the stream is this engine's own (the reference
// never seeds its RNG).
__host__ __device__ inline unsigned long long pa_mix64(unsigned long long z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__device__ __forceinline__ double pa_uniform(unsigned long long seed, unsigned long long step, unsigned sample, unsigned slot) {
    unsigned long long k = pa_mix64(seed ^ pa_mix64(step * 0x100000001B3ull + ((unsigned long long)sample << 8) + slot));
    return ((double)(k >> 11) + 0.5) * (1.0 / 9007199254740992.0);
}

__device__ __forceinline__ double pa_normal(unsigned long long seed, unsigned long long step, unsigned sample, unsigned slot) {
    const double u1 = pa_uniform(seed, step, sample, slot), u2 = pa_uniform(seed, step, sample, slot + 1);
    return sqrt(-2.0 * log(u1)) * cos(6.283185307179586476925 * u2);
}

// mode 0: regular law; mode 1: agent law with bins (scale_idx, rot_idx); mode 2/3: agent law, scale only / rotation only
// meta[b] = {objpos_x, objpos_y, scale (already MPII-normalised), frame_width}
// draws (optional, [B][7] float64 = {N(0,1) for the scale, N(0,1) for the rotation, U for "rotation forced to 0", U for the
// flip, U x 3 for the colour gains}) replaces the engine's own counter-based stream: with the draws of a numpy generator the
// kernel reproduces the reference's laws value for value (parity tests).
__global__ void sample_aug_kernel(const float* meta, const int* scale_idx, const int* rot_idx, int mode, unsigned long long seed,
                                  unsigned long long step, const double* draws, int B, double* params) {
#pragma clang fp contract(off)        // numpy evaluates mu + z * var and low + (high - low) * u with two roundings each
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float* m = meta + (size_t)b * 4;
    const double* d = draws ? draws + (size_t)b * 7 : nullptr;
    const double zs = d ? d[0] : pa_normal(seed, step, b, 0), zr = d ? d[1] : pa_normal(seed, step, b, 2);
    float s = m[2];                       // torch.FloatTensor in the reference: s * (2 ** x) is an fp32 product
    double r = 0.0;
    double cx = (double)m[0];
    double flip = 0.0, g0 = 1.0, g1 = 1.0, g2 = 1.0;
    if (mode == 0) {
        // data/mpii_for_mpii.py:12-13,119-123: s *= 2 ** clip(N(0, .25), +-.5); r = clip(N(0, 30), +-60), 0 with probability .6
        s = s * (float)exp2(fmax(-0.5, fmin(0.5, zs * 0.25)));
        r = fmax(-60.0, fmin(60.0, zr * 30.0));
        if ((d ? d[2] : pa_uniform(seed, step, b, 4)) <= 0.6) r = 0.0;
    } else {
        // data/joint_train_s_r_agent.py:15-16,33-36,134-139: bin means arange(-.6, .61, .2) / arange(-60, 61, 20)
        if (mode == 1 || mode == 2) {
            const double mu = -0.6 + (double)scale_idx[b] * 0.2;
            const double f = fmax(mu - 0.05 + 1e-3, fmin(mu + 0.05, mu + zs * 0.05));
            s = s * (float)exp2(f);
        }
        if (mode == 1 || mode == 3) {
            const double mu = -60.0 + 20.0 * (double)rot_idx[b];
            r = fmax(mu - 5.0 + 1e-3, fmin(mu + 5.0, mu + zr * 5.0));
        }
    }
    if (mode == 0 || mode == 1) {
        // data/mpii_for_mpii.py:126-135: flip with probability .5 (c.x <- W - c.x), per-channel gain U(.6, 1.4)
        if ((d ? d[3] : pa_uniform(seed, step, b, 5)) <= 0.5) { flip = 1.0; cx = (double)m[3] - cx; }
        g0 = 0.6 + (1.4 - 0.6) * (d ? d[4] : pa_uniform(seed, step, b, 6));
        g1 = 0.6 + (1.4 - 0.6) * (d ? d[5] : pa_uniform(seed, step, b, 7));
        g2 = 0.6 + (1.4 - 0.6) * (d ? d[6] : pa_uniform(seed, step, b, 8));
    }
    // centre and scale are fp32 quantities in the reference (torch tensors), the rotation a float64
    double* p = params + (size_t)b * 8;
    p[0] = (double)(float)cx; p[1] = (double)m[1]; p[2] = (double)s; p[3] = r; p[4] = flip; p[5] = g0; p[6] = g1; p[7] = g2;
}

int pa_launch_sample_aug(const float* meta, const int* scale_idx, const int* rot_idx, int mode, unsigned long long seed,
                         unsigned long long step, const double* draws, int B, double* params, hipStream_t st) {
    hipLaunchKernelGGL(sample_aug_kernel, dim3((B + 63) / 64), dim3(64), 0, st, meta, scale_idx, rot_idx, mode, seed, step, draws, B, params);
    return (int)hipGetLastError();
}

// softmax over K <= 64 logits per row and one categorical draw per row (inverse CDF)
// (reference joint-train-pose-s-r-agent.py:252-271: softmax -> np.random.choice(K, p=...))
__global__ void sample_categorical_kernel(const float* logits, int B, int K, unsigned long long seed, unsigned long long step,
                                          unsigned slot, float* probs, int* idx) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float* l = logits + (size_t)b * K;
    float mx = l[0];
    for (int k = 1; k < K; ++k) mx = fmaxf(mx, l[k]);
    float sum = 0.f;
    for (int k = 0; k < K; ++k) sum += expf(l[k] - mx);
    const double u = pa_uniform(seed, step, b, slot);
    double cdf = 0.0;
    int pick = K - 1;
    bool done = false;
    for (int k = 0; k < K; ++k) {
        const float p = expf(l[k] - mx) / sum;
        if (probs) probs[(size_t)b * K + k] = p;
        cdf += (double)p;
        if (!done && u < cdf) { pick = k; done = true; }
    }
    if (idx) idx[b] = pick;
}

int pa_launch_sample_categorical(const float* logits, int B, int K, unsigned long long seed, unsigned long long step, unsigned slot,
                                 float* probs, int* idx, hipStream_t st) {
    hipLaunchKernelGGL(sample_categorical_kernel, dim3((B + 63) / 64), dim3(64), 0, st, logits, B, K, seed, step, slot, probs, idx);
    return (int)hipGetLastError();
}

// _Hourglass._sample_mask (reference models/asn_stacked_hg.py:102-136): softmax over the K = H*W cells of each sample's mask
// logits, `k` DISTINCT cells drawn with those probabilities, mask = 1 except at the drawn cells.  np.random.choice(K, k, p,
// replace=False) draws k, keeps the unique ones and redraws the rest with the found cells zeroed: that is sequential drawing
// without replacement (P(a then b) = p_a p_b / (1 - p_a)), done here by inverse CDF in float64 over the fp32 softmax.
// uniforms (optional, [B][k] in (0,1)) replace the engine's own counter-based stream (parity tests).
__global__ void sample_dropout_masks_kernel(const float* logits, int B, int K, int k, unsigned long long seed, unsigned long long step,
                                            const double* uniforms, float* probs, float* masks, int* indexes) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float* l = logits + (size_t)b * K;
    float p[64];
    float mx = l[0];
    for (int c = 1; c < K; ++c) mx = fmaxf(mx, l[c]);
    float sum = 0.f;
    for (int c = 0; c < K; ++c) { p[c] = expf(l[c] - mx); sum += p[c]; }
    for (int c = 0; c < K; ++c) {
        p[c] = p[c] / sum;
        if (probs) probs[(size_t)b * K + c] = p[c];
        masks[(size_t)b * K + c] = 1.f;
    }
    for (int j = 0; j < k; ++j) {
        double total = 0.0;
        for (int c = 0; c < K; ++c) total += (double)p[c];
        const double u = (uniforms ? uniforms[(size_t)b * k + j] : pa_uniform(seed, step, b, 16 + j)) * total;
        double cdf = 0.0;
        int pick = -1, last = 0;
        for (int c = 0; c < K; ++c) {
            cdf += (double)p[c];
            if (p[c] > 0.f) last = c;
            if (pick < 0 && u < cdf) pick = c;
        }
        if (pick < 0) pick = last;
        while (p[pick] == 0.f && pick > 0) --pick;
        masks[(size_t)b * K + pick] = 0.f;
        if (indexes) indexes[(size_t)b * k + j] = pick;
        p[pick] = 0.f;
    }
}

int pa_launch_sample_dropout_masks(const float* logits, int B, int K, int k, unsigned long long seed, unsigned long long step,
                                   const double* uniforms, float* probs, float* masks, int* indexes, hipStream_t st) {
    hipLaunchKernelGGL(sample_dropout_masks_kernel, dim3((B + 63) / 64), dim3(64), 0, st, logits, B, K, k, seed, step, uniforms, probs,
                       masks, indexes);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// validation with flip test-time augmentation (reference stack-hg.py:222-230)
// mirror the 4-channel-padded NHWC bf16 network input along W (img.numpy()[:, :, :, ::-1])
__global__ void flip_lr_nhwc4_kernel(const bf16x4* src, bf16x4* dst, int rows, int W) {
    const size_t total = (size_t)rows * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / W;
        const int x = (int)(i - r * W);
        dst[i] = src[r * W + (W - 1 - x)];
    }
}

int pa_launch_flip_lr_nhwc4(const bf16* src, bf16* dst, int B, int H, int W, hipStream_t st) {
    const size_t total = (size_t)B * H * W;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(flip_lr_nhwc4_kernel, dim3(blocks), dim3(256), 0, st, reinterpret_cast<const bf16x4*>(src),
                       reinterpret_cast<bf16x4*>(dst), B * H, W);
    return (int)hipGetLastError();
}

// out = (a + shuffle_channels(flip_channels(b))) / 2 over NCHW fp32 heat maps: HumanAug.flip_channels
// (pylib/HumanAug.py:198-210) mirrors W, shuffle_channels_for_horizontal_flipping (:179-196) swaps the MPII
// left/right joints [1,4] [0,5] [12,13] [11,14] [10,15] [2,3]
__constant__ int pa_flip_perm[16] = {5, 4, 3, 2, 1, 0, 6, 7, 8, 9, 15, 14, 13, 12, 11, 10};

__global__ void flip_tta_merge_kernel(const float* a, const float* b, float* out, int B, int H, int W) {
    const size_t total = (size_t)B * 16 * H * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % W);
        size_t r = i / W;
        const int y = (int)(r % H); r /= H;
        const int j = (int)(r % 16);
        const size_t n = r / 16;
        const float fb = b[((n * 16 + pa_flip_perm[j]) * H + y) * W + (W - 1 - x)];
        out[i] = (a[i] + fb) / 2.f;
    }
}

int pa_launch_flip_tta_merge(const float* a, const float* b, float* out, int B, int H, int W, hipStream_t st) {
    const size_t total = (size_t)B * 16 * H * W;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(flip_tta_merge_kernel, dim3(blocks), dim3(256), 0, st, a, b, out, B, H, W);
    return (int)hipGetLastError();
}
