// HumanAug.crop on the device (reference pylib/HumanAug.py:117-176 behind data/mpii_for_mpii.py:114-135), stage for stage:
//
//   source byte   = toimage(fp32 frame / 255, mirrored, times gain, clamped)            (utils/imutils.py:31-40, scipy pilutil)
//   [scale_factor = scale * 200 / res >= 2]  whole-frame antialiased downscale by 1 / scale_factor          (:121-133)
//   integer window ul / br from the int-TRUNCATED inverse transform, padded for the rotation, zero outside   (:136-164)
//   [rot != 0]    PIL rotate (bilinear, about the window centre, zero fill), un-pad                           (:166-170)
//   PIL resize of the window to res x res (bilinear = triangle filter widened by the scale)                  (:175)
//
// The PIL stages are restated with Pillow's own arithmetic (Resample.c: double coefficients -> 22-bit fixed point,
// horizontal pass, uint8, vertical pass, uint8; Geometry.c: affine map at pixel centres, 2x2 bilinear in double, truncation),
// so the result equals the reference's crop over Pillow byte for byte (oracle/crop.py is the numpy statement of the same
// arithmetic, pinned to the reference's output in tests/golden/crop.npz).  Only the part of every intermediate image that the
// crop window reads is computed.  scipy's per-image min/max byte stretching is NOT applied: the
// frame is scaled by its full range.
//
// One thread per output pixel of a stage, grid.y = sample; the per-sample geometry comes from a plan kernel.
#include "pose_ops.h"

#pragma clang fp contract(off)       // Pillow's coefficient arithmetic is plain IEEE double: no fused multiply-add

#define CW_PREC 22                   // Resample.c PRECISION_BITS = 32 - 8 - 2

struct CropPlan {
    int wb, hb;            // this sample's frame
    int case_b;            // pre-downscale on
    int wd, hd;            // size of the image the window is cut from (frame, or the downscaled frame)
    int ulx, uly;          // window origin (padded when rotating)
    int nw, nh;            // window size (padded)
    int cw, ch;            // un-padded window = input of the final resize
    int pad, rot_mode;     // rot_mode 0 none, 1 general, 2 = 180 deg, 3 = 90 deg, 4 = 270 deg (square window only)
    int dx0, dx1, dy0, dy1;   // part of the downscaled frame the window reads
    int sy0, sy1;          // source rows that part reads
    int sx0, sx1;          // source columns (in the mirrored frame's coordinates) that part reads
    int flip;
    float gain[3];
    double m[6];           // PIL rotate's inverse affine map
    double sx, sy;         // pre-downscale ratios wb / wd, hb / hd
};

// ---- byte of the image crop() is handed: fp32 frame / 255 (torch), mirrored, times gain (fp32), clamped, toimage()'d
// toimage()'s bytescale runs in the precision of the array it is handed: fp32 for the whole fp32 frame (the pre-downscale,
// :130), float64 for the window (np.zeros(new_shape), :150).  The two differ in about 3e-5 of the pixels when a gain is on.
__device__ __forceinline__ int src_byte(unsigned char k, float gain, bool f32) {
    float v = ((float)k / 255.f) * gain;
    v = fminf(fmaxf(v, 0.f), 1.f);
    if (f32) { const float t = v * 255.f; return (int)(t + 0.5f); }
    return (int)((double)v * 255.0 + 0.5);
}

// ---- Resample.c precompute_coeffs for ONE output index xx of an axis in_size -> out_size (bilinear filter)
struct Taps { int xmin, n; double ww; double center, ss; };
__device__ __forceinline__ Taps taps_of(int xx, int in_size, double scale) {
    Taps t;
    double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 1.0 * filterscale;
    t.center = 0.0 + ((double)xx + 0.5) * scale;
    t.ss = 1.0 / filterscale;
    int xmin = (int)(t.center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(t.center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    t.xmin = xmin; t.n = xmax - xmin;
    double ww = 0.0;
    for (int x = 0; x < t.n; ++x) {
        double v = ((double)(x + xmin) - t.center + 0.5) * t.ss;
        if (v < 0.0) v = -v;
        ww += v < 1.0 ? 1.0 - v : 0.0;
    }
    t.ww = ww;
    return t;
}
__device__ __forceinline__ int tap_coeff(const Taps& t, int x) {
    double v = ((double)(x + t.xmin) - t.center + 0.5) * t.ss;
    if (v < 0.0) v = -v;
    double w = v < 1.0 ? 1.0 - v : 0.0;
    if (t.ww != 0.0) w = w / t.ww;
    return (int)(0.5 + w * (double)(1 << CW_PREC));
}
__device__ __forceinline__ int clip8(int ss) { ss >>= CW_PREC; return ss < 0 ? 0 : (ss > 255 ? 255 : ss); }

// Coefficients are computed ONCE per output index of an axis (the double divisions are the expensive part) into a table the
// passes read: per sample 4 axes (pre-downscale x / y over the window's part of the downscaled frame, final resize x / y),
// CW_AXIS_MAX entries each, entry = {xmin, n, coeff[CW_KMAX]}.  n > CW_KMAX (scale factors above 11): n is stored negated and
// the pass computes its taps inline.
#define CW_KMAX 24
#define CW_ENTRY (2 + CW_KMAX)
__device__ __forceinline__ const int* tap_entry(const int* table, int b, int axis, int axis_max, int i) {
    return table + (((size_t)b * 4 + axis) * axis_max + i) * CW_ENTRY;
}

// ------------------------------------------------------------------------------------------------ plan
__global__ void crop_plan_kernel(const double* params, const int* sizes, int Hs, int Ws, int B, int res, CropPlan* plans) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const double* p = params + (size_t)b * 8;
    CropPlan P;
    P.wb = sizes ? sizes[2 * b] : Ws; P.hb = sizes ? sizes[2 * b + 1] : Hs;
    P.flip = p[4] != 0.0;
    for (int c = 0; c < 3; ++c) P.gain[c] = (float)p[5 + c];
    float cx = (float)p[0], cy = (float)p[1], s = (float)p[2];       // fp32 quantities in the reference (torch tensors)
    const double rot = p[3];
    double sf = (double)(s * 200.f) / (double)res;                    // float(scale * size) / float(res), :120
    P.case_b = sf >= 2.0;
    if (!P.case_b) sf = 1.0;
    P.wd = P.wb; P.hd = P.hb; P.sx = P.sy = 1.0;
    if (P.case_b) {
        const double frac = 1.0 / sf;                                 // imresize(img, size=1/scale_factor): (im.size * size).astype(int)
        P.wd = (int)((double)P.wb * frac); P.hd = (int)((double)P.hb * frac);
        if (P.wd < 1) P.wd = 1;
        if (P.hd < 1) P.hd = 1;
        P.sx = (double)P.wb / (double)P.wd; P.sy = (double)P.hb / (double)P.hd;
        const float sff = (float)sf;                                  // fp32 array / Python float: fp32 arithmetic
        cx = cx / sff; cy = cy / sff; s = s / sff;
    }
    // TransformSinglePts(invert=1, rot=0), :136-138: the inverse of [[a,0,tx],[0,a,ty],[0,0,1]] as LAPACK forms it
    const double h = 200.0 * (double)s, a = (double)res / h;
    const double tx = (double)res * (-(double)cx / h + .5), ty = (double)res * (-(double)cy / h + .5);
    const double ia = 1.0 / a, ix = -(tx * ia), iy = -(ty * ia);
    int ulx = (int)ix, uly = (int)iy;
    int brx = (int)(ia * (double)res + ix), bry = (int)(ia * (double)res + iy);
    if (P.case_b) { brx = ulx + res; bry = uly + res; }               // br - (br - ul - res), :141-142
    const int dx = brx - ulx, dy = bry - uly;
    P.cw = dx; P.ch = dy;
    P.pad = (int)ceil(sqrt((double)dx * dx + (double)dy * dy) / 2.0 - (double)dy / 2.0);
    P.rot_mode = 0;
    if (rot != 0.0) { ulx -= P.pad; uly -= P.pad; brx += P.pad; bry += P.pad; }
    P.ulx = ulx; P.uly = uly; P.nw = brx - ulx; P.nh = bry - uly;
    for (int i = 0; i < 6; ++i) P.m[i] = 0.0;
    if (rot != 0.0) {                                                 // Image.rotate(angle, BILINEAR), PIL/Image.py
        double angle = fmod(rot, 360.0);
        if (angle < 0.0) angle += 360.0;
        if (angle == 0.0) P.rot_mode = 0;                             // (still padded and un-padded: a copy)
        else if (angle == 180.0) P.rot_mode = 2;
        else if ((angle == 90.0 || angle == 270.0) && P.nw == P.nh) P.rot_mode = angle == 90.0 ? 3 : 4;
        else {
            P.rot_mode = 1;
            const double ar = -(angle * (3.14159265358979323846 / 180.0));
            const double cs = cos(ar), sn = sin(ar);
            const double rcx = (double)P.nw / 2.0, rcy = (double)P.nh / 2.0;
            P.m[0] = cs; P.m[1] = sn; P.m[3] = -sn; P.m[4] = cs;
            P.m[2] = (P.m[0] * (-rcx) + P.m[1] * (-rcy) + 0.0) + rcx;
            P.m[5] = (P.m[3] * (-rcx) + P.m[4] * (-rcy) + 0.0) + rcy;
        }
    }
    // the part of the downscaled frame the window covers, and the source rows its vertical pass reads
    P.dx0 = ulx < 0 ? 0 : ulx; P.dx1 = brx < P.wd ? brx : P.wd;
    P.dy0 = uly < 0 ? 0 : uly; P.dy1 = bry < P.hd ? bry : P.hd;
    if (P.dx1 < P.dx0) P.dx1 = P.dx0;
    if (P.dy1 < P.dy0) P.dy1 = P.dy0;
    P.sy0 = 0; P.sy1 = 0; P.sx0 = 0; P.sx1 = 0;
    if (P.case_b && P.dy1 > P.dy0) {
        const Taps t0 = taps_of(P.dy0, P.hb, P.sy), t1 = taps_of(P.dy1 - 1, P.hb, P.sy);
        P.sy0 = t0.xmin; P.sy1 = t1.xmin + t1.n;
    }
    if (P.case_b && P.dx1 > P.dx0) {
        const Taps t0 = taps_of(P.dx0, P.wb, P.sx), t1 = taps_of(P.dx1 - 1, P.wb, P.sx);
        P.sx0 = t0.xmin; P.sx1 = t1.xmin + t1.n;
    }
    plans[b] = P;
}

__global__ void crop_coeff_kernel(const CropPlan* plans, int res, int* table, int axis_max) {
    const CropPlan& P = plans[blockIdx.y];
    const int axis = blockIdx.z;
    int count, first, in_size; double scale;
    if (axis == 0) { if (!P.case_b) return; count = P.dx1 - P.dx0; first = P.dx0; in_size = P.wb; scale = P.sx; }
    else if (axis == 1) { if (!P.case_b) return; count = P.dy1 - P.dy0; first = P.dy0; in_size = P.hb; scale = P.sy; }
    else if (axis == 2) { if (P.cw == res) return; count = res; first = 0; in_size = P.cw; scale = (double)P.cw / (double)res; }
    else { if (P.ch == res) return; count = res; first = 0; in_size = P.ch; scale = (double)P.ch / (double)res; }
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count || i >= axis_max) return;
    const Taps t = taps_of(first + i, in_size, scale);
    int* e = table + (((size_t)blockIdx.y * 4 + axis) * axis_max + i) * CW_ENTRY;
    e[0] = t.xmin;
    if (t.n > CW_KMAX) { e[1] = -t.n; return; }
    e[1] = t.n;
    for (int k = 0; k < t.n; ++k) e[2 + k] = tap_coeff(t, k);
}

// ------------------------------------------------------------------------------------------------ pre-downscale
// source pass: S[r - sy0][x - sx0] = the byte image crop() is handed (mirror, byte -> byte map of fp32 / 255 * gain, clamp,
// toimage) as one uchar4 per pixel, for the source rows / columns the window's part of the downscaled frame reads.  The
// horizontal pass then loads ONE dword per tap instead of three bytes and three table look-ups (that pass was bound by its
// vector-memory and LDS instruction count: 110-280 us per batch; now a coalesced stream + register-resident taps).
__global__ __launch_bounds__(256) void crop_src_kernel(const unsigned char* src, int Hs, int Ws, const CropPlan* plans, uchar4* s4, size_t s4_stride,
                                                       int s4_pitch) {
    const CropPlan& P = plans[blockIdx.y];
    if (!P.case_b) return;
    const int ncol = P.sx1 - P.sx0, nrow = P.sy1 - P.sy0;
    const unsigned char* img = src + (size_t)blockIdx.y * Hs * Ws * 3;
    uchar4* out = s4 + (size_t)blockIdx.y * s4_stride;
    __shared__ unsigned char lut[3][256];
    for (int k = threadIdx.x; k < 768; k += blockDim.x) lut[k >> 8][k & 255] = (unsigned char)src_byte((unsigned char)(k & 255), P.gain[k >> 8], true);
    __syncthreads();
    if ((Ws & 3) == 0 && ((size_t)src & 3) == 0) {
        // an item = 4 consecutive SOURCE pixels at a 4-pixel boundary = three aligned dwords (one byte load per channel kept 192 bytes
        // per wave in flight: 0.5 TB/s on frames that are not in any cache); the window's source columns are [lo, hi) -- mirrored or not
        const int lo = P.flip ? P.wb - P.sx1 : P.sx0, hi = lo + ncol;
        const int g0 = lo >> 2, ng = ((hi - 1) >> 2) - g0 + 1;
        const unsigned total = (unsigned)ng * (unsigned)nrow;
        for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
            const unsigned r = i / (unsigned)ng, g = i - r * (unsigned)ng;
            const int c0 = (g0 + (int)g) << 2;
            const unsigned* w = reinterpret_cast<const unsigned*>(img + ((size_t)(P.sy0 + (int)r) * Ws + c0) * 3);
            const unsigned w0 = w[0], w1 = w[1], w2 = w[2];
            const unsigned char b[12] = {(unsigned char)w0, (unsigned char)(w0 >> 8), (unsigned char)(w0 >> 16), (unsigned char)(w0 >> 24),
                                         (unsigned char)w1, (unsigned char)(w1 >> 8), (unsigned char)(w1 >> 16), (unsigned char)(w1 >> 24),
                                         (unsigned char)w2, (unsigned char)(w2 >> 8), (unsigned char)(w2 >> 16), (unsigned char)(w2 >> 24)};
            uchar4* orow = out + (size_t)r * s4_pitch;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int cs = c0 + k;
                if (cs < lo || cs >= hi) continue;
                const int x = (P.flip ? P.wb - 1 - cs : cs) - P.sx0;
                orow[x] = make_uchar4(lut[0][b[3 * k]], lut[1][b[3 * k + 1]], lut[2][b[3 * k + 2]], 0);
            }
        }
        return;
    }
    const long total = (long)ncol * nrow;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int r = (int)(i / ncol), x = (int)(i - (long)r * ncol);
        const int xs = P.sx0 + x;
        const unsigned char* px = img + ((size_t)(P.sy0 + r) * Ws + (P.flip ? P.wb - 1 - xs : xs)) * 3;
        out[(size_t)r * s4_pitch + x] = make_uchar4(lut[0][px[0]], lut[1][px[1]], lut[2][px[2]], 0);
    }
}

// horizontal pass: T1[r - sy0][x - dx0] for source rows r in [sy0, sy1), downscaled columns x in [dx0, dx1).
// A thread owns ONE output column (its <= CW_HREG taps live in registers) and walks CW_HROWS rows of it.
#define CW_HREG 12
#define CW_HROWS 16
template <int NT>
__device__ __forceinline__ void down_h_rows(const uchar4* col, int s4_pitch, uchar4* out, int t1_pitch, int xc, int r0, int nrow, const int* e, int n) {
    int w[NT];
#pragma unroll
    for (int k = 0; k < NT; ++k) w[k] = k < n ? e[2 + k] : 0;
    // taps beyond n read the following pixels of the row (inside the buffer: it is padded by CW_HREG pixels) with weight 0
#pragma unroll 2
    for (int j = 0; j < CW_HROWS; ++j) {
        const int r = r0 + 4 * j;
        if (r >= nrow) break;
        const uchar4* row = col + (size_t)r * s4_pitch;
        int a0 = 1 << (CW_PREC - 1), a1 = a0, a2 = a0;
#pragma unroll
        for (int k = 0; k < NT; ++k) {
            const uchar4 px = row[k];
            a0 += (int)px.x * w[k]; a1 += (int)px.y * w[k]; a2 += (int)px.z * w[k];
        }
        out[(size_t)r * t1_pitch + xc] = make_uchar4((unsigned char)clip8(a0), (unsigned char)clip8(a1), (unsigned char)clip8(a2), 0);
    }
}

__global__ __launch_bounds__(256) void crop_down_h_kernel(const CropPlan* plans, const int* table, int axis_max, const uchar4* s4, size_t s4_stride,
                                                          int s4_pitch, uchar4* t1, size_t t1_stride, int t1_pitch, int col_blocks) {
    const CropPlan& P = plans[blockIdx.y];
    if (!P.case_b) return;
    const int ncol = P.dx1 - P.dx0, nrow = P.sy1 - P.sy0;
    const int cb = blockIdx.x % col_blocks, rb = blockIdx.x / col_blocks;
    const int xc = cb * 64 + (threadIdx.x & 63);
    const int r0 = rb * (4 * CW_HROWS) + (threadIdx.x >> 6);
    if (xc >= ncol || r0 >= nrow) return;
    const uchar4* in = s4 + (size_t)blockIdx.y * s4_stride;
    uchar4* out = t1 + (size_t)blockIdx.y * t1_stride;
    const int* e = tap_entry(table, blockIdx.y, 0, axis_max, xc);
    const int xmin = e[0], n = e[1];
    if (n >= 0 && n <= CW_HREG) {
        // (a sample's columns have n or n + 1 taps: the branch is uniform for almost every wave)
        if (n <= 8) down_h_rows<8>(in + (xmin - P.sx0), s4_pitch, out, t1_pitch, xc, r0, nrow, e, n);
        else down_h_rows<CW_HREG>(in + (xmin - P.sx0), s4_pitch, out, t1_pitch, xc, r0, nrow, e, n);
        return;
    }
    for (int j = 0; j < CW_HROWS; ++j) {               // wide filters (scale factors above 5.5): taps from the table, or computed inline above 11
        const int r = r0 + 4 * j;
        if (r >= nrow) break;
        int a0 = 1 << (CW_PREC - 1), a1 = a0, a2 = a0;
        if (n >= 0) {
            const uchar4* row = in + (size_t)r * s4_pitch + (xmin - P.sx0);
            for (int k = 0; k < n; ++k) {
                const uchar4 px = row[k];
                const int w = e[2 + k];
                a0 += (int)px.x * w; a1 += (int)px.y * w; a2 += (int)px.z * w;
            }
        } else {
            const Taps t = taps_of(P.dx0 + xc, P.wb, P.sx);
            const uchar4* row = in + (size_t)r * s4_pitch + (t.xmin - P.sx0);
            for (int k = 0; k < t.n; ++k) {
                const uchar4 px = row[k];
                const int w = tap_coeff(t, k);
                a0 += (int)px.x * w; a1 += (int)px.y * w; a2 += (int)px.z * w;
            }
        }
        out[(size_t)r * t1_pitch + xc] = make_uchar4((unsigned char)clip8(a0), (unsigned char)clip8(a1), (unsigned char)clip8(a2), 0);
    }
}

// vertical pass: D[y - dy0][x - dx0]
__global__ void crop_down_v_kernel(const CropPlan* plans, const int* table, int axis_max, const uchar4* t1, size_t t1_stride, int t1_pitch,
                                   uchar4* d, size_t d_stride, int d_pitch) {
    const CropPlan& P = plans[blockIdx.y];
    if (!P.case_b) return;
    const int ncol = P.dx1 - P.dx0, nrow = P.dy1 - P.dy0;
    const long total = (long)ncol * nrow;
    const uchar4* in = t1 + (size_t)blockIdx.y * t1_stride;
    uchar4* out = d + (size_t)blockIdx.y * d_stride;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int yr = (int)(i / ncol), xc = (int)(i - (long)yr * ncol);
        const int* e = tap_entry(table, blockIdx.y, 1, axis_max, yr);
        int a0 = 1 << (CW_PREC - 1), a1 = a0, a2 = a0;
        if (e[1] >= 0) {
            const int ymin = e[0], n = e[1];
            for (int k = 0; k < n; ++k) {
                const uchar4 px = in[(size_t)(ymin + k - P.sy0) * t1_pitch + xc];
                const int w = e[2 + k];
                a0 += (int)px.x * w; a1 += (int)px.y * w; a2 += (int)px.z * w;
            }
        } else {
            const Taps t = taps_of(P.dy0 + yr, P.hb, P.sy);
            for (int k = 0; k < t.n; ++k) {
                const uchar4 px = in[(size_t)(t.xmin + k - P.sy0) * t1_pitch + xc];
                const int w = tap_coeff(t, k);
                a0 += (int)px.x * w; a1 += (int)px.y * w; a2 += (int)px.z * w;
            }
        }
        out[(size_t)yr * d_pitch + xc] = make_uchar4((unsigned char)clip8(a0), (unsigned char)clip8(a1), (unsigned char)clip8(a2), 0);
    }
}

// ---- pixel (yy, xx) of the zero-padded window new_img (:150-164)
__device__ __forceinline__ void window_px(const CropPlan& P, const unsigned char* img, int Ws, const uchar4* d, int d_pitch, int yy, int xx, int (&v)[3]) {
    const int gx = P.ulx + xx, gy = P.uly + yy;
    v[0] = v[1] = v[2] = 0;
    if (gx < 0 || gy < 0 || gx >= P.wd || gy >= P.hd) return;
    if (P.case_b) {
        const uchar4 px = d[(size_t)(gy - P.dy0) * d_pitch + (gx - P.dx0)];
        v[0] = px.x; v[1] = px.y; v[2] = px.z;
    } else {
        const unsigned char* px = img + ((size_t)gy * Ws + (P.flip ? P.wb - 1 - gx : gx)) * 3;
        v[0] = src_byte(px[0], P.gain[0], false); v[1] = src_byte(px[1], P.gain[1], false); v[2] = src_byte(px[2], P.gain[2], false);
    }
}

// ------------------------------------------------------------------------------------------------ rotate + un-pad
// U[y][x] = rotate(new_img)[y + pad][x + pad] (Geometry.c ImagingGenericTransform + bilinear_filter32RGB)
__global__ void crop_rotate_kernel(const unsigned char* src, int Hs, int Ws, const CropPlan* plans, const uchar4* d, size_t d_stride, int d_pitch,
                                   uchar4* u, size_t u_stride, int u_pitch) {
    const CropPlan& P = plans[blockIdx.y];
    if (P.rot_mode == 0) return;
    const long total = (long)P.cw * P.ch;
    const unsigned char* img = src + (size_t)blockIdx.y * Hs * Ws * 3;
    const uchar4* dd = d + (size_t)blockIdx.y * d_stride;
    uchar4* out = u + (size_t)blockIdx.y * u_stride;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int y = (int)(i / P.cw), x = (int)(i - (long)y * P.cw);
        const int X = x + P.pad, Y = y + P.pad;
        int o[3] = {0, 0, 0};
        if (P.rot_mode == 2) window_px(P, img, Ws, dd, d_pitch, P.nh - 1 - Y, P.nw - 1 - X, o);
        else if (P.rot_mode == 3) window_px(P, img, Ws, dd, d_pitch, X, P.nw - 1 - Y, o);           // ROTATE_90: out[y][x] = in[x][w-1-y]
        else if (P.rot_mode == 4) window_px(P, img, Ws, dd, d_pitch, P.nh - 1 - X, Y, o);           // ROTATE_270
        else {
            const double xc = (double)X + 0.5, yc = (double)Y + 0.5;
            double xin = P.m[0] * xc + P.m[1] * yc + P.m[2];
            double yin = P.m[3] * xc + P.m[4] * yc + P.m[5];
            if (!(xin < 0.0 || xin >= (double)P.nw || yin < 0.0 || yin >= (double)P.nh)) {
                xin -= 0.5; yin -= 0.5;
                const int fx = (int)floor(xin), fy = (int)floor(yin);
                const double ddx = xin - (double)fx, ddy = yin - (double)fy;
                const int x0 = fx < 0 ? 0 : (fx < P.nw ? fx : P.nw - 1), x1 = fx + 1 < 0 ? 0 : (fx + 1 < P.nw ? fx + 1 : P.nw - 1);
                const int y0 = fy < 0 ? 0 : (fy < P.nh ? fy : P.nh - 1);
                int p00[3], p01[3], p10[3], p11[3];
                window_px(P, img, Ws, dd, d_pitch, y0, x0, p00); window_px(P, img, Ws, dd, d_pitch, y0, x1, p01);
                const bool has1 = fy + 1 >= 0 && fy + 1 < P.nh;
                if (has1) { window_px(P, img, Ws, dd, d_pitch, fy + 1, x0, p10); window_px(P, img, Ws, dd, d_pitch, fy + 1, x1, p11); }
                for (int c = 0; c < 3; ++c) {
                    const double v1 = (double)p00[c] + (double)(p01[c] - p00[c]) * ddx;
                    const double v2 = has1 ? (double)p10[c] + (double)(p11[c] - p10[c]) * ddx : v1;
                    o[c] = (int)(unsigned char)(v1 + (v2 - v1) * ddy);
                }
            }
        }
        out[(size_t)y * u_pitch + x] = make_uchar4((unsigned char)o[0], (unsigned char)o[1], (unsigned char)o[2], 0);
    }
}

// ---- pixel (y, x) of the un-padded window the final resize reads
__device__ __forceinline__ void crop_px(const CropPlan& P, const unsigned char* img, int Ws, const uchar4* d, int d_pitch, const uchar4* u, int u_pitch,
                                        int y, int x, int (&v)[3]) {
    if (P.rot_mode != 0) { const uchar4 px = u[(size_t)y * u_pitch + x]; v[0] = px.x; v[1] = px.y; v[2] = px.z; return; }
    const int off = (P.nw - P.cw) / 2;         // rot given but a multiple of 360 degrees: padded, copied, un-padded
    window_px(P, img, Ws, d, d_pitch, y + off, x + off, v);
}

// ------------------------------------------------------------------------------------------------ final resize
// horizontal pass: T2[y][uo], y < ch, uo < res (a copy when cw == res)
__global__ void crop_resize_h_kernel(const unsigned char* src, int Hs, int Ws, const CropPlan* plans, const uchar4* d, size_t d_stride, int d_pitch,
                                     const uchar4* u, size_t u_stride, int u_pitch, int res, const int* table, int axis_max, uchar4* t2, size_t t2_stride) {
    const CropPlan& P = plans[blockIdx.y];
    if (P.cw == res && P.ch == res) return;                      // Image.resize to the same size: a copy (done by the last kernel)
    const long total = (long)P.ch * res;
    const unsigned char* img = src + (size_t)blockIdx.y * Hs * Ws * 3;
    const uchar4* dd = d + (size_t)blockIdx.y * d_stride;
    const uchar4* uu = u + (size_t)blockIdx.y * u_stride;
    uchar4* out = t2 + (size_t)blockIdx.y * t2_stride;
    const double scale = (double)P.cw / (double)res;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int y = (int)(i / res), uo = (int)(i - (long)y * res);
        int v[3];
        if (P.cw == res) {
            crop_px(P, img, Ws, dd, d_pitch, uu, u_pitch, y, uo, v);
        } else {
            const int* e = tap_entry(table, blockIdx.y, 2, axis_max, uo);
            int a0 = 1 << (CW_PREC - 1), a1 = a0, a2 = a0;
            if (e[1] >= 0) {
                const int xmin = e[0], n = e[1];
                for (int k = 0; k < n; ++k) {
                    int px[3];
                    crop_px(P, img, Ws, dd, d_pitch, uu, u_pitch, y, xmin + k, px);
                    const int w = e[2 + k];
                    a0 += px[0] * w; a1 += px[1] * w; a2 += px[2] * w;
                }
            } else {
                const Taps t = taps_of(uo, P.cw, scale);
                for (int k = 0; k < t.n; ++k) {
                    int px[3];
                    crop_px(P, img, Ws, dd, d_pitch, uu, u_pitch, y, t.xmin + k, px);
                    const int w = tap_coeff(t, k);
                    a0 += px[0] * w; a1 += px[1] * w; a2 += px[2] * w;
                }
            }
            v[0] = clip8(a0); v[1] = clip8(a1); v[2] = clip8(a2);
        }
        out[(size_t)y * res + uo] = make_uchar4((unsigned char)v[0], (unsigned char)v[1], (unsigned char)v[2], 0);
    }
}

// vertical pass + im_to_torch (uint8 / 255, utils/imutils.py:31-36) + network layouts
__global__ void crop_resize_v_kernel(const unsigned char* src, int Hs, int Ws, const CropPlan* plans, const uchar4* d, size_t d_stride, int d_pitch,
                                     const uchar4* u, size_t u_stride, int u_pitch, const uchar4* t2, size_t t2_stride, int res,
                                     const int* table, int axis_max, bf16* out4, float* outf, unsigned char* out8) {
    const CropPlan& P = plans[blockIdx.y];
    const int b = blockIdx.y;
    const long total = (long)res * res;
    const unsigned char* img = src + (size_t)b * Hs * Ws * 3;
    const uchar4* dd = d + (size_t)b * d_stride;
    const uchar4* uu = u + (size_t)b * u_stride;
    const uchar4* tt = t2 + (size_t)b * t2_stride;
    const double scale = (double)P.ch / (double)res;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int vo = (int)(i / res), uo = (int)(i - (long)vo * res);
        int v[3];
        if (P.cw == res && P.ch == res) {
            crop_px(P, img, Ws, dd, d_pitch, uu, u_pitch, vo, uo, v);
        } else if (P.ch == res) {
            const uchar4 px = tt[(size_t)vo * res + uo]; v[0] = px.x; v[1] = px.y; v[2] = px.z;
        } else {
            const int* e = tap_entry(table, b, 3, axis_max, vo);
            int a0 = 1 << (CW_PREC - 1), a1 = a0, a2 = a0;
            if (e[1] >= 0) {
                const int ymin = e[0], n = e[1];
                for (int k = 0; k < n; ++k) {
                    const uchar4 px = tt[(size_t)(ymin + k) * res + uo];
                    const int w = e[2 + k];
                    a0 += (int)px.x * w; a1 += (int)px.y * w; a2 += (int)px.z * w;
                }
            } else {
                const Taps t = taps_of(vo, P.ch, scale);
                for (int k = 0; k < t.n; ++k) {
                    const uchar4 px = tt[(size_t)(t.xmin + k) * res + uo];
                    const int w = tap_coeff(t, k);
                    a0 += (int)px.x * w; a1 += (int)px.y * w; a2 += (int)px.z * w;
                }
            }
            v[0] = clip8(a0); v[1] = clip8(a1); v[2] = clip8(a2);
        }
        const float o0 = (float)v[0] / 255.f, o1 = (float)v[1] / 255.f, o2 = (float)v[2] / 255.f;
        const size_t t = (size_t)b * total + i;
        if (out4) { bf16x4 ob = {(bf16)o0, (bf16)o1, (bf16)o2, (bf16)0.f}; *reinterpret_cast<bf16x4*>(out4 + t * 4) = ob; }
        if (outf) {
            outf[((size_t)b * 3 + 0) * total + i] = o0; outf[((size_t)b * 3 + 1) * total + i] = o1; outf[((size_t)b * 3 + 2) * total + i] = o2;
        }
        if (out8) { out8[t * 3] = (unsigned char)v[0]; out8[t * 3 + 1] = (unsigned char)v[1]; out8[t * 3 + 2] = (unsigned char)v[2]; }
    }
}

// ------------------------------------------------------------------------------------------------ host
struct CropLayout {
    int pad_b, nw_b, cw_max;           // worst-case padded window of the pre-downscale case, worst-case un-padded window
    size_t plans, table, s4, t1, d, u, t2, total; // byte offsets
    int axis_max;
    size_t s4_stride, t1_stride, d_stride, u_stride, t2_stride;   // per-sample strides in pixels
    int s4_pitch, t1_pitch, d_pitch, u_pitch;
};

static CropLayout crop_layout(int B, int Hs, int Ws, int res) {
    CropLayout L;
    // pre-downscale case: window = res + 2 * ceil(res * (sqrt(2) - 1) / 2); other case: side = trunc(200 * scale) < 2 * res + 2
    L.pad_b = (int)ceil((double)res * 0.70710678118654752 - (double)res / 2.0) + 1;
    L.nw_b = res + 2 * L.pad_b;
    L.cw_max = 2 * res + 2;
    auto align = [](size_t v) { return (v + 255) & ~(size_t)255; };
    size_t off = 0;
    L.plans = off; off = align(off + (size_t)B * sizeof(CropPlan));
    L.axis_max = L.nw_b > res ? L.nw_b : res;
    L.table = off; off = align(off + (size_t)B * 4 * L.axis_max * CW_ENTRY * sizeof(int));
    L.s4_pitch = Ws; L.s4_stride = (size_t)Hs * Ws;                     // (+ CW_HREG pixels behind the last sample: zero-weight taps)
    L.s4 = off; off = align(off + ((size_t)B * L.s4_stride + CW_HREG) * 4);
    L.t1_pitch = L.nw_b; L.t1_stride = (size_t)Hs * L.t1_pitch;
    L.t1 = off; off = align(off + (size_t)B * L.t1_stride * 4);
    L.d_pitch = L.nw_b; L.d_stride = (size_t)L.nw_b * L.d_pitch;
    L.d = off; off = align(off + (size_t)B * L.d_stride * 4);
    L.u_pitch = L.cw_max; L.u_stride = (size_t)L.cw_max * L.u_pitch;
    L.u = off; off = align(off + (size_t)B * L.u_stride * 4);
    L.t2_stride = (size_t)L.cw_max * res;
    L.t2 = off; off = align(off + (size_t)B * L.t2_stride * 4);
    L.total = off;
    return L;
}

size_t pa_crop_workspace_size(int B, int Hs, int Ws, int res) { return crop_layout(B, Hs, Ws, res).total; }

// Upper bound of the bytes a pa_launch_crop call moves, from the geometry of its own buffers (bench.py's floor): every stage reads its
// input image once and writes its output image once at the WORST-CASE window sizes of the layout above (the actual windows depend on
// the per-sample scale and are smaller; the pre-downscale stages only run for scale * 200 / res >= 2).
void pa_crop_bytes_bound(int B, int Hs, int Ws, int res, double* rd, double* wr) {
    const CropLayout L = crop_layout(B, Hs, Ws, res);
    const double frame3 = (double)Hs * Ws * 3, frame4 = (double)Hs * Ws * 4, t1 = (double)Hs * L.nw_b * 4, d = (double)L.nw_b * L.nw_b * 4,
                 u = (double)L.cw_max * L.cw_max * 4, t2 = (double)L.cw_max * res * 4, out = (double)res * res * 4 * 2;
    *rd = B * (frame3 + frame4 + t1 + d + u + t2) + (double)B * 4 * L.axis_max * CW_ENTRY * sizeof(int) * 3;      // (coefficient tables: three passes read them)
    *wr = B * (frame4 + t1 + d + u + t2 + out) + (double)B * 4 * L.axis_max * CW_ENTRY * sizeof(int);
}

int pa_launch_crop(const unsigned char* src, int Hs, int Ws, const int* sizes, const double* params, int B, int res, void* workspace,
                   bf16* out4, float* outf, unsigned char* out8, hipStream_t st) {
    const CropLayout L = crop_layout(B, Hs, Ws, res);
    char* ws = reinterpret_cast<char*>(workspace);
    CropPlan* plans = reinterpret_cast<CropPlan*>(ws + L.plans);
    int* table = reinterpret_cast<int*>(ws + L.table);
    uchar4* s4 = reinterpret_cast<uchar4*>(ws + L.s4);
    uchar4* t1 = reinterpret_cast<uchar4*>(ws + L.t1);
    uchar4* d = reinterpret_cast<uchar4*>(ws + L.d);
    uchar4* u = reinterpret_cast<uchar4*>(ws + L.u);
    uchar4* t2 = reinterpret_cast<uchar4*>(ws + L.t2);
    hipLaunchKernelGGL(crop_plan_kernel, dim3((B + 63) / 64), dim3(64), 0, st, params, sizes, Hs, Ws, B, res, plans);
    const int T = 256;
    auto blocks = [&](long n) { long g = (n + T - 1) / T; return (unsigned)(g < 1 ? 1 : (g > 1024 ? 1024 : g)); };
    hipLaunchKernelGGL(crop_coeff_kernel, dim3((L.axis_max + 63) / 64, B, 4), dim3(64), 0, st, plans, res, table, L.axis_max);
    hipLaunchKernelGGL(crop_src_kernel, dim3(96, B), dim3(T), 0, st, src, Hs, Ws, plans, s4, L.s4_stride, L.s4_pitch);
    const int col_blocks = (L.axis_max + 63) / 64, row_blocks = (Hs + 4 * CW_HROWS - 1) / (4 * CW_HROWS);
    hipLaunchKernelGGL(crop_down_h_kernel, dim3(col_blocks * row_blocks, B), dim3(T), 0, st, plans, table, L.axis_max, s4, L.s4_stride, L.s4_pitch,
                       t1, L.t1_stride, L.t1_pitch, col_blocks);
    hipLaunchKernelGGL(crop_down_v_kernel, dim3(blocks((long)L.nw_b * L.nw_b), B), dim3(T), 0, st, plans, table, L.axis_max, t1, L.t1_stride, L.t1_pitch,
                       d, L.d_stride, L.d_pitch);
    hipLaunchKernelGGL(crop_rotate_kernel, dim3(blocks((long)L.cw_max * L.cw_max / 2), B), dim3(T), 0, st, src, Hs, Ws, plans, d, L.d_stride, L.d_pitch,
                       u, L.u_stride, L.u_pitch);
    hipLaunchKernelGGL(crop_resize_h_kernel, dim3(blocks((long)L.cw_max * res / 2), B), dim3(T), 0, st, src, Hs, Ws, plans, d, L.d_stride, L.d_pitch,
                       u, L.u_stride, L.u_pitch, res, table, L.axis_max, t2, L.t2_stride);
    hipLaunchKernelGGL(crop_resize_v_kernel, dim3(blocks((long)res * res), B), dim3(T), 0, st, src, Hs, Ws, plans, d, L.d_stride, L.d_pitch,
                       u, L.u_stride, L.u_pitch, t2, L.t2_stride, res, table, L.axis_max, out4, outf, out8);
    return (int)hipGetLastError();
}
