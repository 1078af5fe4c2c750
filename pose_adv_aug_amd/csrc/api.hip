// extern "C" boundary of libposeadv_hip.so (declared in include/poseadv.h).
#include "../../include/poseadv.h"
#include "net.h"
#include <stdio.h>
#include <string.h>
#include <new>
#include <string>

static thread_local char g_err[1024] = "";
void pa_set_error(const char* what, hipError_t e, const char* file, int line) {
    snprintf(g_err, sizeof g_err, "%s:%d: %s -> %s", file, line, what, hipGetErrorString(e));
}
void pa_set_error_msg(const char* msg) { snprintf(g_err, sizeof g_err, "%s", msg); }

struct pa_net { Net n; };
#define ST(s) reinterpret_cast<hipStream_t>(s)
#define TRY(x) do { int _r = (x); if (_r) { if (!g_err[0]) pa_set_error("kernel launch", (hipError_t)_r, __FILE__, __LINE__); return _r; } } while (0)

extern "C" {

const char* pa_last_error(void) { return g_err; }
float pa_grad_scale(void) { return PA_GRAD_SCALE; }
int pa_dtype(void) { return PA_DTYPE_ID; }
int pa_version(void) { return 1; }

int pa_gaussian_heatmap(const double* pts, float* out, int B, int J, int H, int W, void* s) {
    TRY(pa_launch_gaussian_heatmap(pts, out, B, J, H, W, ST(s))); return 0;
}
int pa_weighted_l2(const float* pred, const float* gt, const float* weight, size_t n, float* loss, void* s) {
    TRY(pa_launch_weighted_l2(pred, gt, weight, n, loss, ST(s))); return 0;
}
int pa_get_preds(const float* maps, int B, int J, int H, int W, float* preds, float* maxval, void* s) {
    TRY(pa_launch_argmax(maps, (long)J * H * W, (long)H * W, 1, B, J, H, W, preds, maxval, ST(s))); return 0;
}
int pa_final_preds(const float* maps, const float* center, const float* scale, const float* rot, int B, int J, int H, int W,
                   float* out, float* scratch_preds, void* s) {
    TRY(pa_launch_argmax(maps, (long)J * H * W, (long)H * W, 1, B, J, H, W, scratch_preds, nullptr, ST(s)));
    TRY(pa_launch_final_preds(maps, (long)J * H * W, (long)H * W, 1, scratch_preds, center, scale, rot, B, J, H, W, out, ST(s)));
    return 0;
}
int pa_pck(const float* pred, const float* gt, const float* norm, float boundary, const int32_t* idxs, int nidx, float thr,
           const float* vis, int B, int J, float* acc, float* person, float* dists, void* s) {
    TRY(pa_launch_pck(pred, gt, norm, boundary, idxs, nidx, thr, vis, B, J, acc, person, dists, ST(s))); return 0;
}
int pa_affine_params(const double* params, int B, int res_in, int res_out, double* t_out, double* tinv_in, void* s) {
    TRY(pa_launch_affine_params(params, B, res_in, res_out, t_out, tinv_in, ST(s))); return 0;
}
int pa_transform_pts(const float* pts, const double* params, const double* t, int B, int J, float width, double* out,
                     float* pts_img, void* s) {
    TRY(pa_launch_transform_pts(pts, params, t, B, J, width, nullptr, out, pts_img, ST(s))); return 0;
}
int pa_transform_pts_sized(const float* pts, const double* params, const double* t, int B, int J, const int32_t* sizes, double* out,
                           float* pts_img, void* s) {
    if (!sizes) { pa_set_error_msg("pa_transform_pts_sized: sizes is NULL"); return 1; }
    TRY(pa_launch_transform_pts(pts, params, t, B, J, 0.f, sizes, out, pts_img, ST(s))); return 0;
}
int pa_affine_warp_bilinear(const uint8_t* src, int Hs, int Ws, const double* tinv, const double* params, int B, int res,
                            void* out4, float* outf, void* s) {
    TRY(pa_launch_warp(src, Hs, Ws, nullptr, tinv, params, B, res, reinterpret_cast<bf16*>(out4), outf, ST(s))); return 0;
}
int pa_affine_warp_bilinear_sized(const uint8_t* src, int Hs, int Ws, const int32_t* sizes, const double* tinv, const double* params,
                                  int B, int res, void* out4, float* outf, void* s) {
    if (!sizes) { pa_set_error_msg("pa_affine_warp_bilinear_sized: sizes is NULL"); return 1; }
    TRY(pa_launch_warp(src, Hs, Ws, sizes, tinv, params, B, res, reinterpret_cast<bf16*>(out4), outf, ST(s))); return 0;
}
size_t pa_crop_workspace_bytes(int B, int Hs, int Ws, int res) { return pa_crop_workspace_size(B, Hs, Ws, res); }
int pa_crop_design_bytes(int B, int Hs, int Ws, int res, double* out) { if (!out) return 1; pa_crop_bytes_bound(B, Hs, Ws, res, out, out + 1); return 0; }
// calibration kernel of bench.py's floor (elementwise.hip): dst[i] = src[i], 16 bytes per lane, grid-stride over `bytes` (a multiple of 16)
int pa_copy_probe(void* dst, const void* src, size_t bytes, void* s) {
    g_err[0] = 0;
    if (!dst || !src || (bytes & 15)) { pa_set_error_msg("pa_copy_probe: NULL buffer or a size that is not a multiple of 16"); return 1; }
    TRY(pa_launch_copy16(dst, src, bytes, ST(s))); return 0;
}
// the same copy in another form (1: one 16-byte chunk per thread, no loop; 2: grid-stride with non-temporal accesses; 0 = pa_copy_probe):
// bench.py reports which form reaches what on the box, beside the guide's 6.29 TB/s "float4 copy"
int pa_copy_probe_form(void* dst, const void* src, size_t bytes, int form, void* s) {
    g_err[0] = 0;
    if (!dst || !src || (bytes & 15) || form < 0 || form > 2) { pa_set_error_msg("pa_copy_probe_form: NULL buffer, a size that is not a multiple of 16, or form not in 0..2"); return 1; }
    TRY(pa_launch_copy16(dst, src, bytes, ST(s), form)); return 0;
}
// fp32 copies of the augmentation parameters the metrics take (c [B][2], s [B], r [B]; data.py handed them over through three framework
// elementwise kernels per step): csr = [4 B] floats = c | s | r, from params [B][8] float64
int pa_params_csr(const double* params, int B, float* csr, void* s) {
    g_err[0] = 0;
    if (!params || !csr || B < 1) { pa_set_error_msg("pa_params_csr: bad arguments"); return 1; }
    TRY(pa_launch_params_csr(params, B, csr, ST(s))); return 0;
}
int pa_crop(const uint8_t* src, int Hs, int Ws, const int32_t* sizes, const double* params, int B, int res, void* workspace,
            void* out4, float* outf, uint8_t* out8, void* s) {
    if (!src || !params || !workspace || B <= 0 || res <= 0) { pa_set_error_msg("pa_crop: bad arguments"); return 1; }
    TRY(pa_launch_crop(src, Hs, Ws, sizes, params, B, res, workspace, reinterpret_cast<bf16*>(out4), outf, out8, ST(s))); return 0;
}
int pa_flip_lr_nhwc4(const void* src, void* dst, int B, int H, int W, void* s) {
    TRY(pa_launch_flip_lr_nhwc4(reinterpret_cast<const bf16*>(src), reinterpret_cast<bf16*>(dst), B, H, W, ST(s))); return 0;
}
int pa_flip_tta_merge(const float* out, const float* out_flipped, float* merged, int B, int J, int H, int W, void* s) {
    if (J != 16) { pa_set_error_msg("pa_flip_tta_merge: the left/right joint table is MPII's (16 joints)"); return 1; }
    TRY(pa_launch_flip_tta_merge(out, out_flipped, merged, B, H, W, ST(s))); return 0;
}
int pa_sample_aug(const float* meta, const int32_t* scale_idx, const int32_t* rot_idx, int mode, uint64_t seed, uint64_t step,
                  int B, double* params, void* s) {
    if (mode != 0 && ((mode != 3 && !scale_idx) || (mode != 2 && !rot_idx))) { pa_set_error_msg("pa_sample_aug: the agent laws need bin indices"); return 1; }
    TRY(pa_launch_sample_aug(meta, scale_idx, rot_idx, mode, seed, step, nullptr, B, params, ST(s))); return 0;
}
int pa_sample_aug_given(const float* meta, const int32_t* scale_idx, const int32_t* rot_idx, int mode, const double* draws,
                        int B, double* params, void* s) {
    if (!draws) { pa_set_error_msg("pa_sample_aug_given: draws is NULL"); return 1; }
    if (mode != 0 && ((mode != 3 && !scale_idx) || (mode != 2 && !rot_idx))) { pa_set_error_msg("pa_sample_aug_given: the agent laws need bin indices"); return 1; }
    TRY(pa_launch_sample_aug(meta, scale_idx, rot_idx, mode, 0, 0, draws, B, params, ST(s))); return 0;
}
int pa_sample_categorical(const float* logits, int B, int K, uint64_t seed, uint64_t step, unsigned slot, float* probs,
                          int32_t* idx, void* s) {
    if (K > 64) { pa_set_error_msg("pa_sample_categorical: K <= 64"); return 1; }
    TRY(pa_launch_sample_categorical(logits, B, K, seed, step, slot, probs, idx, ST(s))); return 0;
}
int pa_sample_dropout_masks(const float* logits, int B, int cells, int dropout_num, uint64_t seed, uint64_t step, const double* uniforms,
                            float* probs, float* masks, int32_t* indexes, void* s) {
    if (cells < 1 || cells > 64 || dropout_num < 0 || dropout_num >= cells || !masks) {
        pa_set_error_msg("pa_sample_dropout_masks: need 1 <= cells <= 64, 0 <= dropout_num < cells, masks != NULL"); return 1;
    }
    TRY(pa_launch_sample_dropout_masks(logits, B, cells, dropout_num, seed, step, uniforms, probs, masks, indexes, ST(s))); return 0;
}
int pa_cell_mask(const void* x, const float* masks, void* out, int B, int H, int W, int C, void* s) {
    PaEpilogue e; memset(&e, 0, sizeof e); e.mode = PA_OUT_PLAIN;
    TRY(pa_launch_cell_mask(pa_plain(reinterpret_cast<const bf16*>(x)), masks, e, reinterpret_cast<bf16*>(out), B, H, W, C, ST(s))); return 0;
}
int pa_rmsprop_step(float* p, const float* g, float* v, size_t n, float lr, float alpha, float eps, float gscale, void* s) {
    TRY(pa_launch_rmsprop(p, g, v, n, lr, alpha, eps, gscale, nullptr, ST(s))); return 0;
}
int pa_rmsprop_skipped_steps(long long* count, void* s) { if (!count) return 1; TRY(pa_rmsprop_skipped(nullptr, count, ST(s))); return 0; }
// the same step with the optimizer's OWN skip state (two int32 on the device, zeroed by the caller once): optimizers on different
// streams do not share the "this gradient is non-finite" flag
int pa_rmsprop_step_state(float* p, const float* g, float* v, size_t n, float lr, float alpha, float eps, float gscale, int32_t* state, void* s) {
    if (!state) { pa_set_error_msg("pa_rmsprop_step_state: state is NULL (two int32 on the device)"); return 1; }
    TRY(pa_launch_rmsprop(p, g, v, n, lr, alpha, eps, gscale, state, ST(s))); return 0;
}
int pa_rmsprop_skipped_steps_state(const int32_t* state, long long* count, void* s) {
    if (!count || !state) return 1;
    TRY(pa_rmsprop_skipped(state, count, ST(s))); return 0;
}
int pa_nchw_to_nhwc(const float* src, void* dst, int B, int C, int H, int W, void* s) {
    TRY(pa_launch_nchw_f32_to_nhwc_bf16(src, reinterpret_cast<bf16*>(dst), B, C, H, W, ST(s))); return 0;
}
int pa_nhwc_to_nchw(const void* src, float* dst, int B, int C, int H, int W, void* s) {
    TRY(pa_launch_nhwc_bf16_to_nchw_f32(pa_plain(reinterpret_cast<const bf16*>(src)), dst, B, C, H, W, ST(s))); return 0;
}

// ---------------------------------------------------------------------------- residual block op
struct ResidualOp {
    Net n; Residual r; Act in;
    size_t build(int B, int H, int W, int C, char* base) {
        Arena a; a.base = base;
        n.loss_dev = a.get<float>(64);
        a.take(0);
        n.stats_arena = reinterpret_cast<float*>(base); n.stats_arena_floats = a.off / sizeof(float);
        n.prep_jobs = a.get<PaPrepJob>(n.convs.size());
        n.red_jobs = a.get<PaWgradReduceJob>(n.convs.size());
        n.bneval_jobs = a.get<PaBnEvalJob>(n.bns.size());
        in = n.new_act(a, B, H, W, C, nullptr, true);
        r.layout(n, a, B, H, W, true);
        n.layout_shared(a);
        a.take(0);
        return a.off;
    }
};

size_t pa_residual_workspace_bytes(int B, int H, int W, int C) {
    ResidualOp op; op.n.is_agent = true; op.n.B = B; op.r.declare(op.n, "", C, C, false);      // (same state as pa_residual_fwd_bwd: the layout depends on it)
    return op.build(B, H, W, C, nullptr);
}

int pa_residual_fwd_bwd(const float* x, const float* dy, const float* params, float* y, float* dx, float* grads, float* buffers,
                        int B, int C, int H, int W, void* ws, void* s) {
    g_err[0] = 0;
    ResidualOp op; Net& n = op.n;
    n.is_agent = true; n.B = B; n.st = ST(s);
    op.r.declare(n, "", C, C, false);
    op.build(B, H, W, C, reinterpret_cast<char*>(ws));
    n.params = const_cast<float*>(params); n.grads = grads; n.buffers = buffers;
    TRY(n.upload_tables());
    TRY(n.prepare_weights());
    TRY(n.begin_step());
    n.train_bn = true;
    TRY(pa_launch_nchw_f32_to_nhwc_bf16(x, op.in.raw, B, C, H, W, n.st));
    TRY(op.r.fwd(n, op.in));
    TRY(pa_launch_nhwc_bf16_to_nchw_f32(n.op(op.r.x3), y, B, C, H, W, n.st));
    if (dy) {
        bf16* g = op.r.x1.grad;    // borrow as staging for the incoming gradient (overwritten later by bwd)
        // x1 is narrower than x3: stage in the input's gradient buffer instead
        g = op.in.grad;
        TRY(pa_launch_nchw_f32_to_nhwc_bf16(dy, g, B, C, H, W, n.st));
        TRY(pa_launch_ep_apply(pa_plain(g), n.final_ep(op.r.x3), op.r.x3.grad, (size_t)B * H * W, C, n.st));
        TRY(n.finish_grad(op.r.x3));
        TRY(op.r.bwd(n, op.in, pa_none(), true));
        TRY(n.reduce_grads());
        TRY(pa_launch_nhwc_bf16_to_nchw_f32(pa_plain(op.in.grad), dx, B, C, H, W, n.st));
    }
    PA_CHECK(hipStreamSynchronize(n.st));
    return 0;
}

// ---------------------------------------------------------------------------- single convolution ops
// Plain conv2d (k = 1 or 3, stride 1, 'same') through the implicit-GEMM kernels: NCHW fp32 in/out,
// PyTorch-layout fp32 weights; bf16 operands, fp32 accumulation.  mode: 0 forward (y = conv(x,w)+b),
// 1 data gradient (dx = conv^T(dy, w)), 2 weight gradient (dw, db from dy and x).
struct ConvOp {
    Net n; ConvLayer c; bf16 *xin = nullptr, *yout = nullptr;
    size_t build(int B, int H, int W, char* base) {
        Arena a; a.base = base;
        n.prep_jobs = a.get<PaPrepJob>(1); n.red_jobs = a.get<PaWgradReduceJob>(1); n.bneval_jobs = a.get<PaBnEvalJob>(1);
        n.layout_conv(c, a, B * H * W, H, W);
        xin = a.get<bf16>((size_t)B * H * W * c.pcin);
        yout = a.get<bf16>((size_t)B * H * W * c.pcout);
        n.layout_shared(a);
        a.take(0);
        return a.off;
    }
};

size_t pa_conv2d_workspace_bytes(int B, int Cin, int Cout, int H, int W, int k) {
    ConvOp op; op.n.is_agent = true; op.n.declare_conv(op.c, "c", Cin, Cout, k, false);
    return op.build(B, H, W, nullptr);
}

int pa_conv2d(int mode, const float* a_in, const float* b_in, const float* w, const float* bias, float* out, float* out2,
              int B, int Cin, int Cout, int H, int W, int k, void* ws, void* s) {
    g_err[0] = 0;
    if (Cin % 64 || Cout % 64 || (k != 1 && k != 3)) { pa_set_error_msg("pa_conv2d: channels % 64 == 0, k in {1,3}"); return 1; }
    ConvOp op; Net& n = op.n; n.is_agent = true; n.B = B; n.st = ST(s);
    // mode 2 without out2: no bias gradient wanted (the case of every conv in front of a BatchNorm)
    n.declare_conv(op.c, "c", Cin, Cout, k, mode == 2 && out2 == nullptr);
    op.build(B, H, W, reinterpret_cast<char*>(ws));
    // parameter block: [weight | bias] as declared
    float* pbuf = nullptr; float* gbuf = nullptr;
    PA_CHECK(hipMallocAsync((void**)&pbuf, n.n_params * sizeof(float) * 2 + 64, n.st));
    gbuf = pbuf + ((n.n_params + 3) & ~(size_t)3);
    PA_CHECK(hipMemsetAsync(pbuf, 0, n.n_params * sizeof(float) * 2 + 64, n.st));
    PA_CHECK(hipMemcpyAsync(pbuf + op.c.p_w, w, (size_t)Cout * Cin * k * k * sizeof(float), hipMemcpyDeviceToDevice, n.st));
    if (bias) PA_CHECK(hipMemcpyAsync(pbuf + op.c.p_b, bias, Cout * sizeof(float), hipMemcpyDeviceToDevice, n.st));
    n.params = pbuf; n.grads = gbuf; n.buffers = pbuf;
    TRY(n.upload_tables());
    TRY(n.prepare_weights());
    n.train_bn = false;
    if (mode == 0) {
        TRY(pa_launch_nchw_f32_to_nhwc_bf16(a_in, op.xin, B, Cin, H, W, n.st));
        TRY(n.conv_fwd(op.c, pa_plain(op.xin), B, H, W, pa_none(), pa_none(), op.yout, nullptr));
        TRY(pa_launch_nhwc_bf16_to_nchw_f32(pa_plain(op.yout), out, B, Cout, H, W, n.st));
    } else if (mode == 1) {
        PaEpilogue ep; memset(&ep, 0, sizeof ep); ep.mode = PA_OUT_PLAIN;
        TRY(pa_launch_nchw_f32_to_nhwc_bf16(a_in, op.yout, B, Cout, H, W, n.st));
        TRY(n.conv_dgrad(op.c, pa_plain(op.yout), B, H, W, pa_none(), pa_none(), ep, op.xin));
        TRY(pa_launch_nhwc_bf16_to_nchw_f32(pa_plain(op.xin), out, B, Cin, H, W, n.st));
    } else {
        TRY(pa_launch_nchw_f32_to_nhwc_bf16(a_in, op.yout, B, Cout, H, W, n.st));
        TRY(pa_launch_nchw_f32_to_nhwc_bf16(b_in, op.xin, B, Cin, H, W, n.st));
        TRY(n.conv_wgrad(op.c, pa_plain(op.yout), pa_plain(op.xin), B, H, W));
        TRY(n.reduce_grads());
        PA_CHECK(hipMemcpyAsync(out, gbuf + op.c.p_w, (size_t)Cout * Cin * k * k * sizeof(float), hipMemcpyDeviceToDevice, n.st));
        if (out2) PA_CHECK(hipMemcpyAsync(out2, gbuf + op.c.p_b, Cout * sizeof(float), hipMemcpyDeviceToDevice, n.st));
    }
    PA_CHECK(hipStreamSynchronize(n.st));
    PA_CHECK(hipFree(pbuf));
    return 0;
}

// ---------------------------------------------------------------------------- grouped weight gradients
// n <= 8 INDEPENDENT weight gradients (the three or four of a residual block, reference models/asn_stacked_hg.py:17-24) as the training
// step submits them: ONE wgrad_group_kernel launch (conv_wgrad_tile.hip) + the slab reduction -- or, for comparison, the same layers
// launched one by one.  The operator-level door to the group kernel for the parity tests (tests/test_gpu_conv.py).
struct WgradGroupOp {
    Net n; ConvLayer c[PA_WG_GROUP_MAX]; bf16 *dy[PA_WG_GROUP_MAX], *dyq[PA_WG_GROUP_MAX], *x[PA_WG_GROUP_MAX];
    float *pbuf = nullptr, *gbuf = nullptr;
    size_t build(const pa_wgrad_job* jobs, int nj, char* base) {
        Arena a; a.base = base;
        n.prep_jobs = a.get<PaPrepJob>(nj); n.red_jobs = a.get<PaWgradReduceJob>(nj); n.bneval_jobs = a.get<PaBnEvalJob>(1);
        for (int j = 0; j < nj; ++j) {
            const pa_wgrad_job& q = jobs[j];
            n.B = q.B;                                      // (layout_conv picks the grouped split count for M == B * H * W)
            n.layout_conv(c[j], a, q.B * q.H * q.W, q.H, q.W);
            const size_t M = (size_t)q.B * q.H * q.W;
            dy[j] = a.get<bf16>(M * c[j].pcout);
            dyq[j] = q.dy_q ? a.get<bf16>(M * c[j].pcout) : nullptr;
            x[j] = a.get<bf16>(M * c[j].pcin);
        }
        pbuf = a.get<float>(n.n_params + 8);
        gbuf = a.get<float>(n.n_params + 8);
        n.layout_shared(a);
        a.take(0);
        return a.off;
    }
    int declare(const pa_wgrad_job* jobs, int nj) {
        if (nj < 1 || nj > PA_WG_GROUP_MAX) { pa_set_error_msg("pa_wgrad_group: 1 .. 8 jobs"); return 1; }
        n.is_agent = true;
        for (int j = 0; j < nj; ++j) {
            const pa_wgrad_job& q = jobs[j];
            if (q.Cin % 64 || q.Cout % 64 || (q.k != 1 && q.k != 3) || q.B < 1 || q.H < 1 || q.W < 1) { pa_set_error_msg("pa_wgrad_group: channels % 64 == 0, k in {1,3}"); return 1; }
            if ((q.dy_q == nullptr) != (q.dy_k == nullptr)) { pa_set_error_msg("pa_wgrad_group: dy_q and dy_k come together"); return 1; }
            char name[16]; snprintf(name, sizeof name, "j%d", j);
            n.declare_conv(c[j], name, q.Cin, q.Cout, q.k, q.db == nullptr);
        }
        return 0;
    }
};

size_t pa_wgrad_group_workspace_bytes(const pa_wgrad_job* jobs, int njobs) {
    g_err[0] = 0;
    WgradGroupOp op;
    if (op.declare(jobs, njobs)) return 0;
    return op.build(jobs, njobs, nullptr);
}

int pa_wgrad_group(const pa_wgrad_job* jobs, int njobs, int mode, void* ws, void* s) {
    g_err[0] = 0;
    WgradGroupOp op; Net& n = op.n;
    TRY(op.declare(jobs, njobs));
    n.st = ST(s);
    op.build(jobs, njobs, reinterpret_cast<char*>(ws));
    PA_CHECK(hipMemsetAsync(op.gbuf, 0, (n.n_params + 8) * sizeof(float), n.st));
    n.params = op.pbuf; n.grads = op.gbuf; n.buffers = op.pbuf;
    TRY(n.upload_tables());
    PaWgradArgs args[PA_WG_GROUP_MAX]; const PaWgradArgs* ptrs[PA_WG_GROUP_MAX];
    for (int j = 0; j < njobs; ++j) {
        const pa_wgrad_job& q = jobs[j];
        ConvLayer& c = op.c[j];
        TRY(pa_launch_nchw_f32_to_nhwc_bf16(q.dy, op.dy[j], q.B, q.Cout, q.H, q.W, n.st));
        if (q.dy_q) TRY(pa_launch_nchw_f32_to_nhwc_bf16(q.dy_q, op.dyq[j], q.B, q.Cout, q.H, q.W, n.st));
        TRY(pa_launch_nchw_f32_to_nhwc_bf16(q.x, op.x[j], q.B, q.Cin, q.H, q.W, n.st));
        PaWgradArgs& a = args[j]; memset(&a, 0, sizeof a);
        a.dy = pa_plain(op.dy[j]);
        if (q.dy_q) { a.dy.mode = PA_LD_LIN2; a.dy.q = op.dyq[j]; a.dy.k0 = q.dy_k; a.dy.k1 = q.dy_k + q.Cout; a.dy.k2 = q.dy_k + 2 * q.Cout; }
        a.x = pa_plain(op.x[j]);
        if (q.x_k) { a.x.mode = PA_LD_BNRELU; a.x.k0 = q.x_k; a.x.k1 = q.x_k + q.Cin; }
        a.part = c.part; a.dbpart = c.dbpart;
        a.B = q.B; a.H = q.H; a.W = q.W; a.Cin = c.pcin; a.Cout = c.pcout; a.taps = c.taps(); a.splits = c.splits;
        ptrs[j] = &a;
    }
    if (mode == 0) {
        for (int j = 0; j < njobs; ++j) TRY(pa_launch_wgrad(args[j], n.st));
    } else {
        for (int j = 0; j < njobs; ++j)
            if (!pa_wgrad_group_takes(args[j])) { pa_set_error_msg("pa_wgrad_group: a job the grouped kernel does not take (shape / operand modes)"); return 2; }
        TRY(pa_launch_wgrad_group(ptrs, njobs, n.st, mode == 2));
    }
    TRY(pa_launch_wgrad_reduce(n.red_jobs, n.n_red, n.red_max, n.st));
    for (int j = 0; j < njobs; ++j) {
        const pa_wgrad_job& q = jobs[j];
        PA_CHECK(hipMemcpyAsync(q.dw, op.gbuf + op.c[j].p_w, (size_t)q.Cout * q.Cin * q.k * q.k * sizeof(float), hipMemcpyDeviceToDevice, n.st));
        if (q.db) PA_CHECK(hipMemcpyAsync(q.db, op.gbuf + op.c[j].p_b, q.Cout * sizeof(float), hipMemcpyDeviceToDevice, n.st));
    }
    PA_CHECK(hipStreamSynchronize(n.st));
    return 0;
}

// micro-benchmark of ONE conv launch (tools/bench_conv.py): variant bits select what the launch does
//   bit0: input transform BNRELU (fwd) / LIN2 (dgrad, wgrad dy)      bit1: epilogue STATS (fwd) / BWD (dgrad)
//   bit2: one residual addend                                         bit3 (wgrad): x operand BNRELU
// Returns the average milliseconds per launch over `iters` launches (HIP events on the stream).
int pa_conv2d_time(int mode, int variant, int B, int Cin, int Cout, int H, int W, int k, int iters, void* ws, float* ms_out, void* s) {
    g_err[0] = 0;
    ConvOp op; Net& n = op.n; n.is_agent = true; n.B = B; n.st = ST(s);
    n.declare_conv(op.c, "c", Cin, Cout, k, false);
    BNLayer bn_in, bn_out;
    n.declare_bn(bn_in, "bi", mode == 0 ? Cin : Cout);
    n.declare_bn(bn_out, "bo", mode == 0 ? Cout : Cin);
    Arena a; a.base = reinterpret_cast<char*>(ws);
    a.off = op.build(B, H, W, a.base);
    for (BNLayer* b : n.bns) n.layout_bn(*b, a, B * H * W);
    const size_t M = (size_t)B * H * W;
    bf16* extra = a.get<bf16>(M * (Cin > Cout ? Cin : Cout));      // residual addend / xref / second LIN2 operand
    bf16* extra2 = a.get<bf16>(M * (Cin > Cout ? Cin : Cout));
    float* pbuf = a.get<float>(2 * (n.n_params + 8));
    n.params = pbuf; n.grads = pbuf + n.n_params + 8; n.buffers = pbuf;
    TRY(n.upload_tables());
    auto mk = [&](const bf16* p, BNLayer& b, int md) { PaOperand o = pa_plain(p); o.mode = md; o.q = extra2; o.k0 = b.scale; o.k1 = b.shift; o.k2 = b.kA; return o; };
    hipEvent_t e0, e1;
    PA_CHECK(hipEventCreate(&e0)); PA_CHECK(hipEventCreate(&e1));
    auto once = [&]() -> int {
        if (mode == 0) {
            PaConvArgs c; memset(&c, 0, sizeof c);
            c.in = mk(op.xin, bn_in, (variant & 1) ? PA_LD_BNRELU : PA_LD_PLAIN); c.w = op.c.wf; c.bias = pbuf; c.out = op.yout;
            c.add1 = (variant & 4) ? pa_plain(extra) : pa_none(); c.add2 = pa_none();
            c.B = B; c.H = H; c.W = W; c.Cin = Cin; c.Cout = Cout; c.taps = k * k;
            c.ep.mode = (variant & 2) ? PA_OUT_STATS : PA_OUT_PLAIN; c.ep.stats = bn_out.stats;
            return pa_launch_conv(c, n.st);
        } else if (mode == 1) {
            PaConvArgs c; memset(&c, 0, sizeof c);
            c.in = mk(op.yout, bn_in, (variant & 1) ? PA_LD_LIN2 : PA_LD_PLAIN); c.w = op.c.wb; c.out = op.xin;
            c.add1 = (variant & 4) ? pa_plain(extra) : pa_none(); c.add2 = pa_none();
            c.B = B; c.H = H; c.W = W; c.Cin = Cout; c.Cout = Cin; c.taps = k * k;
            c.ep.mode = (variant & 2) ? PA_OUT_BWD : PA_OUT_PLAIN; c.ep.stats = bn_out.bstats; c.ep.xref = extra;
            c.ep.scale = bn_out.scale; c.ep.shift = bn_out.shift; c.ep.mean = bn_out.mean; c.ep.invstd = bn_out.invstd;
            return pa_launch_conv(c, n.st);
        } else {
            PaWgradArgs g; memset(&g, 0, sizeof g);
            g.dy = mk(op.yout, bn_in, (variant & 1) ? PA_LD_LIN2 : PA_LD_PLAIN);
            g.x = mk(op.xin, bn_out, (variant & 8) ? PA_LD_BNRELU : PA_LD_PLAIN);
            g.part = op.c.part; g.dbpart = nullptr; g.B = B; g.H = H; g.W = W; g.Cin = Cin; g.Cout = Cout; g.taps = k * k; g.splits = op.c.splits;
            return pa_launch_wgrad(g, n.st);
        }
    };
    for (int i = 0; i < 3; ++i) TRY(once());
    float ms = 0.f;
    if (variant & 16) {            // COLD: 640 MB of other traffic between launches (the Infinity Cache holds 256 MB), one event pair per launch
        char* scratch = a.get<char>((size_t)640 << 20);
        for (int i = 0; i < iters; ++i) {
            PA_CHECK(hipMemsetAsync(scratch, i & 1, (size_t)640 << 20, n.st));
            PA_CHECK(hipEventRecord(e0, n.st));
            TRY(once());
            PA_CHECK(hipEventRecord(e1, n.st));
            PA_CHECK(hipEventSynchronize(e1));
            float t = 0.f;
            PA_CHECK(hipEventElapsedTime(&t, e0, e1));
            ms += t;
        }
    } else {
        PA_CHECK(hipEventRecord(e0, n.st));
        for (int i = 0; i < iters; ++i) TRY(once());
        PA_CHECK(hipEventRecord(e1, n.st));
        PA_CHECK(hipEventSynchronize(e1));
        PA_CHECK(hipEventElapsedTime(&ms, e0, e1));
    }
    *ms_out = ms / iters;
    hipEventDestroy(e0); hipEventDestroy(e1);
    return 0;
}

// ---------------------------------------------------------------------------- networks
pa_net* pa_hg_create(int num_stacks, int num_classes, int chan, int B, int res) {
    g_err[0] = 0;
    if (num_classes != 16 || chan % 128 != 0 || chan > 2048 || res % 64 != 0 || B < 1 || num_stacks < 1) {
        pa_set_error_msg("pa_hg_create: need num_classes == 16, chan % 128 == 0, res % 64 == 0");
        return nullptr;
    }
    pa_net* p = new (std::nothrow) pa_net;
    if (!p) return nullptr;
    Net& n = p->n;
    n.stacks = num_stacks; n.classes = num_classes; n.chan = chan; n.B = B; n.res = res;
    n.immediate_reduce = pa_getenv("PA_SHARED_SLAB") != nullptr;
    n.declare_pose();
    n.workspace_bytes = n.layout_all(nullptr);
    return p;
}

pa_net* pa_asn_create(int chan, int scale_num, int rotation_num, int B, int res) {
    g_err[0] = 0;
    if (chan % 128 != 0 || chan > 512 || res % 64 != 0 || B < 1 || scale_num < 1 || rotation_num < 1 || scale_num + rotation_num > 64) {
        pa_set_error_msg("pa_asn_create: need chan % 128 == 0, res % 64 == 0, scale_num + rotation_num <= 64");
        return nullptr;
    }
    pa_net* p = new (std::nothrow) pa_net;
    if (!p) return nullptr;
    Net& n = p->n;
    n.chan = chan; n.B = B; n.res = res; n.scale_num = scale_num; n.rot_num = rotation_num; n.stacks = 0;
    n.immediate_reduce = pa_getenv("PA_SHARED_SLAB") != nullptr;
    n.declare_asn();
    n.workspace_bytes = n.layout_asn(nullptr);
    return p;
}

pa_net* pa_asn_create_dropout(int chan, int B, int res) {
    g_err[0] = 0;
    if (chan % 128 != 0 || chan > 512 || res != 256 || B < 1) {      // the 4x4 cell mask is the neck map of a 256 input (reference :79-100)
        pa_set_error_msg("pa_asn_create_dropout: need chan % 128 == 0 and res == 256 (4x4 neck map)");
        return nullptr;
    }
    pa_net* p = new (std::nothrow) pa_net;
    if (!p) return nullptr;
    Net& n = p->n;
    n.chan = chan; n.B = B; n.res = res; n.stacks = 0; n.asn_dropout = true;
    n.immediate_reduce = pa_getenv("PA_SHARED_SLAB") != nullptr;
    n.declare_asn();
    n.workspace_bytes = n.layout_asn(nullptr);
    return p;
}

static int check_agent_pair(const pa_net* asn, const pa_net* pose, const char* who) {
    if (!asn->n.is_agent || pose->n.is_agent || asn->n.B != pose->n.B || asn->n.chan != pose->n.chan || asn->n.res != pose->n.res) {
        char buf[160]; snprintf(buf, sizeof buf, "%s: agent and pose net must be built for the same batch, width and resolution", who);
        pa_set_error_msg(buf);
        return 1;
    }
    return 0;
}

int pa_asn_forward_masks(pa_net* asn, pa_net* pose, int train, float* mask_logits) {
    g_err[0] = 0;
    TRY(check_agent_pair(asn, pose, "pa_asn_forward_masks"));
    asn->n.bn_update = (train == 2) ? 0 : 1;
    const int r = asn->n.asn_forward_masks(pose->n, train != 0, mask_logits);
    asn->n.bn_update = 1;
    return r;
}

int pa_asn_backward_masks(pa_net* asn, pa_net* pose, const float* dlogits) {
    g_err[0] = 0;
    TRY(check_agent_pair(asn, pose, "pa_asn_backward_masks"));
    if (!dlogits) { pa_set_error_msg("pa_asn_backward_masks: dlogits == NULL"); return 1; }
    return asn->n.asn_backward_masks(pose->n, dlogits);
}

int pa_hg_set_dropout_masks(pa_net* net, const float* masks) {
    g_err[0] = 0;
    if (net->n.is_agent) { pa_set_error_msg("pa_hg_set_dropout_masks: not a pose net"); return 1; }
    if (masks && net->n.res != 256) { pa_set_error_msg("pa_hg_set_dropout_masks: the 4x4 cell mask needs res == 256 (4x4 neck map)"); return 1; }
    net->n.drop_mask = masks;
    return 0;
}

int pa_hg_forward_half(pa_net* net, const float* img, const void* img4, int train) {
    g_err[0] = 0;
    if (!img && !img4) { pa_set_error_msg("pa_hg_forward_half: need img or img4"); return 1; }
    net->n.bn_update = (train == 2) ? 0 : 1;        // train == 2: batch statistics, running estimates untouched
    const int r = net->n.forward_half(img, reinterpret_cast<const bf16*>(img4), train != 0);
    net->n.bn_update = 1;
    return r;
}

int pa_asn_forward(pa_net* asn, pa_net* pose, int train, float* logits_scale, float* logits_rot) {
    g_err[0] = 0;
    if (!asn->n.is_agent || pose->n.is_agent || asn->n.B != pose->n.B || asn->n.chan != pose->n.chan || asn->n.res != pose->n.res) {
        pa_set_error_msg("pa_asn_forward: agent and pose net must be built for the same batch, width and resolution");
        return 1;
    }
    asn->n.bn_update = (train == 2) ? 0 : 1;       // train == 2: batch statistics, running estimates untouched
    TRY(asn->n.asn_forward(pose->n, train != 0, logits_scale, logits_rot));
    asn->n.bn_update = 1;
    return 0;
}

int pa_asn_backward(pa_net* asn, pa_net* pose, const float* target_scale, const float* target_rot, float* loss) {
    g_err[0] = 0;
    TRY(asn->n.asn_backward(pose->n, target_scale, target_rot, loss));
    return 0;
}

const float* pa_asn_probs(const pa_net* asn) { return asn->n.asn_probs; }
int pa_asn_set_log_eps(pa_net* asn, float eps) { asn->n.asn_log_eps = eps; return 0; }

void pa_net_destroy(pa_net* net) { if (net) { net->n.release_streams(); delete net; } }

int pa_net_num_tensors(const pa_net* net) { return (int)net->n.tensors.size(); }

int pa_net_tensor_info(const pa_net* net, int i, char* name, int name_cap, int* shape4, int* ndim, size_t* offset, size_t* numel,
                       int* kind) {
    if (i < 0 || i >= (int)net->n.tensors.size()) { pa_set_error_msg("pa_net_tensor_info: index out of range"); return 1; }
    const TensorInfo& t = net->n.tensors[i];
    snprintf(name, name_cap, "%s", t.name.c_str());
    for (int k = 0; k < 4; ++k) shape4[k] = t.shape[k];
    *ndim = t.ndim; *offset = t.offset; *numel = t.numel; *kind = t.is_buffer;
    return 0;
}

size_t pa_net_param_floats(const pa_net* net) { return net->n.n_params; }
size_t pa_net_buffer_floats(const pa_net* net) { return net->n.n_buffers; }
size_t pa_net_workspace_bytes(const pa_net* net) { return net->n.workspace_bytes; }

int pa_net_bind(pa_net* net, float* params, float* grads, float* buffers, void* workspace, void* s) {
    g_err[0] = 0;
    Net& n = net->n;
    n.release_graph();                    // (a captured step holds the old pointers)
    n.params = params; n.grads = grads; n.buffers = buffers; n.workspace = reinterpret_cast<char*>(workspace); n.st = ST(s);
    const size_t bytes = n.is_agent ? n.layout_asn(n.workspace) : n.layout_all(n.workspace);
    // the layout relies on zeros in a few places (padding channels of the 16-channel head tensors, statistic accumulators):
    // clear the whole workspace here, once, so that the caller does not have to
    PA_CHECK(hipMemsetAsync(n.workspace, 0, bytes, n.st));
    TRY(n.upload_tables());
    return n.prepare_weights();
}

int pa_net_prepare_weights(pa_net* net) { g_err[0] = 0; TRY(net->n.prepare_weights()); return 0; }

int pa_hg_forward(pa_net* net, const float* img, const void* img4, const double* pts, int train, float* loss_per_stack) {
    g_err[0] = 0;
    if (!img && !img4) { pa_set_error_msg("pa_hg_forward: need img or img4"); return 1; }
    TRY(net->n.forward_pose(img, reinterpret_cast<const bf16*>(img4), pts, train != 0, loss_per_stack));
    return 0;
}

const float* pa_hg_heatmap_nhwc(const pa_net* net, int stack) { return net->n.heat[stack]; }

int pa_hg_heatmap_nchw(pa_net* net, int stack, float* out) {
    Net& n = net->n;
    TRY(pa_launch_nhwc_to_nchw_f32(n.heat[stack], out, n.B, n.res / 4, n.res / 4, 16, n.st));
    return 0;
}

int pa_hg_backward(pa_net* net) { g_err[0] = 0; TRY(net->n.backward_pose()); return 0; }

int pa_hg_backward_phase(pa_net* net, int phase) {
    g_err[0] = 0;
    Net& n = net->n;
    if (phase < 0 || phase > n.stacks) { pa_set_error_msg("pa_hg_backward_phase: phase out of range"); return 1; }
    TRY(n.ensure_streams());
    if (phase == n.stacks) return n.backward_stem();
    return n.backward_stack(n.stacks - 1 - phase);
}

int pa_hg_bucket_range(const pa_net* net, int stack, size_t* lo, size_t* hi) {
    const Net& n = net->n;
    char prefix[32]; snprintf(prefix, sizeof prefix, "hg.%d.", stack);
    size_t a = (size_t)-1, b = 0;
    for (const TensorInfo& t : n.tensors)
        if (t.is_buffer == 0 && t.name.compare(0, strlen(prefix), prefix) == 0) { if (t.offset < a) a = t.offset; if (t.offset + t.numel > b) b = t.offset + t.numel; }
    if (b == 0) { pa_set_error_msg("pa_hg_bucket_range: no such stack"); return 1; }
    // the caller all-reduces [lo, hi) EARLY and skips it at the end: no other parameter may live inside the range (this holds
    // by the declaration order of declare_pose; a parameter interleaved later would be reduced before its gradient is final)
    for (const TensorInfo& t : n.tensors)
        if (t.is_buffer == 0 && t.name.compare(0, strlen(prefix), prefix) != 0 && t.offset < b && t.offset + t.numel > a) {
            pa_set_error_msg("pa_hg_bucket_range: a parameter of another module lies inside the stack's range");
            return 1;
        }
    *lo = a; *hi = b;
    return 0;
}

int pa_hg_bucket_wait(pa_net* net, int stack, void* stream) {
    g_err[0] = 0;
    Net& n = net->n;
    const int r = n.mark_bucket(stack);
    if (r < 0) return -1;                      // no early bucket in this stream mode: exchange everything at the end
    if (r > 0) return r;
    PA_CHECK(hipStreamWaitEvent(ST(stream), n.ev_bucket[stack], 0));
    return 0;
}

int pa_hg_train_step(pa_net* net, const void* img4, const double* pts, int train, int use_graph, float* loss_per_stack) {
    g_err[0] = 0;
    Net& n = net->n;
    if (!img4 || !pts) { pa_set_error_msg("pa_hg_train_step: img4 and pts are required"); return 1; }
    if (!use_graph) {
        TRY(n.forward_pose(nullptr, reinterpret_cast<const bf16*>(img4), pts, train != 0, loss_per_stack));
        return n.backward_pose();
    }
    // the graph reads the engine's own input buffers: bring the caller's tensors there first (12.6 MB + 6 KB at B = 24)
    if (img4 != n.img4) PA_CHECK(hipMemcpyAsync(n.img4, img4, (size_t)n.B * n.res * n.res * 4 * sizeof(bf16), hipMemcpyDeviceToDevice, n.st));
    if (pts != n.pts_dev) PA_CHECK(hipMemcpyAsync(n.pts_dev, pts, (size_t)n.B * n.classes * 2 * sizeof(double), hipMemcpyDeviceToDevice, n.st));
    TRY(n.train_step_graph(train != 0));
    if (loss_per_stack) PA_CHECK(hipMemcpyAsync(loss_per_stack, n.loss_keep, n.stacks * sizeof(float), hipMemcpyDeviceToDevice, n.st));
    if (n.loss_total_out) PA_CHECK(hipMemcpyAsync(n.loss_total_out, n.loss_keep + n.stacks, sizeof(float), hipMemcpyDeviceToDevice, n.st));
    return 0;
}

// total_dev: a device float that every later pa_hg_forward / pa_hg_train_step with `pts` also writes the SUM of the per-stack losses to
// (NULL: off).  The reference sums them on the host side of autograd (stack-hg.py:156-159); here the caller gets the scalar without a
// reduction launch of its own between the backward pass and the optimizer.
int pa_hg_set_loss_total(pa_net* net, float* total_dev) {
    g_err[0] = 0;
    if (net->n.loss_total_out != total_dev) { net->n.loss_total_out = total_dev; net->n.release_graph(); }      // (a captured step holds the pointer by value)
    return 0;
}

int pa_hg_accuracy(pa_net* net, int stack, const int32_t* idxs, int nidx, float* acc, float* scratch) {
    g_err[0] = 0;
    Net& n = net->n;
    const int H = n.res / 4, J = 16, B = n.B;
    float* tgt = scratch;                         // [B][16][H][H]
    float* gp = tgt + (size_t)B * J * H * H + (size_t)B * J * 2;      // [B][16][2]
    float* norm = gp + (size_t)B * J * 2;         // [B]
    const float* pp = nullptr;                    // [B][16][2] arg-max of the heat maps (shared with pa_hg_pckh)
    hipStream_t ms;
    TRY(n.meter_stream(&ms));
    TRY(n.heat_argmax(stack, &pp, ms));
    TRY(pa_launch_gaussian_heatmap(n.pts_dev, tgt, B, J, H, H, ms));
    TRY(pa_launch_argmax(tgt, (long)J * H * H, (long)H * H, 1, B, J, H, H, gp, nullptr, ms));
    TRY(pa_launch_fill(norm, (float)H / 10.f, B, ms));
    TRY(pa_launch_pck(pp, gp, norm, 1.f, idxs, nidx, 0.5f, nullptr, B, J, acc, nullptr, nullptr, ms));
    return n.meter_done(ms);
}

// pa_net_meters_async (include/poseadv.h): on != 0 -- pa_hg_accuracy / pa_hg_pckh calls are launched on the engine's meter stream, behind
// everything the main stream holds at the call and beside whatever it is given next; the main stream joins them at the end of the next
// pa_hg_backward / the last pa_hg_backward_phase (or in front of the next forward pass).  Until then their outputs and scratch buffers
// must stay allocated and unread.
int pa_net_meters_async(pa_net* net, int on) { g_err[0] = 0; net->n.meters_async = on != 0; return 0; }


// Evaluation.accuracy_origin_res (pylib/Evaluation.py:77-97) and per_person_pckh (:99-167) of stack i's
// heat maps, straight from the engine's NHWC fp32 maps (no layout copy, no host sync)
int pa_hg_pckh(pa_net* net, int stack, const float* center, const float* scale, const float* rot, const float* gt_pts,
               const float* norm, const int32_t* idxs, int nidx, float* acc, float* person, float* scratch) {
    g_err[0] = 0;
    Net& n = net->n;
    const int H = n.res / 4, J = 16, B = n.B;
    const float* pp = nullptr;                    // [B][16][2] arg-max (computed once per forward, shared with pa_hg_accuracy)
    hipStream_t ms;
    TRY(n.meter_stream(&ms));
    TRY(n.heat_argmax(stack, &pp, ms));
    float* fp = scratch + (size_t)B * J * 2;      // [B][16][2] back-projected predictions
    float* vis = fp + (size_t)B * J * 2;          // [B][16][2] arg-max of the augmented target
    float* tgt = vis + (size_t)B * J * 2;         // [B][16][H][H] (only when person != NULL)
    TRY(pa_launch_final_preds(n.heat[stack], (long)H * H * 16, 1, 16, pp, center, scale, rot, B, J, H, H, fp, ms));
    if (person) {
        TRY(pa_launch_gaussian_heatmap(n.pts_dev, tgt, B, J, H, H, ms));
        TRY(pa_launch_argmax(tgt, (long)J * H * H, (long)H * H, 1, B, J, H, H, vis, nullptr, ms));
    }
    TRY(pa_launch_pck(fp, gt_pts, norm, 0.f, idxs, nidx, 0.5f, person ? vis : nullptr, B, J, acc, person, nullptr, ms));
    return n.meter_done(ms);
}

// per-launch HIP-event timing of the MFMA kernels (bench.py roofline): enable, run steps, then report.
int pa_net_profile_begin(pa_net* net) { net->n.prof.used = 0; net->n.prof.on = true; return 0; }
// out[cap_classes][4] = {total ms, launches, algorithmic bytes, flops} per class (include/poseadv.h).  Reporting synchronises and disables.
int pa_net_profile_report(pa_net* net, double* out, int cap_classes, int* n_classes) {
    net->n.prof.on = false;
    double full[PA_PROF_NCLS * 4];
    int r = net->n.prof.report(full);
    if (r) { pa_set_error("profile report", (hipError_t)r, __FILE__, __LINE__); return r; }
    for (int c = 0; c < PA_PROF_NCLS && c < cap_classes; ++c)
        for (int j = 0; j < 4; ++j) out[c * 4 + j] = full[c * 4 + j];
    if (n_classes) *n_classes = PA_PROF_NCLS;
    return 0;
}
int pa_net_set_fin_prologue(pa_net* net, int max_rows) {
    net->n.fin_rows_max = max_rows < 0 ? 0 : (max_rows > PA_FIN_SMALL_ROWS ? PA_FIN_SMALL_ROWS : max_rows);
    net->n.release_graph();
    return 0;
}
int pa_net_design_bytes(const pa_net* net, double* out) { out[0] = net->n.dbytes_rd; out[1] = net->n.dbytes_wr; return 0; }

int pa_net_profile_classes(const pa_net* net, int32_t* out, int cap) {
    const std::vector<int>& q = net->n.prof.last_seq;
    for (int i = 0; i < cap && i < (int)q.size(); ++i) out[i] = q[i];
    return (int)q.size();
}

int pa_net_set_multi_stream(pa_net* net, int on) {
    if (net->n.ensure_streams()) return 1;
    net->n.release_graph();
    net->n.multi_stream = on != 0 && net->n.side[0] != nullptr;
    return 0;
}

// debug / test hook: copy an internal activation (BatchNorm+ReLU applied) or its gradient buffer out as
// NCHW fp32.  which: "stem", "res1", "pool0", "res2", "res3", "hg<i>.skip<k>", "hg<i>.pool<k>",
// "hg<i>.down<k>", "hg<i>.neck", "hg<i>.up<k>", "hg<i>.merge<k>", "post<i>", "lin<i>", "xin<i>" (k = 1..4)
// and "<name>.x1"/".x2" for the inner tensors of a residual block; "hg<i>.maskedskip<k>", "hg<i>.maskedneck" (occlusion branch).
// grad != 0 -> the raw gradient buffer.
int pa_hg_debug_tensor(pa_net* net, const char* which, int grad, float* out, int* shape4) {
    Net& n = net->n;
    std::string w(which);
    const Act* a = nullptr;
    auto pick_res = [&](Residual& r, const std::string& rest) -> const Act* {
        if (rest == ".x1") return &r.x1;
        if (rest == ".x2") return &r.x2;
        return &r.x3;
    };
    auto starts = [&](const char* p) { return w.rfind(p, 0) == 0; };
    if (starts("stem")) a = &n.a0;
    else if (starts("res1")) a = pick_res(n.res1, w.substr(4));
    else if (starts("pool0")) a = &n.pool0;
    else if (starts("res2")) a = pick_res(n.res2, w.substr(4));
    else if (starts("res3")) a = pick_res(n.res3, w.substr(4));
    else if (starts("post")) { int i = w[4] - '0'; a = pick_res(n.post[i], w.substr(5)); }
    else if (starts("lin")) { int i = w[3] - '0'; a = &n.lin_out[i]; }
    else if (starts("xin")) { int i = w[3] - '0'; a = &n.xin[i]; }
    else if (starts("hg")) {
        int i = w[2] - '0';
        std::string r = w.substr(4);
        Hourglass& h = n.hg[i];
        if (r.rfind("neck", 0) == 0) a = pick_res(h.neck, r.substr(4));
        else if (r == "maskedneck") a = &h.neckm;                                  // occlusion branch: neck * cell mask
        else if (r.rfind("maskedskip", 0) == 0) a = &h.skipm[r[10] - '1'];
        else {
            int k = r[r.find_first_of("1234")] - '1';
            std::string rest = r.substr(r.find_first_of("1234") + 1);
            if (r.rfind("skip", 0) == 0) a = pick_res(h.skip[k], rest);
            else if (r.rfind("down", 0) == 0) a = pick_res(h.down[k], rest);
            else if (r.rfind("up", 0) == 0) a = pick_res(h.up[k], rest);
            else if (r.rfind("pool", 0) == 0) a = &h.pooled[k];
            else if (r.rfind("merge", 0) == 0) a = &h.merged[k];
        }
    }
    if (!a) { pa_set_error_msg("pa_hg_debug_tensor: unknown tensor name"); return 1; }
    shape4[0] = a->B; shape4[1] = a->C; shape4[2] = a->H; shape4[3] = a->W;
    if (!out) return 0;
    PaOperand src = grad ? pa_plain(a->grad) : n.op(*a);
    TRY(pa_launch_nhwc_bf16_to_nchw_f32(src, out, a->B, a->C, a->H, a->W, n.st));
    return 0;
}


// test hook for the agent, like pa_hg_debug_tensor: names "in<k>" (k=0..4: residual_skip1-4, residual_neck),
// "pa<k>", "merge<k>" (k=0..3), "deep<k>" (k=0..2), optional ".x1"/".x2"
int pa_asn_debug_tensor(pa_net* net, const char* which, int grad, float* out, int* shape4) {
    Net& n = net->n;
    std::string w(which);
    const Act* a = nullptr;
    auto pick = [&](Residual& r, const std::string& rest) -> const Act* {
        if (rest == ".x1") return &r.x1;
        if (rest == ".x2") return &r.x2;
        return &r.x3;
    };
    auto starts = [&](const char* p) { return w.rfind(p, 0) == 0; };
    if (starts("in")) a = pick(n.asn_in[w[2] - '0'], w.substr(3));
    else if (starts("pa")) a = &n.asn_pa[w[2] - '0'];
    else if (starts("merge")) a = pick(n.asn_merge[w[5] - '0'], w.substr(6));
    else if (starts("deep")) a = pick(n.asn_deep[w[4] - '0'], w.substr(5));
    if (!a) { pa_set_error_msg("pa_asn_debug_tensor: unknown tensor name"); return 1; }
    shape4[0] = a->B; shape4[1] = a->C; shape4[2] = a->H; shape4[3] = a->W;
    if (!out) return 0;
    PaOperand src = grad ? pa_plain(a->grad) : n.op(*a);
    TRY(pa_launch_nhwc_bf16_to_nchw_f32(src, out, a->B, a->C, a->H, a->W, n.st));
    return 0;
}

}  // extern "C"
