// ASN scale/rotation agent (reference models/asn_stacked_hg.py:349-439, aug branch) on the same
// residual-block kernels as the pose net, reading the pose net's DETACHED features in place
// (hg[0].skip1-4 and neck: raw bf16 tensors + their BatchNorm scale/shift; reference :159-164), plus
// the half-hourglass forward that produces those features (:300-304) and the KL loss of
// joint-train-pose-s-r-agent.py:399-407.
#include "net.h"
#include <string.h>

#define TRY(x) do { int _r = (x); if (_r) return _r; } while (0)

static PaEpilogue ep_plain() { PaEpilogue e; memset(&e, 0, sizeof e); e.mode = PA_OUT_PLAIN; return e; }

// ------------------------------------------------------------------------------------------------
// out = maxpool2x2(hi) + lo        (reference :404-415: x = maxpool(x); x = x + skip)
__global__ void pooladd_fwd_kernel(PaOperand hi, PaOperand lo, bf16* out, int B, int H, int W, int C) {
    // H, W: OUTPUT dims; hi is [B][2H][2W][C]
    const int CG = C / 8;
    const size_t total = (size_t)B * H * W * CG;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const int cg = (int)(t % CG);
        const size_t p = t / CG;
        const int x = (int)(p % W);
        const size_t q = p / W;
        const int y = (int)(q % H), b = (int)(q / H);
        const int c = cg * 8;
        float m[8], v[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const size_t idx = (((size_t)b * 2 * H + 2 * y + (k >> 1)) * 2 * W + 2 * x + (k & 1)) * C + c;
            pa_read8(hi, idx, c, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) m[j] = (k == 0) ? v[j] : fmaxf(m[j], v[j]);
        }
        pa_read8(lo, p * C + c, c, v);
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (bf16)(m[j] + v[j]);
        *reinterpret_cast<bf16x8*>(out + p * C + c) = o;
    }
}

static int launch_pooladd_fwd(const PaOperand& hi, const PaOperand& lo, bf16* out, int B, int H, int W, int C, hipStream_t st) {
    size_t total = (size_t)B * H * W * (C / 8);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(pooladd_fwd_kernel, dim3(blocks), dim3(256), 0, st, hi, lo, out, B, H, W, C);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// AvgPool2d(4) over the 4x4 map + fc_scale / fc_rotation (Linear C -> K) + softmax  (reference :430-436;
// softmax: joint-train-pose-s-r-agent.py:252-253).  One workgroup per sample.
__global__ void asn_head_fwd_kernel(PaOperand x, int HW, int C, const float* ws, const float* bs, const float* wr, const float* br,
                                    int Ks, int Kr, float* feat, float* logits, float* probs) {
    extern __shared__ float sm[];          // [C] features, then [Ks+Kr] logits
    const int b = blockIdx.x;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float acc = 0.f;
        for (int p = 0; p < HW; ++p) {
            float v = (float)x.p[((size_t)b * HW + p) * C + c];
            if (x.mode == PA_LD_BNRELU) v = fmaxf(fmaf(x.k0[c], v, x.k1[c]), 0.f);
            acc += v;
        }
        acc /= (float)HW;
        sm[c] = acc;
        feat[(size_t)b * C + c] = acc;
    }
    __syncthreads();
    const int K = Ks + Kr;
    float* lg = sm + C;
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        const float* w = k < Ks ? ws + (size_t)k * C : wr + (size_t)(k - Ks) * C;
        float acc = k < Ks ? bs[k] : br[k - Ks];
        for (int c = 0; c < C; ++c) acc = fmaf(sm[c], w[c], acc);
        lg[k] = acc;
        logits[(size_t)b * K + k] = acc;
    }
    __syncthreads();
    if (threadIdx.x < 2) {
        const int k0 = threadIdx.x == 0 ? 0 : Ks, kn = threadIdx.x == 0 ? Ks : Kr;
        float mx = lg[k0];
        for (int k = 1; k < kn; ++k) mx = fmaxf(mx, lg[k0 + k]);
        float s = 0.f;
        for (int k = 0; k < kn; ++k) s += expf(lg[k0 + k] - mx);
        for (int k = 0; k < kn; ++k) probs[(size_t)b * K + k0 + k] = expf(lg[k0 + k] - mx) / s;
    }
}

// KL loss (joint-train-pose-s-r-agent.py:399-407): sum over the two heads of
//   K * mean_{b,k} t * (log t - log(p + eps)),  p = softmax(logits); eps = 1e-7 in the joint loop, 0 in the agent
//   pre-training (pretrain-s-r-agent.py:177-190 uses LogSoftmax)
// d/dlogits, d/dfeat (spread back over the HW pixels as the gradient of the average pool), loss value.
__global__ void asn_head_bwd_kernel(const float* probs, const float* target_s, const float* target_r, const float* ws, const float* wr,
                                    int Ks, int Kr, int HW, int C, int B, float* dlogits, bf16* dact, float* loss, float leps) {
    extern __shared__ float sm[];          // [Ks+Kr] dlogits
    const int b = blockIdx.x, K = Ks + Kr;
    if (threadIdx.x < 2) {
        const int k0 = threadIdx.x == 0 ? 0 : Ks, kn = threadIdx.x == 0 ? Ks : Kr;
        const float* t = threadIdx.x == 0 ? target_s + (size_t)b * Ks : target_r + (size_t)b * Kr;
        const float* p = probs + (size_t)b * K + k0;
        float l = 0.f, gp = 0.f;
        for (int k = 0; k < kn; ++k) {
            if (t[k] > 0.f) l += t[k] * (logf(t[k]) - logf(p[k] + leps));
            gp += (-t[k] / ((float)B * (p[k] + leps))) * p[k];
        }
        for (int k = 0; k < kn; ++k) {
            const float g = -t[k] / ((float)B * (p[k] + leps));
            const float dz = PA_GRAD_SCALE * (p[k] * (g - gp));     // (fp16 build: scaled gradients, common.h)
            sm[k0 + k] = dz;
            dlogits[(size_t)b * K + k0 + k] = dz;
        }
        atomicAdd(loss, l / (float)B);
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float df = 0.f;
        for (int k = 0; k < Ks; ++k) df = fmaf(sm[k], ws[(size_t)k * C + c], df);
        for (int k = 0; k < Kr; ++k) df = fmaf(sm[Ks + k], wr[(size_t)k * C + c], df);
        const bf16 v = (bf16)(df / (float)HW);
        for (int p = 0; p < HW; ++p) dact[((size_t)b * HW + p) * C + c] = v;
    }
}

// dW[k][c] = sum_b dlogits[b][k] * feat[b][c];  db[k] = sum_b dlogits[b][k]
__global__ void asn_fc_wgrad_kernel(const float* dlogits, const float* feat, int B, int Ks, int Kr, int C, float* dws, float* dbs,
                                    float* dwr, float* dbr) {
    const int K = Ks + Kr;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < K * C + K; e += gridDim.x * blockDim.x) {
        if (e < K * C) {
            const int k = e / C, c = e - k * C;
            float acc = 0.f;
            for (int b = 0; b < B; ++b) acc = fmaf(dlogits[(size_t)b * K + k], feat[(size_t)b * C + c], acc);
            if (k < Ks) dws[(size_t)k * C + c] = acc; else dwr[(size_t)(k - Ks) * C + c] = acc;
        } else {
            const int k = e - K * C;
            float acc = 0.f;
            for (int b = 0; b < B; ++b) acc += dlogits[(size_t)b * K + k];
            if (k < Ks) dbs[k] = acc; else dbr[k - Ks] = acc;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Occlusion agent head (reference :378-379, :437-439): out_conv = Conv2d(C, 1, 1) on the 4x4 map -> one logit per cell.
// One wave per (sample, cell).
__global__ void asn_mask_head_fwd_kernel(PaOperand x, int C, const float* w, const float* bias, float* logits) {
    const size_t row = blockIdx.x;                     // b * HW + cell
    float acc = 0.f;
    for (int c = threadIdx.x; c < C; c += 64) {
        float v = (float)x.p[row * C + c];
        if (x.mode == PA_LD_BNRELU) v = fmaxf(fmaf(x.k0[c], v, x.k1[c]), 0.f);
        acc = fmaf(v, w[c], acc);
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (threadIdx.x == 0) logits[row] = acc + bias[0];
}

// given d(loss)/d(logits) [B*HW]: d/d(activation) (plain, bf16), d/d(out_conv.weight), d/d(out_conv.bias); one workgroup
__global__ void asn_mask_head_bwd_kernel(PaOperand x, const float* dlogits, const float* w, int rows, int C, bf16* dact, float* dw, float* db) {
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float acc = 0.f;
        const float wc = w[c];
        for (int r = 0; r < rows; ++r) {
            float v = (float)x.p[(size_t)r * C + c];
            if (x.mode == PA_LD_BNRELU) v = fmaxf(fmaf(x.k0[c], v, x.k1[c]), 0.f);
            acc = fmaf(dlogits[r], v, acc);
            dact[(size_t)r * C + c] = (bf16)(PA_GRAD_SCALE * dlogits[r] * wc);
        }
        dw[c] = PA_GRAD_SCALE * acc;
    }
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int r = 0; r < rows; ++r) s += dlogits[r];
        db[0] = PA_GRAD_SCALE * s;
    }
}

// ------------------------------------------------------------------------------------------------
void Net::declare_asn() {
    is_agent = true;
    const char* names[5] = {"residual_skip1.", "residual_skip2.", "residual_skip3.", "residual_skip4.", "residual_neck."};
    for (int k = 0; k < 5; ++k) asn_in[k].declare(*this, names[k], chan, chan, false);
    char buf[32];
    for (int k = 0; k < 4; ++k) { snprintf(buf, sizeof buf, "merge%d.", k + 1); asn_merge[k].declare(*this, buf, chan, chan, false); }
    for (int k = 0; k < 3; ++k) { snprintf(buf, sizeof buf, "deep_merge.%d.", k); asn_deep[k].declare(*this, buf, chan, chan, false); }
    if (asn_dropout) {
        p_oc_w = add_param("out_conv.weight", {1, chan, 1, 1});
        p_oc_b = add_param("out_conv.bias", {1});
    } else {
        p_fcs_w = add_param("fc_scale.weight", {scale_num, chan});
        p_fcs_b = add_param("fc_scale.bias", {scale_num});
        p_fcr_w = add_param("fc_rotation.weight", {rot_num, chan});
        p_fcr_b = add_param("fc_rotation.bias", {rot_num});
    }
    n_params = (n_params + 3) & ~(size_t)3;
}

size_t Net::layout_asn(char* base) {
    Arena a; a.base = base;
    loss_dev = a.get<float>(64);
    a.take(0);
    stats_arena = base ? reinterpret_cast<float*>(base) : nullptr;
    stats_arena_floats = a.off / sizeof(float);
    prep_jobs = a.get<PaPrepJob>(convs.size());
    red_jobs = a.get<PaWgradReduceJob>(convs.size());
    bneval_jobs = a.get<PaBnEvalJob>(bns.size());
    const int H = res / 4;                          // skip1 resolution (64 for a 256 input)
    for (int k = 0; k < 5; ++k) asn_in[k].layout(*this, a, B, H >> k, H >> k, true);
    for (int k = 0; k < 4; ++k) {
        asn_pa[k] = new_act(a, B, H >> (k + 1), H >> (k + 1), chan, nullptr, true);
        asn_merge[k].layout(*this, a, B, H >> (k + 1), H >> (k + 1), true);
    }
    for (int k = 0; k < 3; ++k) asn_deep[k].layout(*this, a, B, H >> 4, H >> 4, true);
    asn_feat = a.get<float>((size_t)B * chan);
    asn_logits = a.get<float>((size_t)B * (asn_dropout ? 16 : scale_num + rot_num));
    asn_probs = a.get<float>((size_t)B * (scale_num + rot_num));
    asn_dlogits = a.get<float>((size_t)B * (scale_num + rot_num));
    layout_shared(a);
    a.take(0);
    return a.off;
}

// stem + the down path of hg[0]: everything the agent reads (reference :300-304 with is_half_hg)
int Net::forward_half(const float* img_nchw, const bf16* img4_in, bool train) {
    train_bn = train;
    TRY(ensure_streams());
    TRY(begin_step());
    if (!train) TRY(pa_launch_bn_eval(bneval_jobs, n_bneval, eps, st));
    const bf16* image = img4_in ? img4_in : img4;
    cur_image = image;
    if (!img4_in) TRY(pa_launch_nchw_to_nhwc4(img_nchw, img4, B, res, res, st));
    TRY(conv_fwd(stem_conv, pa_plain(image), B, res / 2, res / 2, pa_none(), pa_none(), a0.raw, &stem_bn));
    TRY(res1.fwd(*this, a0));
    TRY(pa_launch_maxpool_fwd(op(res1.x3), pool0.raw, B, res / 2, res / 2, 128, st));
    TRY(res2.fwd(*this, pool0));
    TRY(res3.fwd(*this, res2.x3));
    TRY(hg[0].encode(*this, xin[0]));
    if (multi_stream)                      // no decoder here: join the skip branches before anyone reads them
        for (int k = 0; k < 4; ++k) if (forks(k)) TRY(wait_join(k));
    return 0;
}

int Net::asn_forward_trunk(Net& pose, bool train, const Act** top) {
    train_bn = train;
    st = pose.st;
    TRY(begin_step());
    if (!train) TRY(pa_launch_bn_eval(bneval_jobs, n_bneval, eps, st));
    Hourglass& h = pose.hg[0];
    const Act* feats[5] = {&h.skip[0].x3, &h.skip[1].x3, &h.skip[2].x3, &h.skip[3].x3, &h.neck.x3};
    for (int k = 0; k < 5; ++k) TRY(asn_in[k].fwd(*this, *feats[k]));
    // the agent reads the pose net's BatchNorm scale/shift through op(): they belong to `pose`
    const Act* x = &asn_in[0].x3;
    for (int k = 0; k < 4; ++k) {
        const Act& pa = asn_pa[k];
        TRY(launch_pooladd_fwd(op(*x), op(asn_in[k + 1].x3), pa.raw, pa.B, pa.H, pa.W, pa.C, st));
        TRY(asn_merge[k].fwd(*this, pa));
        x = &asn_merge[k].x3;
    }
    for (int k = 0; k < 3; ++k) { TRY(asn_deep[k].fwd(*this, *x)); x = &asn_deep[k].x3; }
    *top = x;
    return 0;
}

// ASN.forward(is_dropout=True): [B][16] mask logits (cell = 4 * y + x)
int Net::asn_forward_masks(Net& pose, bool train, float* mask_logits) {
    if (!asn_dropout) { pa_set_error_msg("asn_forward_masks: this agent was created with the scale/rotation head"); return 1; }
    const Act* x = nullptr;
    TRY(asn_forward_trunk(pose, train, &x));
    hipLaunchKernelGGL(asn_mask_head_fwd_kernel, dim3(x->M()), dim3(64), 0, st, op(*x), chan, params + p_oc_w, params + p_oc_b, asn_logits);
    TRY((int)hipGetLastError());
    if (mask_logits) PA_CHECK(hipMemcpyAsync(mask_logits, asn_logits, (size_t)x->M() * sizeof(float), hipMemcpyDeviceToDevice, st));
    return 0;
}

int Net::asn_backward_masks(Net& pose, const float* dlogits) {
    if (!asn_dropout) { pa_set_error_msg("asn_backward_masks: this agent was created with the scale/rotation head"); return 1; }
    const Act& top = asn_deep[2].x3;
    bf16* dact = asn_pa[3].grad;
    hipLaunchKernelGGL(asn_mask_head_bwd_kernel, dim3(1), dim3(256), 0, st, op(top), dlogits, params + p_oc_w, top.M(), chan, dact,
                       grads + p_oc_w, grads + p_oc_b);
    TRY((int)hipGetLastError());
    return asn_backward_trunk(pose, dact);
}

int Net::asn_forward(Net& pose, bool train, float* logits_s, float* logits_r) {
    if (asn_dropout) { pa_set_error_msg("asn_forward: this agent was created with the occlusion-mask head"); return 1; }
    const Act* x = nullptr;
    TRY(asn_forward_trunk(pose, train, &x));
    const int HW = x->H * x->W, K = scale_num + rot_num;
    hipLaunchKernelGGL(asn_head_fwd_kernel, dim3(B), dim3(256), (chan + K) * sizeof(float), st, op(*x), HW, chan,
                       params + p_fcs_w, params + p_fcs_b, params + p_fcr_w, params + p_fcr_b, scale_num, rot_num, asn_feat,
                       asn_logits, asn_probs);
    TRY((int)hipGetLastError());
    if (logits_s) PA_CHECK(hipMemcpy2DAsync(logits_s, scale_num * sizeof(float), asn_logits, K * sizeof(float),
                                            scale_num * sizeof(float), B, hipMemcpyDeviceToDevice, st));
    if (logits_r) PA_CHECK(hipMemcpy2DAsync(logits_r, rot_num * sizeof(float), asn_logits + scale_num, K * sizeof(float),
                                            rot_num * sizeof(float), B, hipMemcpyDeviceToDevice, st));
    return 0;
}

// gradient of the KL loss w.r.t. every agent parameter; the pose net's features are constants (detached)
int Net::asn_backward(Net& pose, const float* target_s, const float* target_r, float* loss_out) {
    if (asn_dropout) { pa_set_error_msg("asn_backward: this agent was created with the occlusion-mask head"); return 1; }
    const Act& top = asn_deep[2].x3;
    const int HW = top.H * top.W, K = scale_num + rot_num;
    bf16* dact = asn_pa[3].grad;                   // scratch of the right size [B][4][4][C]; rewritten later in this pass
    hipLaunchKernelGGL(asn_head_bwd_kernel, dim3(B), dim3(256), K * sizeof(float), st, asn_probs, target_s, target_r,
                       params + p_fcs_w, params + p_fcr_w, scale_num, rot_num, HW, chan, B, asn_dlogits, dact, loss_dev, asn_log_eps);
    TRY((int)hipGetLastError());
    hipLaunchKernelGGL(asn_fc_wgrad_kernel, dim3(8), dim3(256), 0, st, asn_dlogits, asn_feat, B, scale_num, rot_num, chan,
                       grads + p_fcs_w, grads + p_fcs_b, grads + p_fcr_w, grads + p_fcr_b);
    TRY((int)hipGetLastError());
    TRY(asn_backward_trunk(pose, dact));
    if (loss_out) PA_CHECK(hipMemcpyAsync(loss_out, loss_dev, sizeof(float), hipMemcpyDeviceToDevice, st));
    return 0;
}

int Net::asn_backward_trunk(Net& pose, const bf16* dact) {
    Hourglass& h = pose.hg[0];
    const Act* feats[5] = {&h.skip[0].x3, &h.skip[1].x3, &h.skip[2].x3, &h.skip[3].x3, &h.neck.x3};
    const Act& top = asn_deep[2].x3;
    TRY(pa_launch_ep_apply(pa_plain(dact), final_ep(top), top.grad, (size_t)top.M(), top.C, st));
    TRY(finish_grad(top));
    TRY(asn_deep[2].bwd(*this, asn_deep[1].x3, pa_none(), true)); TRY(finish_grad(asn_deep[1].x3));
    TRY(asn_deep[1].bwd(*this, asn_deep[0].x3, pa_none(), true)); TRY(finish_grad(asn_deep[0].x3));
    TRY(asn_deep[0].bwd(*this, asn_merge[3].x3, pa_none(), true)); TRY(finish_grad(asn_merge[3].x3));
    for (int k = 3; k >= 0; --k) {
        const Act& pa = asn_pa[k];
        TRY(asn_merge[k].bwd(*this, pa, pa_none(), true));                   // -> pa.grad (plain)
        const Act& lo = asn_in[k + 1].x3;
        const Act& hi = (k == 0) ? asn_in[0].x3 : asn_merge[k - 1].x3;
        TRY(pa_launch_ep_apply(pa_plain(pa.grad), final_ep(lo), lo.grad, (size_t)lo.M(), lo.C, st));
        TRY(finish_grad(lo));
        TRY(pa_launch_maxpool_bwd(pa.grad, op(hi), pa_none(), final_ep(hi), hi.grad, hi.B, hi.H, hi.W, hi.C, st));
        TRY(finish_grad(hi));
        TRY(asn_in[k + 1].bwd(*this, *feats[k + 1], pa_none(), false));
    }
    TRY(asn_in[0].bwd(*this, *feats[0], pa_none(), false));
    return reduce_grads();
}
