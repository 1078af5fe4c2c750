/* poseadv.h -- C ABI of libposeadv_hip.so: the MI355X (gfx950) hot path of the stacked-hourglass
 * pose trainer with adversarial scale/rotation augmentation.
 *
 * Conventions:
 *   - every function returns 0 on success, otherwise a hipError_t / library error code;
 *     pa_last_error() gives the message.  No C++ exception crosses the ABI.
 *   - all pointers are CALLER-OWNED DEVICE pointers unless the name ends in _host; the library
 *     never allocates persistent device memory: a network runs inside one caller-provided workspace
 *     whose size comes from pa_net_workspace_bytes() (cleared by pa_net_bind).
 *   - calls are asynchronous on the given hipStream_t (passed as void*); one host thread per net.
 *   - activations are NHWC bf16 inside the library; the entry points that mirror the reference's
 *     Python signatures take/return NCHW fp32 exactly like the reference's torch tensors.
 *
 * Each entry point cites the reference interface it replaces (paths relative to the reference
 * checkout zhiqiangdon/pose-adv-aug).
 */
#ifndef POSEADV_H
#define POSEADV_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

const char* pa_last_error(void);
int pa_version(void);
/* Which build this is: 0 = bfloat16 storage (libposeadv_hip.so), 1 = IEEE-half storage (libposeadv_hip_fp16.so: "fp16 MFMA
 * 1x1 convs" of the deep-stack configuration).  Same ABI; "bf16" in the comments below means this 16-bit type.
 * pa_grad_scale: the factor every fp32 gradient the library returns carries (1 for bf16; the fp16 build runs its backward pass
 * on scaled gradients so that they stay inside half's range): pass gscale = 1 / (world * pa_grad_scale()) to pa_rmsprop_step. */
int pa_dtype(void);
float pa_grad_scale(void);

/* ---------------------------------------------------------------- pose library (pylib/) ---- */

/* HumanPts.pts2heatmap (pylib/HumanPts.py:36-46) + draw_gaussian (:82-116), batched.
 * pts: [B][J][2] float64 (x, y) in heat-map pixels; out: [B][J][H][W] fp32. */
int pa_gaussian_heatmap(const double* pts, float* out, int B, int J, int H, int W, void* stream);

/* Criterion.weighted_L2 (pylib/Criterion.py:12-18); weight may be NULL (== 1), which is the inline
 * loss term of stack-hg.py:156-159.  *loss (fp32, device) is ACCUMULATED into: zero it first. */
int pa_weighted_l2(const float* pred, const float* gt, const float* weight, size_t n, float* loss, void* stream);

/* Evaluation.get_preds (pylib/Evaluation.py:6-23): maps [B][J][H][W] fp32 -> preds [B][J][2]
 * (1-based x, y; zero where the maximum is <= 0).  maxval [B][J] may be NULL. */
int pa_get_preds(const float* maps, int B, int J, int H, int W, float* preds, float* maxval, void* stream);

/* Evaluation.final_preds (pylib/Evaluation.py:169-193 + transform_preds :195-211 + TransformPts
 * :240-248): quarter-pixel refinement, +0.5, back-projection to original-image pixels (integer
 * valued).  center [B][2], scale [B], rot [B] fp32; out [B][J][2] fp32. */
int pa_final_preds(const float* maps, const float* center, const float* scale, const float* rot,
                   int B, int J, int H, int W, float* out, float* scratch_preds, void* stream);

/* PCK from point sets: Evaluation.calc_dists/dist_acc/accuracy (pylib/Evaluation.py:25-75; boundary
 * 1), accuracy_origin_res (:77-97; boundary 0), per_person_pckh (:99-167; `person` output with the
 * visibility points `vis`), HumanAcc.approx_PCKh (pylib/HumanAcc.py:7-44; boundary 0, norm = res/10
 * with Python-2 integer division done by the caller).
 * pred, gt: [B][J][2]; norm: [B]; idxs: int32 [nidx] (device); acc: [nidx+1] or NULL;
 * person: [B] or NULL; vis: [B][J][2] or NULL; dists: [J][B] (calc_dists' matrix) or NULL. */
int pa_pck(const float* pred, const float* gt, const float* norm, float boundary, const int32_t* idxs, int nidx,
           float thr, const float* vis, int B, int J, float* acc, float* person, float* dists, void* stream);

/* HumanAug.GetTransform (pylib/HumanAug.py:10-35) for a batch: params [B][8] float64 =
 * {cx, cy, scale, rot_deg, flip, gain_r, gain_g, gain_b}; t_out [B][6] float64 = forward transform
 * at res_out (first two rows), tinv_in [B][6] = INVERSE transform at res_in (what the warp samples with). */
int pa_affine_params(const double* params, int B, int res_in, int res_out, double* t_out, double* tinv_in, void* stream);
/* fp32 copies of c, s, r as the metric calls take them (`c = meta['center']`, `s = meta['scale']`, `r = meta['rot']` of the reference's
 * batches, stack-hg.py:140-143): csr [4 B] floats = c [B][2] | s [B] | r [B] from params [B][8]. */
int pa_params_csr(const double* params, int B, float* csr, void* stream);

/* HumanAug.TransformPts (pylib/HumanAug.py:45-54) + shufflelr (:236-257) + the invalid-joint rule of
 * data/mpii_for_mpii.py:142-146.  pts [B][J][2] fp32 image pixels -> out [B][J][2] float64 heat-map
 * coords (0 for invalid joints); pts_img (optional) = mirrored/swapped image-space joints. */
int pa_transform_pts(const float* pts, const double* params, const double* t, int B, int J, float width,
                     double* out, float* pts_img, void* stream);

/* The PURE inverse-affine bilinear sampler (2 x 2 taps, no pre-filter): the geometry of HumanAug.crop
 * (pylib/HumanAug.py:117-176) + flip / colour gain of data/mpii_for_mpii.py:126-135 as one gather.  NOT the reference's
 * pixels (those are pa_crop below, which the training / validation paths use); kept as an operator.  src: uint8 [B][Hs][Ws][3]; out4: bf16 [B][res][res][4]
 * (network input layout, 4th channel 0) and/or outf: fp32 [B][3][res][res]; either may be NULL. */
int pa_affine_warp_bilinear(const uint8_t* src, int Hs, int Ws, const double* tinv, const double* params,
                            int B, int res, void* out4, float* outf, void* stream);
/* The same two calls for a batch of frames of DIFFERENT sizes stored top-left aligned in a padded [B][Hs][Ws][3] buffer
 * (real MPII images: data/mpii_for_mpii.py:97-135 loads one JPEG per person).  sizes: int32 [B][2] = width, height of
 * each sample's own frame: the zero padding outside it, the mirror of a flipped frame (x <- w_b - 1 - x) and the
 * mirror of its joints (x <- w_b - x, pylib/HumanAug.py:236-257) use the sample's own width. */
int pa_affine_warp_bilinear_sized(const uint8_t* src, int Hs, int Ws, const int32_t* sizes, const double* tinv,
                                  const double* params, int B, int res, void* out4, float* outf, void* stream);
int pa_transform_pts_sized(const float* pts, const double* params, const double* t, int B, int J, const int32_t* sizes,
                           double* out, float* pts_img, void* stream);

/* HumanAug.crop (pylib/HumanAug.py:117-176) behind the dataset's pre-processing (data/mpii_for_mpii.py:114-135,
 * utils/imutils.py:31-40), stage for stage with the arithmetic of the libraries the reference delegates its pixels to
 * (scipy.misc.pilutil over Pillow): source bytes = fp32 frame / 255, mirrored, times gain, clamped; for
 * scale * 200 / res >= 2 the antialiased whole-frame pre-downscale (:121-133); the int-truncated, rotation-padded,
 * zero-filled window (:136-164); PIL's bilinear rotate about the window centre (:166-170); PIL's bilinear (triangle
 * filter) resize to res x res (:175); uint8 / 255.  Byte-exact against the reference's crop over Pillow 12 for crops
 * that hold a black and a white pixel (scipy's per-image min/max stretching is not applied).
 * src: uint8 [B][Hs][Ws][3]; sizes: int32 [B][2] = (width, height) of each sample's own frame inside the padded
 * buffer, or NULL (all Ws x Hs); params: [B][8] as for pa_affine_params (centre AFTER the mirror);
 * workspace: pa_crop_workspace_bytes() bytes (scratch, no initialisation needed);
 * out4: bf16 [B][res][res][4] network input, outf: fp32 [B][3][res][res], out8: uint8 [B][res][res][3]; any may be NULL. */
size_t pa_crop_workspace_bytes(int B, int Hs, int Ws, int res);
/* Upper bound of the bytes one pa_crop call reads (out[0]) and writes (out[1]), from the geometry of its own stage buffers at their
 * worst-case window sizes (bench.py's roofline.floor; the reference's crop, pylib/HumanAug.py:117-176, has no such notion). */
int pa_crop_design_bytes(int B, int Hs, int Ws, int res, double* out_host);
int pa_crop(const uint8_t* src, int Hs, int Ws, const int32_t* sizes, const double* params, int B, int res, void* workspace,
            void* out4, float* outf, uint8_t* out8, void* stream);

/* Validation with flip test-time augmentation (stack-hg.py:222-230).
 * pa_flip_lr_nhwc4: mirror the bf16 NHWC4 network input along W (img.numpy()[:, :, :, ::-1], :223).
 * pa_flip_tta_merge: merged = (out + shuffle_channels_for_horizontal_flipping(flip_channels(out_flipped))) / 2
 * (pylib/HumanAug.py:198-210, :179-196, stack-hg.py:228-230) over NCHW fp32 heat maps [B][16][H][W]. */
int pa_flip_lr_nhwc4(const void* src, void* dst, int B, int H, int W, void* stream);
int pa_flip_tta_merge(const float* out, const float* out_flipped, float* merged, int B, int J, int H, int W, void* stream);

/* augmentation laws: mode 0 = data/mpii_for_mpii.py:119-135 (regular), 1 = agent bins
 * (data/joint_train_s_r_agent.py:15-16,33-36,134-139), 2 = agent scale only, 3 = agent rotation only.
 * meta [B][4] = {objpos_x, objpos_y, scale, frame_width}; params [B][8] as above. */
int pa_sample_aug(const float* meta, const int32_t* scale_idx, const int32_t* rot_idx, int mode,
                  uint64_t seed, uint64_t step, int B, double* params, void* stream);
/* the same laws with CALLER-GIVEN draws instead of the engine's counter-based stream: draws [B][7] float64 =
 * {N(0,1) scale, N(0,1) rotation, U "rotation forced to 0", U flip, U gain_r, U gain_g, U gain_b} -- with the draws of
 * np.random in the reference's call order the result equals the reference's parameters value for value. */
int pa_sample_aug_given(const float* meta, const int32_t* scale_idx, const int32_t* rot_idx, int mode,
                        const double* draws, int B, double* params, void* stream);

/* softmax + np.random.choice(K, p) of joint-train-pose-s-r-agent.py:252-271.  logits [B][K];
 * probs [B][K] and idx int32 [B] may be NULL. */
int pa_sample_categorical(const float* logits, int B, int K, uint64_t seed, uint64_t step, unsigned slot,
                          float* probs, int32_t* idx, void* stream);

/* _Hourglass._sample_mask (models/asn_stacked_hg.py:102-136): softmax over the `cells` (= 16) mask logits of each sample,
 * `dropout_num` (= 2) distinct cells drawn with those probabilities (the law of np.random.choice(replace=False)),
 * masks [B][cells] fp32 = 1 except 0 at the drawn cells, indexes int32 [B][dropout_num] (may be NULL), probs [B][cells]
 * (may be NULL).  uniforms: NULL = the engine's counter-based stream (seed, step); else float64 [B][dropout_num] in (0,1). */
int pa_sample_dropout_masks(const float* logits, int B, int cells, int dropout_num, uint64_t seed, uint64_t step,
                            const double* uniforms, float* probs, float* masks, int32_t* indexes, void* stream);
/* _Hourglass._dropout (models/asn_stacked_hg.py:79-100) on one NHWC bf16 tensor [B][H][W][C]: the [B][16] 4x4 cell mask,
 * nearest-upsampled to H x W (H, W multiples of 4), times every channel. */
int pa_cell_mask(const void* x_bf16, const float* masks, void* out_bf16, int B, int H, int W, int C, void* stream);

/* torch.optim.RMSprop(alpha, eps, momentum=0) step (stack-hg.py:51-52) on flat fp32 arrays;
 * gscale multiplies the gradient first (1/world after a sum all-reduce). */
int pa_rmsprop_step(float* param, const float* grad, float* square_avg, size_t n, float lr, float alpha,
                    float eps, float gscale, void* stream);
/* Half-precision build (pa_dtype() == 1) only: a step whose gradient holds inf / NaN (an fp16 overflow in the scaled backward
 * pass) is SKIPPED -- parameters and square_avg untouched, the loss-scaling convention the reference's fp32 path never needs --
 * and counted.  *count = steps skipped so far in this process (synchronises `stream`); always 0 in the bf16 build. */
int pa_rmsprop_skipped_steps(long long* count, void* stream);
/* The same two calls with the skip state owned by the OPTIMIZER instead of the process: `state` = two int32 on the device
 * ({this gradient is non-finite, steps skipped so far}), zeroed once by the caller.  Two optimizers that step on different streams
 * (pose net and agent, joint-train-pose-s-r-agent.py:86-94) each pass their own pair; pa_rmsprop_step shares one pair per process. */
int pa_rmsprop_step_state(float* param, const float* grad, float* square_avg, size_t n, float lr, float alpha,
                          float eps, float gscale, int32_t* state, void* stream);
int pa_rmsprop_skipped_steps_state(const int32_t* state, long long* count, void* stream);

/* ---------------------------------------------------------------- operator level ---------- */
/* One residual bottleneck block _Residual(C, C) (models/asn_stacked_hg.py:11-49), forward + backward,
 * NCHW fp32 in/out, parameters in the block's own state_dict order (flat fp32).  Used by the parity
 * tests; the networks below run the same kernels.  ws: zero-filled workspace of
 * pa_residual_workspace_bytes(). */
size_t pa_residual_workspace_bytes(int B, int H, int W, int C);
int pa_residual_fwd_bwd(const float* x, const float* dy, const float* params, float* y, float* dx, float* grads,
                        float* buffers, int B, int C, int H, int W, void* ws, void* stream);

/* One plain convolution (k = 1 or 3, stride 1, 'same'; nn.Conv2d of models/asn_stacked_hg.py:17-24)
 * through the implicit-GEMM kernels, NCHW fp32 in/out, PyTorch-layout fp32 weights.
 *   mode 0: out  = conv(a_in = x, w) + bias
 *   mode 1: out  = data gradient for a_in = dy
 *   mode 2: out  = weight gradient, out2 = bias gradient, for a_in = dy, b_in = x
 * ws: pa_conv2d_workspace_bytes() bytes. */
size_t pa_conv2d_workspace_bytes(int B, int Cin, int Cout, int H, int W, int k);
int pa_conv2d(int mode, const float* a_in, const float* b_in, const float* w, const float* bias, float* out, float* out2,
              int B, int Cin, int Cout, int H, int W, int k, void* ws, void* stream);

/* Several INDEPENDENT weight gradients the way the training step submits them (the conv1 / conv2 / conv3 / adapter gradients of one
 * _Residual, models/asn_stacked_hg.py:17-24, wait for one event and leave as ONE grouped launch): jobs[j] describes layer j
 * (device pointers, NCHW fp32 operands, PyTorch-layout fp32 outputs).
 *   dy_q / dy_k given: the dy operand arrives through the BatchNorm backward, dy_eff = kA[c] dy + kB[c] dy_q + kC[c], dy_k = [3][Cout]
 *   x_k given        : the x operand arrives through BatchNorm + ReLU,        x_eff  = max(0, k0[c] x + k1[c]),     x_k  = [2][Cin]
 *   db               : bias gradient (1x1 layers only) or NULL
 * mode 0: every layer as a launch of its own; 1: one grouped launch (jobs ordered longest first, as the step does); 2: one grouped
 * launch with the jobs in the caller's order.  Modes 1 / 2 return 2 when a job is not a shape / operand mode the grouped kernel takes
 * (3x3: Cin, Cout % 64 == 0, H % 8 == 0, W % 16 == 0, >= 48 tiles of 8 x 16 pixels, x through BatchNorm + ReLU; 1x1: >= 3 tiles of 128 pixels).
 * ws: pa_wgrad_group_workspace_bytes() bytes. */
#define PA_WG_GROUP_MAX 8
typedef struct pa_wgrad_job {
    int B, Cin, Cout, H, W, k;
    const float* dy;   const float* dy_q; const float* dy_k;
    const float* x;    const float* x_k;
    float* dw;         float* db;
} pa_wgrad_job;
size_t pa_wgrad_group_workspace_bytes(const pa_wgrad_job* jobs, int njobs);
int pa_wgrad_group(const pa_wgrad_job* jobs, int njobs, int mode, void* ws, void* stream);

/* NCHW fp32 <-> NHWC bf16 */
int pa_nchw_to_nhwc(const float* src, void* dst_bf16, int B, int C, int H, int W, void* stream);
int pa_nhwc_to_nchw(const void* src_bf16, float* dst, int B, int C, int H, int W, void* stream);

/* ---------------------------------------------------------------- networks ---------------- */
typedef struct pa_net pa_net;

/* create_hg(num_stacks, num_modules=1, num_classes=16, chan) (models/asn_stacked_hg.py:344-347) for a
 * fixed per-GPU batch B and input resolution res (256).  chan must be a multiple of 128. */
pa_net* pa_hg_create(int num_stacks, int num_classes, int chan, int B, int res);
/* create_asn(chan, chan, scale_num, rotation_num, is_aug=True) (models/asn_stacked_hg.py:441-444) bound
 * to the feature shapes of a pose net with neck resolution res/64. */
pa_net* pa_asn_create(int chan, int scale_num, int rotation_num, int B, int res);
/* create_asn(chan, chan, is_dropout=True) (models/asn_stacked_hg.py:378-379,437-439): the same trunk with the occlusion
 * head out_conv = Conv2d(chan, 1, 1) on the 4x4 map.  res must be 256 (the mask is the 4x4 neck map). */
pa_net* pa_asn_create_dropout(int chan, int B, int res);
void pa_net_destroy(pa_net* net);

/* state_dict description (names, shapes, offsets) in the reference's registration order */
int pa_net_num_tensors(const pa_net* net);
/* kind: 0 parameter (offset into the flat parameter array), 1 running_mean/var (offset into the flat
 * buffer array), 2 num_batches_tracked (no storage in the library) */
int pa_net_tensor_info(const pa_net* net, int i, char* name, int name_cap, int* shape4, int* ndim,
                       size_t* offset, size_t* numel, int* kind);
size_t pa_net_param_floats(const pa_net* net);
size_t pa_net_buffer_floats(const pa_net* net);
size_t pa_net_workspace_bytes(const pa_net* net);

/* params / grads / buffers: flat fp32 device arrays of pa_net_param_floats / pa_net_buffer_floats;
 * workspace: pa_net_workspace_bytes bytes; pa_net_bind clears it (no initialisation needed from the caller). */
int pa_net_bind(pa_net* net, float* params, float* grads, float* buffers, void* workspace, void* stream);
/* refresh the bf16 compute copies of the weights (after load_state_dict / an optimizer step) */
int pa_net_prepare_weights(pa_net* net);

/* _Hourglass_Wrapper.forward (models/asn_stacked_hg.py:282-342) + loss (stack-hg.py:156-159).
 * img: NCHW fp32 [B][3][res][res], or img4: bf16 NHWC4 from pa_affine_warp_bilinear (one of them).
 * pts: [B][16][2] float64 heat-map coordinates of the target joints, or NULL (no loss).
 * train != 0: BatchNorm uses batch statistics and updates the running estimates.
 * loss_per_stack: device fp32 [num_stacks] or NULL. */
int pa_hg_forward(pa_net* net, const float* img, const void* img4, const double* pts, int train, float* loss_per_stack);
/* heat maps of stack i: NHWC fp32 [B][H/4][W/4][16] inside the workspace (valid until the next forward) */
const float* pa_hg_heatmap_nhwc(const pa_net* net, int stack);
/* copy stack i's heat maps out as NCHW fp32 [B][16][res/4][res/4] (the reference's output tensors) */
int pa_hg_heatmap_nchw(pa_net* net, int stack, float* out);
/* loss.backward() (stack-hg.py:164): fills the flat gradient array bound with pa_net_bind */
int pa_hg_backward(pa_net* net);

/* loss.backward() in phases, for a data-parallel gradient exchange that overlaps the backward pass:
 * phase p = 0 .. num_stacks-1 enqueues the backward pass of stack num_stacks-1-p (its head layers, post block and hourglass),
 * phase num_stacks the stem and the final reductions; all phases in order == pa_hg_backward.
 * pa_hg_bucket_range: [lo, hi) = the contiguous range of the flat parameter / gradient array that holds hg.<stack> (13
 * residual blocks: 42 % of a 2-stack net's parameters each).
 * pa_hg_bucket_wait: call after the phase of `stack`: makes `stream` (the caller's communication stream) wait until every
 * gradient in that range is final -- the main chain's work so far AND the slab reductions of the engine's weight-gradient
 * stream.  Returns 0, or -1 if the current stream mode finishes gradients only at the end (exchange everything after the
 * last phase then). */
int pa_hg_backward_phase(pa_net* net, int phase);
int pa_hg_bucket_range(const pa_net* net, int stack, size_t* lo, size_t* hi);
int pa_hg_bucket_wait(pa_net* net, int stack, void* stream);

/* pa_hg_forward(img4, pts) + pa_hg_backward in ONE call (stack-hg.py:153-164 without the optimizer).  use_graph != 0: the
 * ~650 launches of the two passes (main stream + the engine's side / weight-gradient streams, their fork / join events as
 * edges) are captured into a HIP graph on the first call and replayed afterwards; img4 / pts are copied into the engine's
 * own input buffers first (the graph's pointers are fixed).  Not available while dropout masks are set or the launch
 * profiler runs.  The gradient lands in the bound flat array; loss_per_stack (device, [num_stacks]) may be NULL. */
int pa_hg_train_step(pa_net* net, const void* img4, const double* pts, int train, int use_graph, float* loss_per_stack);
/* total_dev (device float, or NULL = off): every later pa_hg_forward / pa_hg_train_step that is given `pts` also writes
 * sum_stacks loss there -- `loss = sum(criterion(o, target))` of stack-hg.py:156-159 as one device scalar, so that the caller needs no
 * reduction launch of its own between the backward pass and the optimizer. */
int pa_hg_set_loss_total(pa_net* net, float* total_dev);

/* Half-hourglass forward (models/asn_stacked_hg.py:300-304 with is_half_hg): stem + the down path of
 * hg[0] up to the neck -- everything the agent reads.  train != 0 updates the BatchNorm running
 * statistics of those layers, as the reference does on the agent-augmentation steps (Appendix A.9);
 * train == 2: batch statistics without touching the running estimates (the occlusion branch's whole-hourglass
 * call runs this pass, samples the masks and then the full forward, which does the one update). */
int pa_hg_forward_half(pa_net* net, const float* img, const void* img4, int train);

/* ASN.forward (models/asn_stacked_hg.py:401-436) on the pose net's detached features of the last
 * (half or full) forward (:159-164): logits [B][scale_num] and [B][rotation_num] (device, may be NULL).
 * train selects the agent's BatchNorm mode. */
int pa_asn_forward(pa_net* asn, pa_net* pose, int train, float* logits_scale, float* logits_rot);
/* softmax of the last pa_asn_forward's logits: [B][scale_num + rotation_num] (scale first), in the workspace */
const float* pa_asn_probs(const pa_net* asn);
/* KL loss of joint-train-pose-s-r-agent.py:399-407 against the target distributions (gen_groundtruth,
 * utils/util.py:147) + backward into the agent's flat gradient; no gradient reaches the pose net.
 * target_* [B][K] fp32 device; loss: device fp32 scalar or NULL. */
int pa_asn_backward(pa_net* asn, pa_net* pose, const float* target_scale, const float* target_rot, float* loss);
/* eps of log(softmax + eps) in that loss: 1e-7 (default, joint-train-pose-s-r-agent.py:399-404) or 0 for the agent
 * pre-training, which uses LogSoftmax (pretrain-s-r-agent.py:177-190). */
int pa_asn_set_log_eps(pa_net* asn, float eps);

/* Occlusion branch.
 * pa_asn_forward_masks: ASN.forward(is_dropout=True) (:437-439) on the pose net's detached features: mask_logits [B][16]
 * (cell = 4 * y + x), device, may be NULL.  train as in pa_asn_forward.
 * pa_asn_backward_masks: backward of the agent from a caller-provided d(loss)/d(mask_logits) [B][16] into the agent's flat
 * gradient (the reference ships no loss for this branch); nothing reaches the pose net.
 * pa_hg_set_dropout_masks: masks [B][16] fp32 (caller-owned device memory, e.g. from pa_sample_dropout_masks), or NULL to
 * switch the branch off.  While set, pa_hg_forward multiplies the neck and the four skip tensors of EVERY stack's hourglass by
 * the nearest-upsampled mask (:172-190, :322-324) and pa_hg_backward differentiates through it. */
int pa_asn_forward_masks(pa_net* asn, pa_net* pose, int train, float* mask_logits);
int pa_asn_backward_masks(pa_net* asn, pa_net* pose, const float* dlogits);
int pa_hg_set_dropout_masks(pa_net* net, const float* masks);

/* Evaluation.accuracy (pylib/Evaluation.py:54-75) of stack i's heat maps against the Gaussian target
 * of the joints given to the last forward: acc [nidx+1]. */
int pa_hg_accuracy(pa_net* net, int stack, const int32_t* idxs, int nidx, float* acc, float* scratch);

/* Evaluation.accuracy_origin_res (pylib/Evaluation.py:77-97) and, when person != NULL,
 * per_person_pckh (:99-167) of stack i's heat maps of the last forward, read in place (NHWC fp32).
 * center [B][2], scale [B], rot [B], gt_pts [B][16][2], norm [B] fp32; idxs int32 [nidx];
 * acc [nidx+1] or NULL; person [B] or NULL; scratch: 6*B*16 (+ B*16*(res/4)^2 if person) floats. */
int pa_hg_pckh(pa_net* net, int stack, const float* center, const float* scale, const float* rot, const float* gt_pts,
               const float* norm, const int32_t* idxs, int nidx, float* acc, float* person, float* scratch);

/* The meters of a TRAINING step beside its backward pass.  stack-hg.py:171-180 computes the accuracies of a batch from the forward pass's
 * heat maps after the optimizer step; they only read those maps, so a caller may ask for them between pa_hg_forward and pa_hg_backward:
 * with on != 0 the pa_hg_accuracy / pa_hg_pckh calls that follow are launched on a stream of the engine's own, behind everything the net's
 * stream holds at the call, and run beside the launches that stream is given next.  The net's stream waits for them at the end of the
 * next pa_hg_backward (the last pa_hg_backward_phase) or in front of the next forward pass; until then `acc`, `person`, `scratch` and the
 * meters' inputs must stay allocated and must not be read.  on == 0 (default): the meters run on the net's stream, ordered like every
 * other call.  Same values either way. */
int pa_net_meters_async(pa_net* net, int on);

/* Per-launch HIP-event timing of the MFMA kernels on the net's stream (bench.py's `roofline`).
 * begin: start recording; report: synchronise, stop recording and fill out_host[c][4] = {total ms, launches, algorithmic
 * bytes, flops} for the classes c = 0 fwd 1x1, 1 fwd 3x3, 2 dgrad 1x1, 3 dgrad 3x3, 4 wgrad 1x1, 5 wgrad 3x3 (maps of >= 16384
 * pixels: 32 x 32 and larger at batch 24), 6 stem fwd, 7 stem wgrad, 8 unused (the fused
 * low-resolution launch of rounds 3-4, removed: DESIGN.md), 9 / 10 / 11 forward / data-gradient / weight-gradient launches of the smaller maps (latency-bound).
 * out_host is a HOST array of `cap_classes` rows; classes beyond the capacity are dropped (never written).  Returns the library's
 * number of classes (PA_PROF_CLASSES = 12 today) through *n_classes when it is not NULL. */
#define PA_PROF_CLASSES 12
int pa_net_profile_begin(pa_net* net);
int pa_net_profile_report(pa_net* net, double* out_host, int cap_classes, int* n_classes);
/* class (0..11 as above) of every timed launch of the last reported pass, in launch order: returns their number and fills
 * out_host[0..min(cap, n)) (HOST array).  tools/trace_classes.py matches a rocprofv3 kernel trace of the same pass with it. */
int pa_net_profile_classes(const pa_net* net, int32_t* out_host, int cap);
/* BatchNorm finalize (models/asn_stacked_hg.py:19,22,25 in training mode) of a residual block's inner tensors at the low-resolution
 * levels: with at most `max_rows` partial statistics rows (default and upper limit 128: the 16 x 16 and smaller maps at batch 24) the
 * consumer convolution computes scale / shift (forward) or the BatchNorm-backward coefficients (backward) in its own prologue instead
 * of a finalize launch in front of it; 0 = always a launch.  Results are bit-identical either way (one summation order). */
int pa_net_set_fin_prologue(pa_net* net, int max_rows);
/* Bytes THIS design moves per step, from the engine's own launch table (bench.py's `roofline.design_bytes_per_step`): every operand
 * a launch of the last forward + backward pass reads or writes, counted once per launch at its storage width -- activations, the
 * second operand of a BatchNorm-backward load, reference tensors of the masked epilogues, shortcut addends, stored dz tensors, fp32
 * weight-gradient slabs written and read back, weights; halo re-reads and cache hits are NOT modelled (it is a floor for this set
 * of fusions, above the "every activation once" figure of SURVEY.md section 8d).  out_host[0] = bytes read, [1] = bytes written. */
int pa_net_design_bytes(const pa_net* net, double* out_host);
/* Bandwidth calibration for the same floor: dst[i] = src[i] with the engine's own plain 16-byte-per-lane streaming kernel (`bytes` a
 * multiple of 16).  Not part of the reference's surface (stack-hg.py has no roofline); bench.py times it on cold buffers of the step's
 * tensor sizes and on GB-sized ones. */
int pa_copy_probe(void* dst, const void* src, size_t bytes, void* stream);
/* ... in a chosen form: 0 = pa_copy_probe (grid-stride, four chunks in flight per thread), 1 = one 16-byte chunk per thread and no loop
 * (the "float4 copy" /opt/skills/guides/MI355X_MICROARCH.md quotes 6.29 TB/s for), 2 = form 0 with non-temporal loads / stores. */
int pa_copy_probe_form(void* dst, const void* src, size_t bytes, int form, void* stream);
/* Micro-benchmark of ONE convolution launch (tools/bench_conv*.py; no reference counterpart): mode 0 forward, 1 data gradient,
 * 2 weight gradient; variant bits: 1 input transform (BatchNorm+ReLU / BatchNorm backward on load), 2 statistics / masked epilogue,
 * 4 one residual addend, 8 (weight gradient) BatchNorm+ReLU on the x operand, 16: cold protocol is the caller's business.  `ws` =
 * caller-provided device workspace (>= 1 GiB for the benchmark's shapes); *ms_out = average milliseconds per launch over `iters`
 * launches (HIP events on `stream`). */
int pa_conv2d_time(int mode, int variant, int B, int Cin, int Cout, int H, int W, int k, int iters, void* ws, float* ms_out, void* stream);

/* The engine enqueues independent branches (hourglass skip blocks, weight gradients) on internal side
 * streams that fork from / join into the caller's stream by events.  on = 0 serialises everything on
 * the caller's stream (clean per-kernel timings: bench.py's roofline pass); on = 1 restores the default. */
int pa_net_set_multi_stream(pa_net* net, int on);

/* Test hook: copy an internal activation (pending BatchNorm+ReLU applied) or, with grad != 0, its raw
 * gradient buffer out as NCHW fp32; shape4 receives {B, C, H, W} (out may be NULL to query the shape).
 * Names: "stem", "res1".."res3", "pool0", "hg<i>.skip<k>|pool<k>|down<k>|up<k>|merge<k>|neck" (k=1..4),
 * "post<i>", "lin<i>", "xin<i>", optional suffix ".x1"/".x2" for a block's inner tensors. */
int pa_hg_debug_tensor(pa_net* net, const char* which, int grad, float* out, int* shape4);

/* The same hook for the agent: "in<k>" (k = 0..4: residual_skip1-4, residual_neck), "pa<k>", "merge<k>"
 * (k = 0..3), "deep<k>" (k = 0..2), optional suffix ".x1"/".x2". */
int pa_asn_debug_tensor(pa_net* net, const char* which, int grad, float* out, int* shape4);

#ifdef __cplusplus
}
#endif
#endif
