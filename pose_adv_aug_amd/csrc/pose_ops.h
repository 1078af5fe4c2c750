// Launch interface of the pose-library kernels (internal; public C ABI in include/poseadv.h).
#pragma once
#include "common.h"

int pa_launch_gaussian_heatmap(const double* pts, float* out, int B, int J, int H, int W, hipStream_t st);
int pa_launch_weighted_l2(const float* pred, const float* gt, const float* w, size_t n, float* loss, hipStream_t st);
int pa_launch_head_fwd(const PaOperand& in, const bf16* w16, const float* bias, float* heat, bf16* heat64,
                       const double* pts, float* loss, int B, int H, int W, int Cin, hipStream_t st);
int pa_launch_heat_grad(const float* heat, const double* pts, const bf16* add64, bf16* dheat64, float gscale,
                        int B, int H, int W, hipStream_t st);
int pa_launch_argmax(const float* maps, long sb, long sj, long sp, int B, int J, int H, int W, float* preds, float* maxval,
                     hipStream_t st);
int pa_launch_final_preds(const float* maps, long sb, long sj, long sp, const float* coords, const float* center, const float* scale,
                          const float* rot, int B, int J, int H, int W, float* out, hipStream_t st);
int pa_launch_pck(const float* pred, const float* gt, const float* norm, float boundary, const int* idxs, int nidx, float thr,
                  const float* vis, int B, int J, float* acc, float* person, float* dists_out, hipStream_t st);
int pa_launch_params_csr(const double* params, int B, float* csr, hipStream_t st);
int pa_launch_affine_params(const double* params, int B, int res_in, int res_out, double* t_out, double* tinv_in, hipStream_t st);
int pa_launch_transform_pts(const float* pts, const double* params, const double* t, int B, int J, float width, const int* sizes,
                            double* out, float* pts_img, hipStream_t st);
int pa_launch_warp(const unsigned char* src, int Hs, int Ws, const int* sizes, const double* tinv, const double* params, int B, int res,
                   bf16* out4, float* outf, hipStream_t st);
int pa_launch_sample_aug(const float* meta, const int* scale_idx, const int* rot_idx, int mode, unsigned long long seed,
                         unsigned long long step, const double* draws, int B, double* params, hipStream_t st);
int pa_launch_sample_categorical(const float* logits, int B, int K, unsigned long long seed, unsigned long long step, unsigned slot,
                                 float* probs, int* idx, hipStream_t st);
int pa_launch_flip_lr_nhwc4(const bf16* src, bf16* dst, int B, int H, int W, hipStream_t st);
int pa_launch_flip_tta_merge(const float* a, const float* b, float* out, int B, int H, int W, hipStream_t st);
int pa_launch_sample_dropout_masks(const float* logits, int B, int K, int k, unsigned long long seed, unsigned long long step,
                                   const double* uniforms, float* probs, float* masks, int* indexes, hipStream_t st);
size_t pa_crop_workspace_size(int B, int Hs, int Ws, int res);
void pa_crop_bytes_bound(int B, int Hs, int Ws, int res, double* rd, double* wr);
int pa_launch_crop(const unsigned char* src, int Hs, int Ws, const int* sizes, const double* params, int B, int res, void* workspace,
                   bf16* out4, float* outf, unsigned char* out8, hipStream_t st);
