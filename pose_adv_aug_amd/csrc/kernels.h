// Host-side launch interface of the HIP kernels (internal; the public C ABI is include/poseadv.h).
#pragma once
#include "common.h"
#include "bn_fin.h"

struct PaConvArgs {
    PaOperand in;        // [B][H][W][Cin] NHWC bf16
    const bf16* w;       // [Cout][taps*Cin] bf16 (tap-major, channel-minor)
    const float* bias;   // [Cout] or nullptr
    PaOperand add1;      // optional epilogue addends over [M][Cout]
    PaOperand add2;
    PaEpilogue ep;
    bf16* out;           // [M][Cout]
    int B, H, W, Cin, Cout, taps;
    bf16* dz_out;        // optional (1x1 row-tile kernel, LIN2 input): the transformed input tile is also stored here [M][Cin] --
                         // later consumers of the same BatchNorm-backward gradient read ONE tensor instead of recomputing it from two
    PaBnFin fin;         // pending BatchNorm finalize of `in`, done in the kernel's prologue (fin.rows > 0; only launches pa_conv_takes_fin() admits)
    int low_prio;        // != 0: a launch of a side branch (skip blocks beside the main chain): the kernel keeps wave priority 0 instead of PA_MAIN_PRIO
    int dbg;             // tuning builds only (0 in the release library): conv3x3_tile.hip phase ablation bits 1 / 2 / 4, bit 8 = per-workgroup tap rotation
    int xcd;             // set by the launchers: workgroup i works on tile (i % 8) * (tiles / 8) + i / 8 (one contiguous range per XCD)
    // round 6 -- the LOW half of the upsample-add backward (reference models/asn_stacked_hg.py:192-203) folded into the 1x1 data gradient that
    // produces d(merged) (pa_conv1x1_tile_up_supported): besides `out` = d(merged) [B][H][W][Cout] the launch stores
    // out2 [B][H/2][W/2][Cout] = ep2(sum of the 2x2 block of d(merged)) with ep2's BatchNorm-backward mask and its two partial reductions
    PaEpilogue ep2;
    bf16* out2;          // nullptr = off
};
// the row-tile instance whose workgroups own 2 image rows x 32 columns and carry the epilogue above
bool pa_conv1x1_tile_up_supported(const PaConvArgs& a);
// stat_rows (optional) receives the number of partial-statistics rows the launch writes (= grid.x)
int pa_launch_conv(const PaConvArgs& a, hipStream_t st, int* stat_rows = nullptr);
// will pa_launch_conv send this launch to a kernel that carries the finalize prologue (generic kernel, 3x3 tile kernel)?
bool pa_conv_takes_fin(const PaConvArgs& a);
// halo-tile 3x3 kernel (conv3x3_tile.hip); pa_launch_conv dispatches to it when the shape is supported
bool pa_conv3x3_tile_supported(const PaConvArgs& a);
bool pa_conv3x3_tile_takes_fin(const PaConvArgs& a);      // ... and the instance it would run exists with the finalize prologue
int pa_launch_conv3x3_tile(const PaConvArgs& a, hipStream_t st, int* stat_rows = nullptr);
// row-tile 1x1 kernel (conv1x1_tile.hip), same dispatch rule
bool pa_conv1x1_tile_supported(const PaConvArgs& a);
int pa_launch_conv1x1_tile(const PaConvArgs& a, hipStream_t st, int* stat_rows = nullptr);
// upper bound of stat rows any conv / elementwise launch writes for a tensor with M pixels
inline int pa_max_stat_rows(int M) { int r = (M + 63) / 64; return r < 512 ? 512 : r; }

// Weight gradient: dw[n][tap][c] = sum_m dy(m)[n] * x(pixel(m)+tap)[c], split over m into `splits`
// deterministic partial slabs part[split][Cout][taps*Cin] (fp32); optional bias-gradient partials
// dbpart[split][Cout] (column sums of dy).
struct PaWgradArgs {
    PaOperand dy;        // [M][Cout]  (PLAIN or LIN2)
    PaOperand x;         // [B][H][W][Cin] (PLAIN or BNRELU)
    float* part;
    float* dbpart;       // or nullptr
    int B, H, W, Cin, Cout, taps, splits;
};
int pa_launch_wgrad(const PaWgradArgs& a, hipStream_t st);
// H, W = 0: spatial shape unknown (generic kernel only)
int pa_wgrad_splits(int M, int H, int W, int Cin, int Cout, int taps);
// tile kernels (conv_wgrad_tile.hip): 0 / -1 = shape not handled there
int pa_wgrad_tile_splits(int B, int H, int W, int Cin, int Cout, int taps);
int pa_launch_stem_wgrad_tile(const PaWgradArgs& a, hipStream_t st);   // -1: not handled
int pa_launch_wgrad_tile(const PaWgradArgs& a, hipStream_t st);
// several independent weight gradients in one launch (conv_wgrad_tile.hip): split count of a layer that will be a job (0 = its shape is not
// taken), may this launch be a job, launch n <= 8 jobs
int pa_wgrad_group_splits(int B, int H, int W, int Cin, int Cout, int taps);
bool pa_wgrad_group_takes(const PaWgradArgs& a);
int pa_wgrad_job_workgroups(const PaWgradArgs& a);          // workgroups of the launch as a job
struct PaWgradReduceJob;
// keep_order: jobs in the caller's order (default: longest first); red_*: red_n <= PA_RED_LIST_MAX entries of the reduce table whose slabs
// (complete on `st` before this launch) are summed by a few more workgroups of the same launch
int pa_launch_wgrad_group(const PaWgradArgs* const* jobs, int n, hipStream_t st, bool keep_order = false,
                          const PaWgradReduceJob* red_jobs = nullptr, const int* red_idx = nullptr, int red_n = 0);
void pa_wgrad_set_launch_flags(unsigned flags);      // hipExtAnyOrderLaunch for the tile weight gradients launched next by this thread (0 = in-order)

// reduce partial slabs into the fp32 gradient in PyTorch layout  dst[n][c][tap]  (real_cin/real_cout
// select the un-padded sub-block for the 16-channel head layers)
struct PaWgradReduceJob {
    const float* part; float* dst; const float* dbpart; float* dbdst;
    int Cout, Cin, taps, splits, real_cout, real_cin;
};
int pa_launch_wgrad_reduce(const PaWgradReduceJob* jobs_dev, int njobs, int max_elems, hipStream_t st);
#define PA_RED_LIST_MAX 24
struct PaRedList { int idx[PA_RED_LIST_MAX]; };
int pa_launch_wgrad_reduce_list(const PaWgradReduceJob* jobs_dev, const int* idx, int n, int max_elems, hipStream_t st);      // any n <= 24 jobs of the table

// ---- BatchNorm bookkeeping
// forward: partial rows stats[rows][C][2] (sum, sumsq; over `count` values in total) -> scale/shift/mean/invstd,
// running stats update
int pa_launch_bn_finalize(const float* stats, int rows, const float* gamma, const float* beta, float* rmean, float* rvar,
                          float* scale, float* shift, float* mean, float* invstd, int C, float count,
                          float momentum, float eps, int update_running, hipStream_t st);
// eval: scale/shift from running stats, for `n` BatchNorms described by a device table
struct PaBnEvalJob { const float* gamma; const float* beta; const float* rmean; const float* rvar; float* scale; float* shift; int C; };
int pa_launch_bn_eval(const PaBnEvalJob* jobs_dev, int njobs, float eps, hipStream_t st);
// backward: partial rows bstats[rows][C][2] (sum dz, sum dz*xhat) -> LIN2 coefficients (A,B,C), dgamma, dbeta
int pa_launch_bn_bwd_finalize2(const float* bs0, int rows0, const float* sc0, const float* mu0, const float* is0, float* kA0, float* kB0, float* kC0,
                               float* dg0, float* db0, int C0, float cnt0,
                               const float* bs1, int rows1, const float* sc1, const float* mu1, const float* is1, float* kA1, float* kB1, float* kC1,
                               float* dg1, float* db1, int C1, float cnt1, hipStream_t st);
int pa_launch_bn_bwd_finalize(const float* bstats, int rows, const float* scale, const float* mean, const float* invstd,
                              float* kA, float* kB, float* kC, float* dgamma, float* dbeta, int C, float count,
                              hipStream_t st);

// ---- pooling / upsampling (NHWC bf16)
int pa_launch_maxpool_fwd(const PaOperand& in, bf16* out, int B, int H, int W, int C, hipStream_t st);
// grad of 2x2 max pool routed to the arg-max (first max in scan order), + optional addend, then epilogue
int pa_launch_maxpool_bwd(const bf16* dout, const PaOperand& in, const PaOperand& add, const PaEpilogue& ep, bf16* din,
                          int B, int H, int W, int C, hipStream_t st, int* stat_rows = nullptr);
// out = nearest_up2(low) + skip        (low: [B][H/2][W/2][C], skip/out: [B][H][W][C])
int pa_launch_upadd_fwd(const PaOperand& low, const PaOperand& skip, bf16* out, int B, int H, int W, int C, hipStream_t st);
// part: 3 = both outputs in one launch; 1 = dlow only, 2 = dskip only (two launches on two streams: pa_upadd_bwd_splits() says whether the
// streaming kernel, the only one with the one-output forms, takes the shape)
int pa_launch_upadd_bwd(const bf16* dout, const PaEpilogue& ep_low, bf16* dlow, const PaEpilogue& ep_skip, bf16* dskip,
                        int B, int H, int W, int C, hipStream_t st, int* stat_rows = nullptr, int part = 3);
bool pa_upadd_bwd_splits(const PaEpilogue& ep_low, const PaEpilogue& ep_skip, int B, int H, int W, int C);

// ---- stem 7x7 stride-2 conv as a K=256 GEMM over the 4-channel-padded NHWC bf16 image
// (in.p / x.p = img4 [B][2H][2W][4], Cin = 256 virtual patch length, Cout = 64, H/W = OUTPUT dims)
int pa_launch_stem_conv(const PaConvArgs& a, hipStream_t st, int* stat_rows = nullptr);
int pa_launch_stem_wgrad(const PaWgradArgs& a, hipStream_t st);
// reduce the stem's partial slabs [splits][64][256] into PyTorch layout dst[64][3][7][7]
int pa_launch_stem_wgrad_reduce(const float* part, int splits, float* dst, hipStream_t st, float* zero64 = nullptr);      // zero64: 64 floats the launch also clears (the stem's bias gradient)

// ---- optimizer / weight preparation
int pa_launch_rmsprop(float* p, const float* g, float* v, size_t n, float lr, float alpha, float eps, float gscale, int* state, hipStream_t st);   // state: the optimizer's own {flag, skipped} device pair or NULL (process-wide pair)
int pa_launch_loss_out(float* acc, float* keep, float* out, float* total, int n, hipStream_t st);      // per-stack losses + their sum out, accumulators cleared
int pa_launch_copy16(void* dst, const void* src, size_t bytes, hipStream_t st, int form = 0);      // plain 16-B/lane streaming copy (bandwidth calibration)
int pa_rmsprop_skipped(const int* state, long long* out, hipStream_t st);     // half-precision build: steps skipped for a non-finite gradient
struct PaPrepJob { const float* w; bf16* wf; bf16* wb; int Cout, Cin, taps, pad_cout, pad_cin; };
int pa_launch_weight_prep(const PaPrepJob* jobs_dev, int njobs, int max_elems, hipStream_t st);

// ---- layout helpers
int pa_launch_nchw_to_nhwc4(const float* src, bf16* dst, int B, int H, int W, hipStream_t st);
int pa_launch_nhwc_to_nchw_f32(const float* src, float* dst, int B, int H, int W, int C, hipStream_t st);
int pa_launch_nchw_f32_to_nhwc_bf16(const float* src, bf16* dst, int B, int C, int H, int W, hipStream_t st);
int pa_launch_nhwc_bf16_to_nchw_f32(const PaOperand& src, float* dst, int B, int C, int H, int W, hipStream_t st);

int pa_launch_cell_mask(const PaOperand& x, const float* mask, const PaEpilogue& ep, bf16* out, int B, int H, int W, int C,
                        hipStream_t st);
int pa_launch_ep_apply(const PaOperand& g, const PaEpilogue& ep, bf16* out, size_t M, int C, hipStream_t st, int* stat_rows = nullptr);
int pa_launch_fill(float* p, float v, size_t n, hipStream_t st);
