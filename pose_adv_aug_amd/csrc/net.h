// Stacked-hourglass pose network + ASN scale/rotation agent as an explicit launch plan over the
// HIP kernels (no autograd, no graph compiler): forward saves exactly what the hand-written backward
// needs.  Mirrors the reference module tree (models/asn_stacked_hg.py) name-for-name so that
// checkpoints interchange.  Internal header; the public C ABI is include/poseadv.h.
#pragma once
#include <string>
#include <vector>
#include "common.h"
#include "kernels.h"
#include "pose_ops.h"

struct Arena {
    char* base = nullptr;
    size_t off = 0;
    void* take(size_t bytes) {
        off = (off + 255) & ~(size_t)255;
        void* p = base ? base + off : nullptr;
        off += bytes;
        return p;
    }
    template <typename T> T* get(size_t n) { return reinterpret_cast<T*>(take(n * sizeof(T))); }
};

struct TensorInfo {          // one state_dict entry
    std::string name;
    int shape[4];
    int ndim;
    size_t offset;           // element offset into the flat parameter (or buffer) array
    size_t numel;
    int is_buffer;           // 0 parameter, 1 running_mean/var buffer, 2 num_batches_tracked (host-side int64)
};

struct BNLayer {
    int C = 0;
    size_t p_gamma = 0, p_beta = 0, b_rmean = 0, b_rvar = 0;
    float *stats = nullptr, *bstats = nullptr;                 // partial rows [rows][C][2] (forward stats / backward reductions)
    int stat_rows = 0, bstat_rows = 0, max_rows = 0;
    float *scale = nullptr, *shift = nullptr, *mean = nullptr, *invstd = nullptr;
    float *kA = nullptr, *kB = nullptr, *kC = nullptr;
    // the finalize of the forward statistics / of the backward reductions has NOT been launched: the next consumer of the tensor does it
    // in its prologue (bn_fin.h; Net::fin_rows_max)
    bool fin_pending = false, bfin_pending = false;
};

struct ConvLayer {
    int Cin = 0, Cout = 0, k = 1;          // real dims
    int pcin = 0, pcout = 0;               // padded (multiple of 64) dims used by the kernels
    size_t p_w = 0, p_b = 0;
    bf16 *wf = nullptr, *wb = nullptr;
    float *part = nullptr, *dbpart = nullptr;
    int splits = 0;
    int red_index = -1;                    // index of this layer's job in the reduce table
    size_t part_floats = 0, db_floats = 0;
    bool has_bn_after = true;
    int taps() const { return k * k; }
};

struct Act {                 // NHWC bf16 activation; `bn` != null means BatchNorm+ReLU is still pending
    bf16* raw = nullptr;
    bf16* grad = nullptr;    // masked gradient dz when bn != null, plain gradient otherwise
    BNLayer* bn = nullptr;
    int B = 0, H = 0, W = 0, C = 0;
    size_t numel() const { return (size_t)B * H * W * C; }
    int M() const { return B * H * W; }
};

struct Net;

// optional per-launch timing of the MFMA kernels with HIP events (bench.py's roofline object)
// classes 0-5: the maps with >= PA_PROF_LOW_M pixels (32 x 32 and larger at batch 24: bandwidth- / MFMA-shaped launches); 9-11: every
// convolution launch below that (16 x 16 ... 4 x 4: bound by launch / staging / epilogue LATENCY, DESIGN.md section 4 -- averaging them
// into the classes above says nothing about either)
enum { PA_PROF_FWD1 = 0, PA_PROF_FWD3, PA_PROF_DGRAD1, PA_PROF_DGRAD3, PA_PROF_WGRAD1, PA_PROF_WGRAD3, PA_PROF_STEM_FWD,
       PA_PROF_STEM_WGRAD, PA_PROF_LOWRES_FWD, PA_PROF_LOW_FWD, PA_PROF_LOW_DGRAD, PA_PROF_LOW_WGRAD, PA_PROF_NCLS };
#define PA_PROF_LOW_M 16384
struct ProfEntry { hipEvent_t e0, e1; int cls; double bytes, flops; };
struct Prof {
    bool on = false;
    std::vector<ProfEntry> entries;
    size_t used = 0;
    std::vector<int> last_seq;            // class of every launch of the last reported pass, in launch order
    ProfEntry* begin(int cls, double bytes, double flops, hipStream_t st);
    void end(ProfEntry* e, hipStream_t st);
    int report(double* out /* [PA_PROF_NCLS][4] = total ms, launches, algorithmic bytes, flops */);
};

struct Residual {
    ConvLayer c1, c2, c3, ad;
    BNLayer b1, b2, b3;
    bool has_adapter = false;
    int cin = 0, cout = 0;
    Act x1, x2, x3;          // raw conv outputs (x3 includes the shortcut)
    bf16* adout = nullptr;   // adapter(x)
    bf16* adgrad = nullptr;  // scratch for the adapter's data gradient
    bf16* dz3 = nullptr;     // BatchNorm-backward gradient of x3, materialised by conv3's data gradient for its later consumers
    bool dz3_valid = false;
    bf16* dz2 = nullptr;     // the same for x2 (stored by conv2's data gradient, read by conv2's weight gradient)
    void declare(Net& n, const std::string& prefix, int cin, int cout, bool adapter);
    void layout(Net& n, Arena& a, int B, int H, int W, bool need_grad);
    int fwd(Net& n, const Act& in);
    int bwd(Net& n, const Act& in, const PaOperand& extra, bool in_needs_grad);
    // round 6: `in` is the upsample-add output merged[k] of an hourglass and low_of_in the low-resolution tensor up[k].x3 under it -- conv1's data
    // gradient (which produces d merged) then also emits d up[k].x3 (2 x 2 sums, mask, reductions) where its tile kernel can
    // (Net::conv_dgrad / pa_conv1x1_tile_up_supported); low_fused says whether it did
    const Act* low_of_in = nullptr;
    bool low_fused = false;
    int b3_rows_scale = 1;   // partial-row capacity of bn3's backward reductions in units of this block's own map (4: the rows may come from a launch over the 2x larger map)
    int bwd_a(Net& n, const Act& in, const PaOperand* extra = nullptr);    // everything except the input gradient (extra: known already -- lets the adapter's data gradient start early)
    bool ad_forked = false;                                            // the adapter's data gradient of this backward pass runs on the side stream
    int bwd_b(Net& n, const Act& in, const PaOperand& extra);           // input gradient (needs `extra`)
};

struct Hourglass {
    Residual down[4], up[4], skip[4], neck;
    Act pooled[4], merged[4];     // pool outputs p_k (k=1..4), upsample-add outputs o_k
    Act skipm[4], neckm;          // occlusion branch only: skip / neck outputs times the 4x4 cell mask (reference :79-100)
    bf16* poolgrad[4] = {nullptr, nullptr, nullptr, nullptr};   // gradient of the pool routed back to its input
    bool low_done[4] = {false, false, false, false};             // d up[k].x3 of this backward pass came out of the data gradient that produced d merged[k]
    void declare(Net& n, const std::string& prefix, int chan);
    void layout(Net& n, Arena& a, int B, int H, int W, bool need_grad);
    int encode(Net& n, const Act& in);
    int decode(Net& n);
    int bwd(Net& n, const Act& in, const PaOperand& extra0);
    const Act& out() const { return merged[0]; }
    bool low_fusable(const Net& n, int k) const;     // may the producer of d merged[k] also emit d up[k].x3 (the skip half then needs the one-output launch)
};

struct Net {
    // configuration
    int stacks = 2, chan = 256, classes = 16, B = 0, res = 256;
    bool is_agent = false;
    // tables
    std::vector<TensorInfo> tensors;           // state_dict order
    size_t n_params = 0, n_buffers = 0;
    std::vector<ConvLayer*> convs;
    std::vector<BNLayer*> bns;
    // bound memory
    float *params = nullptr, *grads = nullptr, *buffers = nullptr;
    char* workspace = nullptr;
    size_t workspace_bytes = 0;
    float *stats_arena = nullptr;
    size_t stats_arena_floats = 0;
    float* loss_dev = nullptr;                 // [stacks] per-stack loss accumulators (atomics of head_fwd), part of the stats arena
    float* loss_keep = nullptr;                // [stacks + 1] the last pose forward pass's losses and their sum (loss_out_kernel)
    float* loss_total_out = nullptr;           // caller's device float that also receives the sum (pa_hg_set_loss_total), or NULL
    bool loss_self_clearing = false;           // pose nets: the accumulators are cleared by loss_out_kernel, begin_step() skips its memset
    // device job tables
    PaPrepJob* prep_jobs = nullptr; int n_prep = 0, prep_max = 0;
    PaWgradReduceJob* red_jobs = nullptr; int n_red = 0, red_max = 0;
    // ONE slab buffer shared by all layers: each weight-gradient launch is reduced right away, while its slabs are
    // still in the 256 MB Infinity Cache, and the next layer overwrites the same lines
    float* shared_part = nullptr; float* shared_db = nullptr; size_t shared_part_floats = 0, shared_db_floats = 0;
    bool immediate_reduce = false;     // measured on MI355X: +0.55 ms/step (100 extra launches) vs one deferred reduce, so off
    PaBnEvalJob* bneval_jobs = nullptr; int n_bneval = 0;
    // run state
    hipStream_t st = nullptr;
    // side streams: the skip branch of hourglass level k runs on side[k] next to the low-resolution path
    // (tiny, latency-bound kernels that leave most CUs idle); fork/join by events, joined before the call returns
    hipStream_t side[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_fork[4], ev_join[4];
    bool streams_ready = false, multi_stream = true;
    // distinct side streams (PA_SIDE_STREAMS); level k uses side[k % n_side].  ONE stream for all four skip branches measured
    // best (3128 vs 3090 img/s with four): main + side + weight-gradient stream = 3 hardware queues (the runtime has 4)
    int n_side = 1;
    int fork_mask = 0xF;                       // bit k: hourglass level k forks its skip block (PA_FORK_LEVELS)
    bool forks(int k) const { return multi_stream && ((fork_mask >> k) & 1); }
    // adapter convolutions of residual1 / residual3 on the side stream beside the block's main path (only outside a side branch, with the
    // streams on).  Measured round 5: 5.917 vs 5.892 ms with them on the main chain (three interleaved pairs) -- the two event pairs per
    // block cost more than the 38 + 78 us of hidden kernels return beside the weight-gradient queue; OFF, switch in tuning builds
    bool adapter_parallel() const {
        static int on = -1;
        if (on < 0) { const char* e = pa_getenv("PA_ADAPTER_PAR"); on = e ? atoi(e) : 0; }
        return on && multi_stream && side[0] != nullptr && !on_side;
    }
    // weight-gradient launches only feed the slab reducer at the end of the backward pass: they run on their own
    // stream behind an event recorded where their operands are final, off the dgrad / BatchNorm critical chain
    hipStream_t wstream = nullptr;
    hipStream_t wstreams[4] = {nullptr, nullptr, nullptr, nullptr}; int n_w = 1, w_rr = 0;   // wstream == wstreams[0]; groups go round robin
    hipEvent_t ev_wdone_x[4];
    hipEvent_t ev_w[16]; int ev_w_next = 0;
    hipEvent_t ev_wdone;
    // weight-gradient launches are collected and flushed in groups: ONE event record on the producing stream per group
    // (an event record between two kernels of a queue costs a 15-20 us bubble on it: ~100 records per step were 1.5 ms)
    struct PendingWgrad { PaWgradArgs a; int cls; double bytes, flops; bool stem; ConvLayer* c; };
    // the slabs of a group are summed right behind it on the weight-gradient stream (part of them still in the 256 MB
    // Infinity Cache) instead of by one 280 us reduction on the main stream at the end of the step: +1.1 % (PA_WREDUCE_LATE = old)
    bool reduce_early = true;
    std::vector<PendingWgrad> pending_wgrads;
    int flush_every = 1, flush_ctr = 0; bool on_side = false;   // main-stream blocks flush every flush_every-th time (PA_WFLUSH_EVERY)
    int flush_wgrads();                        // record on `st`, make wstream wait, launch the collected weight gradients
    std::vector<int> red_stash; int red_stash_mx = 0, red_stash_flushes = 0;      // reduce-table indices of flushed layers whose slabs are not summed yet
    int flush_red_stash(hipStream_t ws);       // one reduction launch for the stash
    // hold the weight gradients of an hourglass' high-resolution levels back until its backward pass reaches level `hold_level`
    // (the low-resolution stretch, where the main chain leaves the GPU almost empty): 0 = off
    int hold_level = 0; bool hold = false;
    int release_held(int k);
    int ensure_streams();
    void release_streams();                    // destroys the side streams / events (pa_net_destroy)
    int fork_to(int k);                        // side[k] waits for everything enqueued on st so far
    int record_join(int k);                    // mark the end of the work enqueued on side[k]
    int wait_join(int k);                      // st waits for that mark
    Prof prof;
    // bytes this design moves (pa_net_design_bytes): operands of every launch since begin_step(), counted once per launch
    double dbytes_rd = 0, dbytes_wr = 0;
    static double opb(const PaOperand& o, double elems) { return o.mode == PA_LD_NONE ? 0.0 : (o.mode == PA_LD_LIN2 ? 4.0 : 2.0) * elems; }
    void cnt(double rd, double wr) { dbytes_rd += rd; dbytes_wr += wr; }
    // forward + loss + backward of the training step captured ONCE into a HIP graph (fork / join events of the side and
    // weight-gradient streams become graph edges) and replayed: pa_hg_train_step with use_graph
    hipGraph_t step_graph = nullptr; hipGraphExec_t step_exec = nullptr; int step_key = -1;
    int train_step_graph(bool train);
    void release_graph();
    bool train_bn = true;
    int bn_update = 1;                         // 0: use batch statistics without touching the running estimates
    float momentum = 0.1f, eps = 1e-5f;

    // ---- pose net modules
    ConvLayer stem_conv; BNLayer stem_bn;
    Residual res1, res2, res3;
    std::vector<Hourglass> hg;
    std::vector<Residual> post;
    std::vector<ConvLayer> lin; std::vector<BNLayer> lin_bn;
    std::vector<ConvLayer> outc, forth, inc;
    bf16* img4 = nullptr;
    Act a0, pool0;
    bf16* pool0grad_unused = nullptr;
    std::vector<Act> lin_out, xin;                 // xin[i] = input of stack i
    std::vector<float*> heat; std::vector<bf16*> heat64, dheat64, dheat_in, forth_tmp, lgrad_tmp;
    double* pts_dev = nullptr;                     // [B][16][2] heat-map coords (caller provided per step)
    std::vector<float*> heat_peak;                 // [B][16][2] arg-max of heat[i], computed once per forward on demand (accuracy AND PCKh use it)
    std::vector<char> heat_peak_valid;
    int heat_argmax(int stack, const float** out, hipStream_t on = nullptr); // launches the arg-max unless this forward's result exists
    // The meters of a training step (pa_hg_accuracy / pa_hg_pckh: ~8 short launches that only READ the forward pass's heat maps) on a stream of
    // their own beside the backward pass (pa_net_meters_async): forked behind the last launch of the main stream, joined by the main stream
    // at the end of the backward pass (or in front of the next forward pass), so every later reader on the main stream is ordered behind them.
    hipStream_t mstream = nullptr; hipEvent_t ev_mfork = nullptr, ev_meter = nullptr;
    bool meters_async = false, meter_pending = false;
    bool capturing = false;                     // inside train_step_graph's stream capture
    int meter_stream(hipStream_t* out);         // the stream the meters are launched on now (forks mstream from st when meters_async)
    int meter_done(hipStream_t ms);             // records the join event when ms is the meter stream
    int join_meters();                          // the main stream waits for the pending meters
    // occlusion (dropout) branch, reference :172-190: [B][16] cell masks (caller-owned device memory) applied to the neck and the
    // four skip tensors of every stack in forward_pose / backward_pose; nullptr = off
    const float* drop_mask = nullptr;

    // ---- declaration helpers
    size_t add_param(const std::string& name, std::initializer_list<int> shape);
    size_t add_buffer(const std::string& name, int C);
    void declare_conv(ConvLayer& c, const std::string& name, int cin, int cout, int k, bool bn_after);
    void declare_bn(BNLayer& b, const std::string& name, int C);
    void layout_conv(ConvLayer& c, Arena& a, int M, int H = 0, int W = 0);
    void layout_bn(BNLayer& b, Arena& a, int M);
    void layout_shared(Arena& a);                  // call last in every layout pass
    Act new_act(Arena& a, int B, int H, int W, int C, BNLayer* bn, bool need_grad);

    void declare_pose();
    size_t layout_all(char* base);
    int upload_tables();

    // ---- runtime helpers
    PaOperand op(const Act& a) const;               // value of an activation (BNRELU pending or PLAIN)
    PaOperand gradop(const Act& a) const;           // gradient w.r.t. the raw tensor (LIN2 or PLAIN)
    PaEpilogue final_ep(const Act& a) const;        // epilogue that finishes a gradient for `a`
    int finish_grad(const Act& a);                  // BatchNorm backward finalize (if pending BN)
    int finish_grad2(const Act& a, const Act& b);   // two of them in one launch
    // pending_in: the BatchNorm of `in` whose finalize may still be pending (done in this launch's prologue then);
    // defer_after: the caller guarantees that the ONLY next reader of bn_after's constants is a launch that takes a pending finalize
    // (fin_consumer_ok), so the finalize launch is skipped when the statistics have <= fin_rows_max partial rows
    int conv_fwd(ConvLayer& c, const PaOperand& in, int B, int H, int W, const PaOperand& add1, const PaOperand& add2,
                 bf16* out, BNLayer* bn_after, BNLayer* pending_in = nullptr, bool defer_after = false);
    int conv_dgrad(ConvLayer& c, const PaOperand& dy, int B, int H, int W, const PaOperand& add1, const PaOperand& add2,
                   const PaEpilogue& ep, bf16* out, bf16* dz_out = nullptr, bool* dz_done = nullptr, BNLayer* pending_in = nullptr,
                   const Act* low = nullptr, bool* low_done = nullptr);      // low: also emit the upsample-add backward's low-resolution output (Residual::low_of_in)
    // BatchNorm finalize in the consumer's prologue for launches with at most this many partial rows (the 16 x 16 and smaller levels;
    // 0 = always a launch of its own).  Same bits either way (bn_fin.h)
    int fin_rows_max = PA_FIN_SMALL_ROWS;
    int fin_mask = 7;       // which finalizes may ride in a consumer: 1 forward x1 / x2, 2 backward x2, 4 backward x3 (PA_FIN_MASK in tuning builds)
    bool fin_consumer_ok(const ConvLayer& c, int B, int H, int W, bool dgrad) const;
    PaBnFin fin_fwd(const BNLayer& b, int M) const;
    PaBnFin fin_bwd(const BNLayer& b, int M) const;
    int finish_grad_or_defer(const Act& a, bool defer);
    int finish_grad2_or_defer(const Act& a, bool defer_a, const Act& b, bool defer_b);
    bool x3_fin_ok(const Residual& r, const Act& in) const;     // can r.x3's backward finalize ride in conv3's data gradient?
    int conv_wgrad(ConvLayer& c, const PaOperand& dy, const PaOperand& x, int B, int H, int W);

    int prepare_weights();
    int begin_step();
    const bf16* cur_image = nullptr;
    int forward_pose(const float* img_nchw, const bf16* img4_in, const double* pts, bool train, float* loss_out_dev);
    int forward_pose_body(const float* img_nchw, const bf16* img4_in, const double* pts, bool train, float* loss_out_dev);
    int backward_pose();
    // the backward pass in phases (data-parallel gradient exchange of a finished stack while the earlier ones still run):
    // phase p < stacks = head layers + post block + hourglass of stack (stacks-1-p); phase == stacks = stem + final reductions
    int backward_stack(int i);
    int backward_stem();
    hipEvent_t ev_bucket[16], ev_bucket_main = nullptr; bool bucket_events = false;
    int mark_bucket(int stack);                     // record "every gradient of hg.<stack> is final" (after its slabs are reduced)
    int reduce_grads();
    int forward_half(const float* img_nchw, const bf16* img4_in, bool train);     // stem + hg[0] down path (agent features)

    // ---- ASN scale/rotation agent (is_agent == true): reference models/asn_stacked_hg.py:349-439
    int scale_num = 7, rot_num = 7;
    Residual asn_in[5];                            // residual_skip1..4, residual_neck
    Residual asn_merge[4];
    Residual asn_deep[3];
    size_t p_fcs_w = 0, p_fcs_b = 0, p_fcr_w = 0, p_fcr_b = 0;
    Act asn_pa[4];                                 // maxpool(previous) + lower feature
    float *asn_feat = nullptr, *asn_logits = nullptr, *asn_probs = nullptr, *asn_dlogits = nullptr;   // [B][C], [B][2K] ...
    void declare_asn();
    size_t layout_asn(char* base);
    int asn_forward(Net& pose, bool train, float* logits_s, float* logits_r);
    float asn_log_eps = 1e-7f;                 // log(softmax + eps) of the agent's KL loss
    int asn_backward(Net& pose, const float* target_s, const float* target_r, float* loss_out);
    int asn_backward_trunk(Net& pose, const bf16* dact);     // from the gradient of deep_merge's output down to every trunk parameter
    // occlusion agent (create_asn(is_dropout=True), reference :378-379,437-439): out_conv 1x1 chan -> 1 on the 4x4 map
    bool asn_dropout = false;
    size_t p_oc_w = 0, p_oc_b = 0;
    int asn_forward_trunk(Net& pose, bool train, const Act** top);
    int asn_forward_masks(Net& pose, bool train, float* mask_logits);
    int asn_backward_masks(Net& pose, const float* dlogits);
};

PaOperand pa_plain(const bf16* p);
PaOperand pa_none();
