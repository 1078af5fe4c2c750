// Launch plan of the stacked-hourglass pose network (reference models/asn_stacked_hg.py:11-347):
// declaration of the parameter table in the reference's state_dict order, workspace layout, forward,
// hand-written backward and gradient reduction.  See net.h.
#include "net.h"
#include <stdio.h>
#include <string.h>

// (the round-1 timing experiment that removed cross-stream waits -- wrong results by design -- is gone from the library)
static const int g_nosync = 0;
static bool nosync() { return false; }
// PA_ABLATE=<mask> (tuning builds only; results are WRONG, timing bounds only): 1 no slab reductions, 2 no weight-gradient
// launches, 4 no BatchNorm finalize launches, 8 no 3x3 convolutions, 16 no 1x1 data gradients; 1024 no upsample-add backward, 2048 no
// pooling backward, 4096 no pooling / upsample-add forward (bounds of what fusing them into their neighbours can return)
static int ablate() { static int v = -1; if (v < 0) { const char* e = pa_getenv("PA_ABLATE"); v = e ? atoi(e) : 0; } return v; }
// bits 128 / 256 / 512: forward convolutions + finalize / data gradients + backward finalize / weight gradients of maps up to
// PA_ABLATE_H (default 8) pixels high are skipped: what the low-resolution stretch costs the step at most (wrong results)
static int ablate_h() { static int v = -1; if (v < 0) { const char* e = pa_getenv("PA_ABLATE_H"); v = e ? atoi(e) : 8; } return v; }
#define PA_STEM_ON_MAIN_DEFAULT 1
#define PA_WG_GROUP_CAP_DEFAULT 256
#define PA_WREDUCE_LAG_DEFAULT 2
#define PA_UPADD_SPLIT_DEFAULT 1
#define PA_UPADD_FUSE_DEFAULT 1
#define PA_WG_RED_IN_GROUP_DEFAULT 1
#define TRY(x) do { int _r = (x); if (_r) return _r; } while (0)

PaOperand pa_plain(const bf16* p) { PaOperand o; o.p = p; o.q = nullptr; o.k0 = o.k1 = o.k2 = nullptr; o.mode = PA_LD_PLAIN; return o; }
PaOperand pa_none() { PaOperand o; o.p = o.q = nullptr; o.k0 = o.k1 = o.k2 = nullptr; o.mode = PA_LD_NONE; return o; }
static PaEpilogue ep_plain() { PaEpilogue e; memset(&e, 0, sizeof e); e.mode = PA_OUT_PLAIN; return e; }
static int round64(int v) { return (v + 63) / 64 * 64; }

// ------------------------------------------------------------------------------------------------
// declaration
size_t Net::add_param(const std::string& name, std::initializer_list<int> shape) {
    TensorInfo t; t.name = name; t.ndim = (int)shape.size(); t.is_buffer = 0;
    size_t n = 1; int i = 0;
    for (int s : shape) { t.shape[i++] = s; n *= (size_t)s; }
    for (; i < 4; ++i) t.shape[i] = 1;
    n_params = (n_params + 3) & ~(size_t)3;           // 16-byte aligned tensors (float4 bias loads)
    t.offset = n_params; t.numel = n;
    n_params += n;
    tensors.push_back(t);
    return t.offset;
}

size_t Net::add_buffer(const std::string& name, int C) {
    TensorInfo t; t.name = name; t.ndim = 1; t.shape[0] = C; t.shape[1] = t.shape[2] = t.shape[3] = 1; t.is_buffer = 1;
    t.offset = n_buffers; t.numel = (size_t)C;
    n_buffers += (size_t)C;
    tensors.push_back(t);
    return t.offset;
}

void Net::declare_conv(ConvLayer& c, const std::string& name, int cin, int cout, int k, bool bn_after) {
    c.Cin = cin; c.Cout = cout; c.k = k; c.has_bn_after = bn_after;
    c.pcin = (k == 7) ? 256 : round64(cin);
    c.pcout = round64(cout);
    c.p_w = add_param(name + ".weight", {cout, cin, k, k});
    c.p_b = add_param(name + ".bias", {cout});
    convs.push_back(&c);
}

void Net::declare_bn(BNLayer& b, const std::string& name, int C) {
    b.C = C;
    b.p_gamma = add_param(name + ".weight", {C});
    b.p_beta = add_param(name + ".bias", {C});
    b.b_rmean = add_buffer(name + ".running_mean", C);
    b.b_rvar = add_buffer(name + ".running_var", C);
    TensorInfo t; t.name = name + ".num_batches_tracked"; t.ndim = 0; t.shape[0] = t.shape[1] = t.shape[2] = t.shape[3] = 1;
    t.is_buffer = 2; t.offset = 0; t.numel = 1;
    tensors.push_back(t);
    bns.push_back(&b);
}

void Residual::declare(Net& n, const std::string& p, int cin_, int cout_, bool adapter) {
    cin = cin_; cout = cout_; has_adapter = adapter;
    const int mid = cout / 2;
    n.declare_conv(c1, p + "conv1", cin, mid, 1, true);  n.declare_bn(b1, p + "bn1", mid);
    n.declare_conv(c2, p + "conv2", mid, mid, 3, true);  n.declare_bn(b2, p + "bn2", mid);
    n.declare_conv(c3, p + "conv3", mid, cout, 1, true); n.declare_bn(b3, p + "bn3", cout);
    if (adapter) n.declare_conv(ad, p + "adapter", cin, cout, 1, true);     // feeds bn3: bias gradient is zero
}

void Hourglass::declare(Net& n, const std::string& p, int chan) {
    // registration order of the reference (:56-68); num_modules == 1 => "<site>.0."
    const char* dn[4] = {"down1", "down2", "down3", "down4"};
    const char* un[4] = {"up1", "up2", "up3", "up4"};
    const char* sn[4] = {"skip1", "skip2", "skip3", "skip4"};
    for (int k = 0; k < 4; ++k) down[k].declare(n, p + dn[k] + ".0.", chan, chan, false);
    for (int k = 0; k < 4; ++k) up[k].declare(n, p + un[k] + ".0.", chan, chan, false);
    for (int k = 0; k < 4; ++k) skip[k].declare(n, p + sn[k] + ".0.", chan, chan, false);
    neck.declare(n, p + "neck.0.", chan, chan, false);
}

void Net::declare_pose() {
    declare_conv(stem_conv, "conv1", 3, 64, 7, true);
    declare_bn(stem_bn, "bn1", 64);
    res1.declare(*this, "residual1.", 64, 128, true);
    res2.declare(*this, "residual2.", 128, 128, false);
    res3.declare(*this, "residual3.", 128, chan, true);
    hg.resize(stacks); post.resize(stacks); lin.resize(stacks); lin_bn.resize(stacks); outc.resize(stacks);
    forth.resize(stacks > 0 ? stacks - 1 : 0); inc.resize(stacks > 0 ? stacks - 1 : 0);
    char buf[64];
    for (int i = 0; i < stacks; ++i) { snprintf(buf, sizeof buf, "hg.%d.", i); hg[i].declare(*this, buf, chan); }
    for (int i = 0; i < stacks; ++i) { snprintf(buf, sizeof buf, "post_res.%d.0.", i); post[i].declare(*this, buf, chan, chan, false); }
    for (int i = 0; i < stacks; ++i) {
        snprintf(buf, sizeof buf, "linear.%d.0", i); declare_conv(lin[i], buf, chan, chan, 1, true);
        snprintf(buf, sizeof buf, "linear.%d.1", i); declare_bn(lin_bn[i], buf, chan);
    }
    for (int i = 0; i < stacks; ++i) { snprintf(buf, sizeof buf, "out_conv.%d", i); declare_conv(outc[i], buf, chan, classes, 1, false); }
    for (int i = 0; i + 1 < stacks; ++i) { snprintf(buf, sizeof buf, "forth_conv.%d", i); declare_conv(forth[i], buf, chan, chan, 1, false); }
    for (int i = 0; i + 1 < stacks; ++i) { snprintf(buf, sizeof buf, "in_conv.%d", i); declare_conv(inc[i], buf, classes, chan, 1, false); }
    n_params = (n_params + 3) & ~(size_t)3;
}

// ------------------------------------------------------------------------------------------------
// workspace layout
void Net::layout_conv(ConvLayer& c, Arena& a, int M, int H, int W) {
    if (H <= 0 && B > 0) {                     // the net's own layers are square maps
        const int hw = M / B;
        int h = 1; while (h * h < hw) ++h;
        if (h * h == hw && hw * B == M) H = W = h;
    }
    const size_t wn = (size_t)c.pcout * (c.k == 7 ? 1 : c.taps()) * c.pcin;
    c.wf = a.get<bf16>(wn);
    c.wb = (c.k == 7) ? nullptr : a.get<bf16>(wn);
    c.splits = c.k == 7 ? pa_wgrad_splits(M, 0, 0, c.pcin, c.pcout, 1) : pa_wgrad_splits(M, H, W, c.pcin, c.pcout, c.taps());
    if (c.k != 7 && H > 0 && B > 0 && M == B * H * W) {          // a layer the grouped weight-gradient kernel takes: few, long splits
        const int gs = pa_wgrad_group_splits(B, H, W, c.pcin, c.pcout, c.taps());
        if (gs > 0) c.splits = gs;
    }
    if (c.k == 7) {                              // stem: 64 KB slabs, so two workgroups per CU (their load / MFMA phases overlap)
        static int ss = -1;
        if (ss < 0) { const char* e = pa_getenv("PA_STEM_SPLITS"); ss = e ? atoi(e) : 512; }       // 6.96 vs 7.00 ms (256)
        if (ss > 0 && ss <= (M + 127) / 128) c.splits = ss;
    }
    c.part_floats = (size_t)c.splits * wn;
    c.db_floats = c.has_bn_after ? 0 : (size_t)c.splits * c.pcout;
    if (immediate_reduce) {                   // shared slab: remember the largest request, bind after the layout pass
        if (c.part_floats > shared_part_floats) shared_part_floats = c.part_floats;
        if (c.db_floats > shared_db_floats) shared_db_floats = c.db_floats;
        c.part = nullptr; c.dbpart = nullptr;
    } else {
        c.part = a.get<float>(c.part_floats);
        c.dbpart = c.db_floats ? a.get<float>(c.db_floats) : nullptr;
    }
}

void Net::layout_shared(Arena& a) {
    if (!immediate_reduce) return;
    shared_part = a.get<float>(shared_part_floats);
    shared_db = a.get<float>(shared_db_floats > 0 ? shared_db_floats : 1);
}

void Net::layout_bn(BNLayer& b, Arena& a, int M) {
    b.max_rows = pa_max_stat_rows(M);
    b.stats = a.get<float>((size_t)b.max_rows * b.C * 2);
    b.bstats = a.get<float>((size_t)b.max_rows * b.C * 2);
    b.scale = a.get<float>(b.C); b.shift = a.get<float>(b.C); b.mean = a.get<float>(b.C); b.invstd = a.get<float>(b.C);
    b.kA = a.get<float>(b.C); b.kB = a.get<float>(b.C); b.kC = a.get<float>(b.C);
}

Act Net::new_act(Arena& a, int B_, int H, int W, int C, BNLayer* bn, bool need_grad) {
    Act t; t.B = B_; t.H = H; t.W = W; t.C = C; t.bn = bn;
    t.raw = a.get<bf16>(t.numel());
    t.grad = need_grad ? a.get<bf16>(t.numel()) : nullptr;
    return t;
}

void Residual::layout(Net& n, Arena& a, int B, int H, int W, bool need_grad) {
    const int M = B * H * W, mid = cout / 2;
    n.layout_conv(c1, a, M, H, W); n.layout_conv(c2, a, M, H, W); n.layout_conv(c3, a, M, H, W);      // explicit map size: the split counts depend on it
    n.layout_bn(b1, a, M); n.layout_bn(b2, a, M); n.layout_bn(b3, a, M * b3_rows_scale);
    x1 = n.new_act(a, B, H, W, mid, &b1, need_grad);
    x2 = n.new_act(a, B, H, W, mid, &b2, need_grad);
    x3 = n.new_act(a, B, H, W, cout, &b3, need_grad);
    dz3 = need_grad ? a.get<bf16>(x3.numel()) : nullptr;
    dz2 = need_grad ? a.get<bf16>(x2.numel()) : nullptr;
    if (has_adapter) {
        n.layout_conv(ad, a, M, H, W);
        adout = a.get<bf16>((size_t)M * cout);
        adgrad = need_grad ? a.get<bf16>((size_t)M * cin) : nullptr;
    }
}

void Hourglass::layout(Net& n, Arena& a, int B, int H, int W, bool need_grad) {
    const int C = neck.cout;
    for (int k = 0; k < 4; ++k) {
        skip[k].layout(n, a, B, H >> k, W >> k, need_grad);
        pooled[k] = n.new_act(a, B, H >> (k + 1), W >> (k + 1), C, nullptr, need_grad);
        down[k].layout(n, a, B, H >> (k + 1), W >> (k + 1), need_grad);
        poolgrad[k] = need_grad ? a.get<bf16>((size_t)B * (H >> k) * (W >> k) * C) : nullptr;
    }
    neck.layout(n, a, B, H >> 4, W >> 4, need_grad);
    for (int k = 3; k >= 0; --k) {
        up[k].b3_rows_scale = 4;          // (its backward reductions can come from the data gradient over merged[k], 4 x the pixels: Residual::low_of_in)
        up[k].layout(n, a, B, H >> (k + 1), W >> (k + 1), need_grad);
        merged[k] = n.new_act(a, B, H >> k, W >> k, C, nullptr, need_grad);
    }
    for (int k = 0; k < 4; ++k) skipm[k] = n.new_act(a, B, H >> k, W >> k, C, nullptr, need_grad);
    neckm = n.new_act(a, B, H >> 4, W >> 4, C, nullptr, need_grad);
}

size_t Net::layout_all(char* base) {
    Arena a; a.base = base;
    // region zeroed at the start of every step: BatchNorm statistic accumulators + per-stack loss
    size_t zero_begin = a.off;
    loss_dev = a.get<float>(64);
    a.take(0);
    stats_arena = base ? reinterpret_cast<float*>(base + zero_begin) : nullptr;
    stats_arena_floats = (a.off - zero_begin) / sizeof(float);
    loss_keep = a.get<float>(64);
    loss_self_clearing = true;                 // (the workspace is zero-filled when it is bound: the first pass starts from zero too)
    // job tables
    prep_jobs = a.get<PaPrepJob>(convs.size());
    red_jobs = a.get<PaWgradReduceJob>(convs.size());
    bneval_jobs = a.get<PaBnEvalJob>(bns.size());

    const int H2 = res / 2, H4 = res / 4;
    img4 = a.get<bf16>((size_t)B * res * res * 4);
    pts_dev = a.get<double>((size_t)B * classes * 2);
    layout_conv(stem_conv, a, B * H2 * H2);
    layout_bn(stem_bn, a, B * H2 * H2);
    a0 = new_act(a, B, H2, H2, 64, &stem_bn, true);
    res1.layout(*this, a, B, H2, H2, true);
    pool0 = new_act(a, B, H4, H4, 128, nullptr, true);
    res2.layout(*this, a, B, H4, H4, true);
    res3.layout(*this, a, B, H4, H4, true);
    lin_out.resize(stacks); xin.resize(stacks); heat.resize(stacks); heat_peak.resize(stacks); heat_peak_valid.assign(stacks, 0); heat64.resize(stacks); dheat64.resize(stacks);
    dheat_in.resize(stacks); forth_tmp.resize(stacks); lgrad_tmp.resize(stacks);
    const int M = B * H4 * H4;
    for (int i = 0; i < stacks; ++i) {
        hg[i].layout(*this, a, B, H4, H4, true);
        post[i].layout(*this, a, B, H4, H4, true);
        layout_conv(lin[i], a, M, H4, H4); layout_bn(lin_bn[i], a, M);
        lin_out[i] = new_act(a, B, H4, H4, chan, &lin_bn[i], true);
        layout_conv(outc[i], a, M, H4, H4);
        heat[i] = a.get<float>((size_t)M * 16);
        heat_peak[i] = a.get<float>((size_t)B * 16 * 2);
        heat64[i] = a.get<bf16>((size_t)M * 64);        // channels 16..63 stay zero (workspace is zero-filled once)
        dheat64[i] = a.get<bf16>((size_t)M * 64);
        dheat_in[i] = a.get<bf16>((size_t)M * 64);
        lgrad_tmp[i] = a.get<bf16>((size_t)M * chan);
        if (i + 1 < stacks) {
            layout_conv(forth[i], a, M, H4, H4); layout_conv(inc[i], a, M, H4, H4);
            forth_tmp[i] = a.get<bf16>((size_t)M * chan);
        }
        if (i == 0) xin[0] = res3.x3;
        if (i + 1 < stacks) xin[i + 1] = new_act(a, B, H4, H4, chan, nullptr, true);
    }
    layout_shared(a);
    a.take(0);
    return a.off;
}

int Net::upload_tables() {
    std::vector<PaPrepJob> pj; std::vector<PaWgradReduceJob> rj; std::vector<PaBnEvalJob> bj;
    prep_max = 0; red_max = 0;
    for (ConvLayer* c : convs) {
        if (immediate_reduce) { c->part = shared_part; c->dbpart = c->db_floats ? shared_db : nullptr; }
        PaPrepJob p; p.w = params + c->p_w; p.wf = c->wf; p.wb = c->wb; p.Cout = c->Cout; p.Cin = c->Cin;
        p.taps = c->k == 7 ? 49 : c->taps(); p.pad_cout = c->pcout; p.pad_cin = c->pcin;
        pj.push_back(p);
        int pe = c->pcout * (c->k == 7 ? 256 : c->taps() * c->pcin);
        if (pe > prep_max) prep_max = pe;
        if (c->k == 7) continue;
        PaWgradReduceJob r; r.part = c->part; r.dst = grads + c->p_w; r.dbpart = c->dbpart; r.dbdst = grads + c->p_b;
        r.Cout = c->pcout; r.Cin = c->pcin; r.taps = c->taps(); r.splits = c->splits; r.real_cout = c->Cout; r.real_cin = c->Cin;
        c->red_index = (int)rj.size();
        rj.push_back(r);
        int re = c->Cout * c->Cin * c->taps() + c->Cout;
        if (re > red_max) red_max = re;
    }
    for (BNLayer* b : bns) {
        PaBnEvalJob j; j.gamma = params + b->p_gamma; j.beta = params + b->p_beta; j.rmean = buffers + b->b_rmean;
        j.rvar = buffers + b->b_rvar; j.scale = b->scale; j.shift = b->shift; j.C = b->C;
        bj.push_back(j);
    }
    n_prep = (int)pj.size(); n_red = (int)rj.size(); n_bneval = (int)bj.size();
    PA_CHECK(hipMemcpyAsync(prep_jobs, pj.data(), pj.size() * sizeof(PaPrepJob), hipMemcpyHostToDevice, st));
    PA_CHECK(hipMemcpyAsync(red_jobs, rj.data(), rj.size() * sizeof(PaWgradReduceJob), hipMemcpyHostToDevice, st));
    PA_CHECK(hipMemcpyAsync(bneval_jobs, bj.data(), bj.size() * sizeof(PaBnEvalJob), hipMemcpyHostToDevice, st));
    PA_CHECK(hipStreamSynchronize(st));
    return 0;
}

// ------------------------------------------------------------------------------------------------
// per-launch HIP-event timing (off unless bench.py asks for it)
ProfEntry* Prof::begin(int cls, double bytes, double flops, hipStream_t st) {
    if (!on) return nullptr;
    if (used == entries.size()) {
        ProfEntry e; e.cls = 0; e.bytes = e.flops = 0;
        if (hipEventCreate(&e.e0) != hipSuccess || hipEventCreate(&e.e1) != hipSuccess) return nullptr;
        entries.push_back(e);
    }
    ProfEntry* e = &entries[used++];
    e->cls = cls; e->bytes = bytes; e->flops = flops;
    hipEventRecord(e->e0, st);
    return e;
}

void Prof::end(ProfEntry* e, hipStream_t st) { if (e) hipEventRecord(e->e1, st); }

int Prof::report(double* out) {
    for (int i = 0; i < PA_PROF_NCLS * 4; ++i) out[i] = 0.0;
    last_seq.clear();
    for (size_t i = 0; i < used; ++i) last_seq.push_back(entries[i].cls);
    for (size_t i = 0; i < used; ++i) {
        ProfEntry& e = entries[i];
        hipError_t r = hipEventSynchronize(e.e1);
        if (r != hipSuccess) return (int)r;
        float ms = 0.f;
        r = hipEventElapsedTime(&ms, e.e0, e.e1);
        if (r != hipSuccess) return (int)r;
        out[e.cls * 4 + 0] += (double)ms; out[e.cls * 4 + 1] += 1.0; out[e.cls * 4 + 2] += e.bytes; out[e.cls * 4 + 3] += e.flops;
    }
    used = 0;
    return 0;
}

// algorithmic work of one conv launch: every activation element read once / written once (2-byte
// elements), weights once; 2*M*N*K flops
static void conv_work(int M, int cin, int cout, int taps, bool wgrad, double& bytes, double& flops) {
    bytes = 2.0 * M * ((double)cin + cout) + (wgrad ? 4.0 : 2.0) * (double)cout * taps * cin;
    flops = 2.0 * M * (double)cout * taps * cin;
}

// ------------------------------------------------------------------------------------------------
// runtime helpers
PaOperand Net::op(const Act& a) const {
    PaOperand o = pa_plain(a.raw);
    if (a.bn) { o.mode = PA_LD_BNRELU; o.k0 = a.bn->scale; o.k1 = a.bn->shift; }
    return o;
}

PaOperand Net::gradop(const Act& a) const {
    PaOperand o = pa_plain(a.grad);
    if (a.bn) { o.mode = PA_LD_LIN2; o.q = a.raw; o.k0 = a.bn->kA; o.k1 = a.bn->kB; o.k2 = a.bn->kC; }
    return o;
}

PaEpilogue Net::final_ep(const Act& a) const {
    PaEpilogue e = ep_plain();
    if (a.bn) {
        e.mode = PA_OUT_BWD; e.stats = a.bn->bstats; e.rows_out = &a.bn->bstat_rows; e.xref = a.raw; e.scale = a.bn->scale; e.shift = a.bn->shift;
        e.mean = a.bn->mean; e.invstd = a.bn->invstd;
    }
    return e;
}

int Net::finish_grad(const Act& a) {
    if (!a.bn) return 0;
    if (ablate() & 4) return 0;
    if ((ablate() & 256) && a.H <= ablate_h()) return 0;
    BNLayer* b = a.bn;
    return pa_launch_bn_bwd_finalize(b->bstats, b->bstat_rows, b->scale, b->mean, b->invstd, b->kA, b->kB, b->kC, grads + b->p_gamma,
                                     grads + b->p_beta, b->C, (float)a.M(), st);
}

// defer: the only next reader of a's BatchNorm-backward constants is a data-gradient launch that takes the pending finalize
int Net::finish_grad_or_defer(const Act& a, bool defer) {
    if (a.bn && defer && a.bn->bstat_rows > 0 && a.bn->bstat_rows <= fin_rows_max) { a.bn->bfin_pending = true; return 0; }
    return finish_grad(a);
}

int Net::finish_grad2_or_defer(const Act& a, bool defer_a, const Act& b, bool defer_b) {
    const bool da = a.bn && defer_a && a.bn->bstat_rows > 0 && a.bn->bstat_rows <= fin_rows_max;
    const bool db = b.bn && defer_b && b.bn->bstat_rows > 0 && b.bn->bstat_rows <= fin_rows_max;
    if (da && db) { a.bn->bfin_pending = true; b.bn->bfin_pending = true; return 0; }
    if (da) { a.bn->bfin_pending = true; return finish_grad(b); }
    if (db) { b.bn->bfin_pending = true; return finish_grad(a); }
    return finish_grad2(a, b);
}

// the first reader of x3's BatchNorm-backward constants is conv3's data gradient (Residual::bwd_a), launched on whatever stream runs the
// block's backward pass; its weight gradient and the shortcut addend come later
bool Net::x3_fin_ok(const Residual& r, const Act& in) const { return (fin_mask & 4) && !drop_mask && fin_consumer_ok(r.c3, in.B, in.H, in.W, true); }

int Net::finish_grad2(const Act& a, const Act& b) {
    if (ablate() & 4) return 0;
    if ((ablate() & 256) && a.H <= ablate_h()) return 0;
    if (!a.bn || !b.bn) { TRY(finish_grad(a)); return finish_grad(b); }
    BNLayer *x = a.bn, *y = b.bn;
    return pa_launch_bn_bwd_finalize2(x->bstats, x->bstat_rows, x->scale, x->mean, x->invstd, x->kA, x->kB, x->kC, grads + x->p_gamma, grads + x->p_beta,
                                      x->C, (float)a.M(),
                                      y->bstats, y->bstat_rows, y->scale, y->mean, y->invstd, y->kA, y->kB, y->kC, grads + y->p_gamma, grads + y->p_beta,
                                      y->C, (float)b.M(), st);
}

bool Net::fin_consumer_ok(const ConvLayer& c, int B_, int H, int W, bool dgrad) const {
    if (fin_rows_max <= 0 || !train_bn || c.k == 7) return false;
    if ((long)B_ * H * W > 64L * PA_FIN_SMALL_ROWS) return false;          // (no producer tiling leaves <= 128 rows above that)
    PaConvArgs a; memset(&a, 0, sizeof a);
    a.B = B_; a.H = H; a.W = W; a.taps = c.taps();
    a.Cin = dgrad ? c.pcout : c.pcin; a.Cout = dgrad ? c.pcin : c.pcout;
    a.in.mode = dgrad ? PA_LD_LIN2 : PA_LD_BNRELU;
    return pa_conv_takes_fin(a);
}

PaBnFin Net::fin_fwd(const BNLayer& b, int M) const {
    PaBnFin f; memset(&f, 0, sizeof f);
    f.stats = b.stats; f.rows = b.stat_rows; f.bwd = 0; f.count = (float)M;
    f.gamma = params + b.p_gamma; f.beta = params + b.p_beta; f.rmean = buffers + b.b_rmean; f.rvar = buffers + b.b_rvar;
    f.scale = b.scale; f.shift = b.shift; f.mean = b.mean; f.invstd = b.invstd;
    f.momentum = momentum; f.eps = eps; f.update_running = bn_update;
    return f;
}

PaBnFin Net::fin_bwd(const BNLayer& b, int M) const {
    PaBnFin f; memset(&f, 0, sizeof f);
    f.stats = b.bstats; f.rows = b.bstat_rows; f.bwd = 1; f.count = (float)M;
    f.scale = b.scale; f.mean = b.mean; f.invstd = b.invstd;
    f.kA = b.kA; f.kB = b.kB; f.kC = b.kC; f.dgamma = grads + b.p_gamma; f.dbeta = grads + b.p_beta;
    return f;
}

int Net::conv_fwd(ConvLayer& c, const PaOperand& in, int B_, int H, int W, const PaOperand& add1, const PaOperand& add2,
                  bf16* out, BNLayer* bn_after, BNLayer* pending_in, bool defer_after) {
    PaConvArgs a; memset(&a, 0, sizeof a);
    if (pending_in && pending_in->fin_pending) { a.fin = fin_fwd(*pending_in, B_ * H * W); pending_in->fin_pending = false; }
    a.in = in; a.w = c.wf; a.bias = params + c.p_b; a.add1 = add1; a.add2 = add2; a.out = out; a.low_prio = on_side ? 1 : 0;
    a.B = B_; a.H = H; a.W = W; a.Cin = c.pcin; a.Cout = c.pcout; a.taps = c.k == 7 ? 1 : c.taps();
    a.ep = ep_plain();
    if (bn_after && train_bn) { a.ep.mode = PA_OUT_STATS; a.ep.stats = bn_after->stats; a.ep.rows_out = &bn_after->stat_rows; }
    double wb, wf; conv_work(B_ * H * W, c.k == 7 ? 147 : c.Cin, c.Cout, c.k == 7 ? 1 : c.taps(), false, wb, wf);
    if (c.k == 7) wb = 2.0 * B_ * (4.0 * H * W * 4 + (double)H * W * 64) + 2.0 * 64 * 147;      // image read once (4-ch padded) + output
    if ((ablate() & 128) && H <= ablate_h()) return 0;
    { const double M = (double)B_ * H * W;
      cnt((c.k == 7 ? 2.0 * B_ * 4.0 * H * W * 4 : opb(in, M * a.Cin)) + opb(add1, M * a.Cout) + opb(add2, M * a.Cout) + 2.0 * a.Cout * a.taps * a.Cin, 2.0 * M * a.Cout); }
    ProfEntry* pe = prof.begin(c.k == 7 ? PA_PROF_STEM_FWD : ((long)B_ * H * W < PA_PROF_LOW_M ? PA_PROF_LOW_FWD : (c.k == 3 ? PA_PROF_FWD3 : PA_PROF_FWD1)), wb, wf, st);
    int rc = (c.k == 3 && (ablate() & 8)) ? 0 : ((c.k == 7) ? pa_launch_stem_conv(a, st) : pa_launch_conv(a, st));
    prof.end(pe, st);
    TRY(rc);
    if (bn_after && train_bn && defer_after && bn_after->stat_rows > 0 && bn_after->stat_rows <= fin_rows_max) { bn_after->fin_pending = true; return 0; }
    if (bn_after && train_bn && !(ablate() & 4))
        TRY(pa_launch_bn_finalize(bn_after->stats, bn_after->stat_rows, params + bn_after->p_gamma, params + bn_after->p_beta,
                                  buffers + bn_after->b_rmean, buffers + bn_after->b_rvar, bn_after->scale, bn_after->shift,
                                  bn_after->mean, bn_after->invstd, bn_after->C, (float)(B_ * H * W), momentum, eps, bn_update, st));
    return 0;
}

int Net::conv_dgrad(ConvLayer& c, const PaOperand& dy, int B_, int H, int W, const PaOperand& add1, const PaOperand& add2,
                    const PaEpilogue& ep, bf16* out, bf16* dz_out, bool* dz_done, BNLayer* pending_in, const Act* low, bool* low_done) {
    PaConvArgs a; memset(&a, 0, sizeof a);
    if (low_done) *low_done = false;
    if (pending_in && pending_in->bfin_pending) { a.fin = fin_bwd(*pending_in, B_ * H * W); pending_in->bfin_pending = false; }
    a.in = dy; a.w = c.wb; a.bias = nullptr; a.add1 = add1; a.add2 = add2; a.out = out; a.ep = ep; a.low_prio = on_side ? 1 : 0;
    a.B = B_; a.H = H; a.W = W; a.Cin = c.pcout; a.Cout = c.pcin; a.taps = c.taps();
    if (dz_done) *dz_done = false;
    if (dz_out && dy.mode == PA_LD_LIN2) {         // only the 1x1 row-tile kernel stores its transformed input
        static int off = -1;
        if (off < 0) off = (pa_getenv("PA_CONV1_OLD") || pa_getenv("PA_NO_DZ3")) ? 1 : 0;
        static int off3 = -1;
        if (off3 < 0) off3 = (pa_getenv("PA_CONV3_OLD") || pa_getenv("PA_NO_DZ2")) ? 1 : 0;
        if ((!off && a.taps == 1 && pa_conv1x1_tile_supported(a)) || (!off3 && a.taps == 9 && pa_conv3x3_tile_supported(a))) { a.dz_out = dz_out; if (dz_done) *dz_done = true; }
    }
    double wb, wf; conv_work(B_ * H * W, c.Cin, c.Cout, c.taps(), false, wb, wf);
    ProfEntry* pe = prof.begin((long)B_ * H * W < PA_PROF_LOW_M ? PA_PROF_LOW_DGRAD : (c.k == 3 ? PA_PROF_DGRAD3 : PA_PROF_DGRAD1), wb, wf, st);
    if ((c.k == 3 && (ablate() & 8)) || (c.k == 1 && (ablate() & 16)) || ((ablate() & 256) && H <= ablate_h())) { if (a.ep.rows_out) *a.ep.rows_out = 1; prof.end(pe, st); return 0; }
    // the low-resolution half of the upsample-add backward in this launch's epilogue (round 6)
    static int up_fuse = -1;
    if (up_fuse < 0) { const char* e = pa_getenv("PA_UPADD_FUSE"); up_fuse = e ? atoi(e) : PA_UPADD_FUSE_DEFAULT; }
    bool up = false;
    if (low && up_fuse && low->bn && !drop_mask && !(ablate() & 1024)) {
        a.out2 = low->grad; a.ep2 = final_ep(*low);
        if (pa_conv1x1_tile_up_supported(a)) up = true; else { a.out2 = nullptr; memset(&a.ep2, 0, sizeof a.ep2); }
    }
    { const double M = (double)B_ * H * W;
      cnt(opb(dy, M * a.Cin) + opb(add1, M * a.Cout) + opb(add2, M * a.Cout) + (ep.mode == PA_OUT_BWD ? 2.0 * M * a.Cout : 0.0) + 2.0 * a.Cout * a.taps * a.Cin
              + (up ? 2.0 * M * a.Cout / 4 : 0.0),
          2.0 * M * a.Cout + (a.dz_out ? 2.0 * M * a.Cin : 0.0) + (up ? 2.0 * M * a.Cout / 4 : 0.0)); }
    int rc = up ? pa_launch_conv1x1_tile(a, st) : pa_launch_conv(a, st);
    if (up && low_done) *low_done = true;
    prof.end(pe, st);
    return rc;
}

int Net::conv_wgrad(ConvLayer& c, const PaOperand& dy, const PaOperand& x, int B_, int H, int W) {
    PaWgradArgs a; memset(&a, 0, sizeof a);
    a.dy = dy; a.x = x; a.part = c.part; a.dbpart = c.dbpart;
    a.B = B_; a.H = H; a.W = W; a.Cin = c.pcin; a.Cout = c.pcout; a.taps = c.k == 7 ? 1 : c.taps(); a.splits = c.splits;
    double wb, wf; conv_work(B_ * H * W, c.k == 7 ? 147 : c.Cin, c.Cout, c.k == 7 ? 1 : c.taps(), true, wb, wf);
    if (c.k == 7) wb = 2.0 * B_ * (4.0 * H * W * 4 + (double)H * W * 64) + 4.0 * 64 * 147;
    const int cls = c.k == 7 ? PA_PROF_STEM_WGRAD : ((long)B_ * H * W < PA_PROF_LOW_M ? PA_PROF_LOW_WGRAD : (c.k == 3 ? PA_PROF_WGRAD3 : PA_PROF_WGRAD1));
    if (ablate() & 2) return 0;
    if ((ablate() & 32) && c.k == 1 && (long)B_ * H * W >= 16384) return 0;      // timing bound: what fusing the large 1x1 weight gradients into their data gradients could save at most
    if ((ablate() & 64) && c.k == 3 && (long)B_ * H * W >= 16384) return 0;
    if ((ablate() & 512) && H <= ablate_h()) return 0;
    { const double M = (double)B_ * H * W, slab = 4.0 * (double)c.splits * a.Cout * a.taps * a.Cin;      // slabs written, read back by the reducer, gradient written
      cnt((c.k == 7 ? 2.0 * B_ * 4.0 * H * W * 4 : opb(x, M * a.Cin)) + opb(dy, M * a.Cout) + slab, slab + 4.0 * a.Cout * a.taps * a.Cin); }
    static int stem_main = -1;
    if (stem_main < 0) { const char* e = pa_getenv("PA_STEM_ON_MAIN"); stem_main = e ? atoi(e) : PA_STEM_ON_MAIN_DEFAULT; }
    // (the stem's weight gradient is the last launch of the step: on the main stream, idle by then, it runs beside the weight-gradient
    // queue's last reduction instead of behind it)
    // single-stream mode (bench.py's roofline leg, debugging): the launches the grouped kernel takes are collected too and leave as the
    // same group launches on the caller's stream -- per-class event times and the bitwise comparison of the stream modes then see the
    // kernels of the real step
    const bool defer_single = !(multi_stream && wstream) && !immediate_reduce && c.k != 7 && pa_wgrad_group_takes(a);
    if (defer_single || (multi_stream && wstream && !(c.k == 7 && stem_main && reduce_early && !immediate_reduce))) {        // deferred: flush_wgrads() launches it (on the weight-gradient stream)
        PendingWgrad p; p.a = a; p.cls = cls; p.bytes = wb; p.flops = wf; p.stem = c.k == 7; p.c = &c;
        pending_wgrads.push_back(p);
        return 0;
    }
    ProfEntry* pe = prof.begin(cls, wb, wf, st);
    int rc = (c.k == 7) ? pa_launch_stem_wgrad(a, st) : pa_launch_wgrad(a, st);
    prof.end(pe, st);
    if (rc) return rc;
    if (c.k == 7 && multi_stream && wstream && stem_main && reduce_early && !immediate_reduce) {
        TRY(pa_launch_stem_wgrad_reduce(c.part, c.splits, grads + c.p_w, st, grads + c.p_b));
        return 0;
    }
    if (immediate_reduce) {
        if (c.k == 7) {
            TRY(pa_launch_stem_wgrad_reduce(c.part, c.splits, grads + c.p_w, st, grads + c.p_b));
        } else {
            TRY(pa_launch_wgrad_reduce(red_jobs + c.red_index, 1, c.Cout * c.Cin * c.taps() + c.Cout, st));
        }
    }
    return 0;
}

// Operands of the collected launches are final on `st` at this point (their producers were enqueued before).  The
// gradient buffers they read are written once per step, so deferring a launch is always safe.
int Net::release_held(int k) {
    if (!hold) return 0;
    hold = false;
    hipStream_t ws = wstreams[w_rr];
    for (int kk = 0; kk < k && kk < 4; ++kk)          // the held groups of the skip branches: their operands are final at ev_join[kk]
        if (forks(kk)) PA_CHECK(hipStreamWaitEvent(ws, ev_join[kk], 0));
    return flush_wgrads();
}

int Net::flush_red_stash(hipStream_t ws) {
    red_stash_flushes = 0;
    if (red_stash.empty()) return 0;
    const int rc = pa_launch_wgrad_reduce_list(red_jobs, red_stash.data(), (int)red_stash.size(), red_stash_mx, ws);
    red_stash.clear(); red_stash_mx = 0;
    return rc;
}

int Net::flush_wgrads() {
    if (pending_wgrads.empty() || hold) return 0;
    const bool ms = multi_stream && wstream;
    hipStream_t ws = ms ? wstreams[w_rr] : st;
    if (ms) w_rr = (w_rr + 1) % n_w;
    if (ms && !(nosync() && (g_nosync & 4))) {
        hipEvent_t ev = ev_w[ev_w_next]; ev_w_next = (ev_w_next + 1) & 15;
        PA_CHECK(hipEventRecord(ev, st));
        PA_CHECK(hipStreamWaitEvent(ws, ev, 0));
    }
    // the launches of a group share no data: from the second on they need no completion / cache round trip behind their predecessor
    // (hipExtAnyOrderLaunch, conv_wgrad_tile.hip; not with the per-launch event timing, not with one slab shared by all layers, and not
    // inside a stream capture -- the engine's own or a caller's around pa_hg_backward: an extension launch cannot be captured)
    static int any_order = -1;
    if (any_order < 0) { const char* e = pa_getenv("PA_WGRAD_ANYORDER"); any_order = e ? atoi(e) : 1; }
    bool in_capture = capturing;
    if (!in_capture) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(ws, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) in_capture = true;
    }
    bool first = true;
    // every launch of the list the grouped kernel takes goes into group launches of up to 8 jobs (also while the per-launch event timing
    // runs: a group launch counts for the class of its largest job); the others -- the stem, shapes of the generic kernel -- are launched
    // one by one behind them
    std::vector<char> grouped(pending_wgrads.size(), 0);
    if (!immediate_reduce) {
        const PaWgradArgs* jobs[8]; int jidx[8]; int nj = 0;
        auto fire = [&]() -> int {
            if (nj == 0) return 0;
            // (event timing: the launch counts for the class of its largest job, with the bytes / flops of all of them)
            int big = 0; double bytes = 0.0, flops = 0.0;
            for (int j = 0; j < nj; ++j) {
                const PendingWgrad& q = pending_wgrads[jidx[j]];
                bytes += q.bytes; flops += q.flops;
                if (q.flops > pending_wgrads[jidx[big]].flops) big = j;
            }
            ProfEntry* pe = prof.begin(pending_wgrads[jidx[big]].cls, bytes, flops, ws);
            // the slabs of EARLIER flushes (the stash) are summed by a few more workgroups of this launch instead of by a launch of their own
            static int red_in_group = -1;
            if (red_in_group < 0) { const char* e = pa_getenv("PA_WG_RED_IN_GROUP"); red_in_group = e ? atoi(e) : PA_WG_RED_IN_GROUP_DEFAULT; }
            const bool carry = red_in_group && ms && n_w == 1 && reduce_early && !(ablate() & 1) && !red_stash.empty();      // (one weight-gradient stream: the stash's slabs were written on THIS stream)
            const int rc = (nj == 1 && !carry) ? pa_launch_wgrad(*jobs[0], ws)
                                               : pa_launch_wgrad_group(jobs, nj, ws, false, red_jobs, carry ? red_stash.data() : nullptr, carry ? (int)red_stash.size() : 0);
            if (carry && !rc) { red_stash.clear(); red_stash_mx = 0; red_stash_flushes = 0; }
            prof.end(pe, ws);
            nj = 0; first = false;
            return rc;
        };
        // a group launch is capped at `cap` workgroups: its workgroups live for tens of microseconds with 242 registers each, two of them
        // close a CU to every other queue -- a launch of <= 256 leaves room for a workgroup of the main chain beside it on every CU
        static int cap = -1;
        if (cap < 0) { const char* e = pa_getenv("PA_WG_GROUP_CAP"); cap = e ? atoi(e) : PA_WG_GROUP_CAP_DEFAULT; }
        int wgs = 0;
        for (size_t pi = 0; pi < pending_wgrads.size(); ++pi) {
            PendingWgrad& p = pending_wgrads[pi];
            if (p.stem || !pa_wgrad_group_takes(p.a)) continue;
            const int w = pa_wgrad_job_workgroups(p.a);
            if (nj > 0 && (nj == 8 || wgs + w > cap)) { const int rc = fire(); if (rc) { pending_wgrads.clear(); return rc; } wgs = 0; }
            grouped[pi] = 1;
            jidx[nj] = (int)pi;
            jobs[nj++] = &p.a;
            wgs += w;
        }
        { const int rc = fire(); if (rc) { pending_wgrads.clear(); return rc; } }
    }
    for (size_t pi = 0; pi < pending_wgrads.size(); ++pi) {
        PendingWgrad& p = pending_wgrads[pi];
        if (grouped[pi]) continue;
        ProfEntry* pe = prof.begin(p.cls, p.bytes, p.flops, ws);
        pa_wgrad_set_launch_flags((any_order && !first && !prof.on && !immediate_reduce && !in_capture) ? 1u : 0u);
        int rc = p.stem ? pa_launch_stem_wgrad(p.a, ws) : pa_launch_wgrad(p.a, ws);
        pa_wgrad_set_launch_flags(0u);
        first = false;
        prof.end(pe, ws);
        if (rc) { pending_wgrads.clear(); return rc; }
        if (immediate_reduce) {            // ONE slab shared by all layers: reduce before the next launch overwrites it
            if (p.stem) {
                TRY(pa_launch_stem_wgrad_reduce(p.c->part, p.c->splits, grads + p.c->p_w, ws, grads + p.c->p_b));
            } else {
                TRY(pa_launch_wgrad_reduce(red_jobs + p.c->red_index, 1, p.c->Cout * p.c->Cin * p.c->taps() + p.c->Cout, ws));
            }
        }
    }
    if (ms && reduce_early && !immediate_reduce && !(ablate() & 1)) {
        // the slabs of this flush join the stash; the stash is summed every `lag`-th flush (one launch for the layers of two or three
        // blocks: 41 reduction launches per step were 0.3 - 0.6 ms of the weight-gradient queue), part of them still in the Infinity Cache
        static int lag = -1;
        if (lag < 0) { const char* e = pa_getenv("PA_WREDUCE_LAG"); lag = e ? atoi(e) : PA_WREDUCE_LAG_DEFAULT; if (lag < 1) lag = 1; }
        for (PendingWgrad& p : pending_wgrads) {
            if (p.stem) { TRY(pa_launch_stem_wgrad_reduce(p.c->part, p.c->splits, grads + p.c->p_w, ws, grads + p.c->p_b)); continue; }
            const int el = p.c->Cout * p.c->Cin * p.c->taps() + p.c->Cout;
            if ((int)red_stash.size() == PA_RED_LIST_MAX) TRY(flush_red_stash(ws));
            red_stash.push_back(p.c->red_index);
            red_stash_mx = el > red_stash_mx ? el : red_stash_mx;
        }
        // (in-group reductions: the stash waits for the next group launch of this stream; a launch of its own only when none came for `lag` + 2 flushes)
        static int rig = -1;
        if (rig < 0) { const char* e = pa_getenv("PA_WG_RED_IN_GROUP"); rig = e ? atoi(e) : PA_WG_RED_IN_GROUP_DEFAULT; }
        if (++red_stash_flushes >= (rig && n_w == 1 ? lag + 2 : lag) || n_w > 1) TRY(flush_red_stash(ws));          // (several weight-gradient streams, tuning builds: no lag across streams)
    }
    pending_wgrads.clear();
    return 0;
}

// the streaming launches with their operand bytes counted (Net::cnt, pa_net_design_bytes)
static int c_maxpool_fwd(Net& n, const PaOperand& in, bf16* out, int B, int H, int W, int C, hipStream_t st) {
    const double e = (double)B * H * W * C; n.cnt(Net::opb(in, e), 2.0 * e / 4);
    if (ablate() & 4096) return 0;
    return pa_launch_maxpool_fwd(in, out, B, H, W, C, st);
}
static int c_maxpool_bwd(Net& n, const bf16* dout, const PaOperand& in, const PaOperand& add, const PaEpilogue& ep, bf16* din, int B, int H, int W, int C,
                         hipStream_t st, int* rows = nullptr) {
    const double e = (double)B * H * W * C; n.cnt(2.0 * e / 4 + Net::opb(in, e) + Net::opb(add, e) + (ep.mode == PA_OUT_BWD ? 2.0 * e : 0.0), 2.0 * e);
    if (ablate() & 2048) { if (ep.rows_out) *ep.rows_out = 1; return 0; }
    return pa_launch_maxpool_bwd(dout, in, add, ep, din, B, H, W, C, st, rows);
}
static int c_upadd_fwd(Net& n, const PaOperand& low, const PaOperand& skip, bf16* out, int B, int H, int W, int C, hipStream_t st) {
    const double e = (double)B * H * W * C; n.cnt(Net::opb(low, e / 4) + Net::opb(skip, e), 2.0 * e);
    if (ablate() & 4096) return 0;
    return pa_launch_upadd_fwd(low, skip, out, B, H, W, C, st);
}
static int c_upadd_bwd(Net& n, const bf16* dout, const PaEpilogue& ep_low, bf16* dlow, const PaEpilogue& ep_skip, bf16* dskip, int B, int H, int W, int C,
                       hipStream_t st, int* rows = nullptr, int part = 3) {
    const double e = (double)B * H * W * C;
    n.cnt(2.0 * e + ((part & 1) && ep_low.mode == PA_OUT_BWD ? 2.0 * e / 4 : 0.0) + ((part & 2) && ep_skip.mode == PA_OUT_BWD ? 2.0 * e : 0.0),
          ((part & 1) ? 2.0 * e / 4 : 0.0) + ((part & 2) ? 2.0 * e : 0.0));
    if (ablate() & 1024) { if (ep_low.rows_out && (part & 1)) *ep_low.rows_out = 1; if (ep_skip.rows_out && (part & 2)) *ep_skip.rows_out = 1; return 0; }
    return pa_launch_upadd_bwd(dout, ep_low, dlow, ep_skip, dskip, B, H, W, C, st, rows, part);
}
template <class... A> static int c_head_fwd(Net& n, const PaOperand& x, A... rest) {
    const double M = (double)n.B * (n.res / 4) * (n.res / 4); n.cnt(Net::opb(x, M * n.chan) + 2.0 * 16 * n.chan, M * 16 * 4 + M * 64 * 2);
    return pa_launch_head_fwd(x, rest...);
}
template <class... A> static int c_heat_grad(Net& n, const float* heat, const double* pts, const bf16* din, A... rest) {
    const double M = (double)n.B * (n.res / 4) * (n.res / 4); n.cnt(M * 16 * 4 + (din ? M * 64 * 2 : 0.0), M * 64 * 2);
    return pa_launch_heat_grad(heat, pts, din, rest...);
}

// ------------------------------------------------------------------------------------------------
// a scope whose launches go to stream `s` (a side branch)
struct StreamScope {
    Net& n; hipStream_t saved; bool saved_side;
    StreamScope(Net& n_, hipStream_t s) : n(n_), saved(n_.st), saved_side(n_.on_side) { n.st = s; n.on_side = true; }
    ~StreamScope() { n.st = saved; n.on_side = saved_side; }
};

// residual block (reference :30-49): x1 = conv1(a), x2 = conv3x3(relu bn1 x1), x3 = conv3(relu bn2 x2) + shortcut
int Residual::fwd(Net& n, const Act& in) {
    const int B = in.B, H = in.H, W = in.W;
    // x1 / x2 have ONE reader each (conv2 / conv3): at the low-resolution levels their BatchNorm finalize runs in that reader's prologue
    const bool d1 = (n.fin_mask & 1) && n.fin_consumer_ok(c2, B, H, W, false), d2 = (n.fin_mask & 1) && n.fin_consumer_ok(c3, B, H, W, false);
    // the adapter (residual1, residual3: in front of the hourglasses, the side stream is idle) depends on the block input only: it runs
    // on the side stream beside conv1 / conv2 instead of between conv2 and conv3 on the main chain
    const bool ad_par = has_adapter && n.adapter_parallel() ;
    if (ad_par) {
        TRY(n.fork_to(0));
        { StreamScope sc(n, n.side[0]); TRY(n.conv_fwd(ad, n.op(in), B, H, W, pa_none(), pa_none(), adout, nullptr)); }
        TRY(n.record_join(0));
    }
    TRY(n.conv_fwd(c1, n.op(in), B, H, W, pa_none(), pa_none(), x1.raw, &b1, nullptr, d1));
    TRY(n.conv_fwd(c2, n.op(x1), B, H, W, pa_none(), pa_none(), x2.raw, &b2, &b1, d2));
    if (has_adapter) {
        if (ad_par) TRY(n.wait_join(0));
        else TRY(n.conv_fwd(ad, n.op(in), B, H, W, pa_none(), pa_none(), adout, nullptr));
        TRY(n.conv_fwd(c3, n.op(x2), B, H, W, pa_plain(adout), pa_none(), x3.raw, &b3, &b2));
    } else {
        TRY(n.conv_fwd(c3, n.op(x2), B, H, W, n.op(in), pa_none(), x3.raw, &b3, &b2));
    }
    return 0;
}

// precondition: x3.grad holds the finished masked gradient and finish_grad(x3) has run.
// bwd = bwd_a (parameter gradients and the inner data gradients) + bwd_b (gradient of the block input, the only
// part that needs `extra`, the gradient arriving at the input from its other consumers)
int Residual::bwd_a(Net& n, const Act& in, const PaOperand* extra) {
    const int B = in.B, H = in.H, W = in.W;
    low_fused = false;
    // conv3's data gradient goes first: its kernel also stores dz3 = BatchNorm-backward(x3.grad, x3.raw), and the weight
    // gradients / the shortcut addend after it read that one tensor instead of recomputing it from two
    // (x3.bn: its finalize may be pending -- then this launch does it in its prologue, Net::x3_fin_ok)
    PaOperand g3 = n.gradop(x3);
    dz3_valid = false;
    TRY(n.conv_dgrad(c3, g3, B, H, W, pa_none(), pa_none(), n.final_ep(x2), x2.grad, dz3, &dz3_valid, x3.bn));
    if (dz3_valid) g3 = pa_plain(dz3);
    // the adapter's data gradient needs g3 and the gradient reaching the block input from elsewhere (`extra`) only: beside conv2's data
    // gradient on the side stream when both are known here (residual1 / residual3: no other consumer of the input)
    ad_forked = false;
    if (has_adapter && extra && n.adapter_parallel()) {
        TRY(n.fork_to(0));
        { StreamScope sc(n, n.side[0]); TRY(n.conv_dgrad(ad, g3, B, H, W, *extra, pa_none(), ep_plain(), adgrad)); }
        TRY(n.record_join(0));
        ad_forked = true;
    }
    TRY(n.conv_wgrad(c3, g3, n.op(x2), B, H, W));
    // (x2's BatchNorm-backward constants are first read by conv2's data gradient, which can compute them in its prologue; conv2's
    // weight gradient, on the weight-gradient stream, is launched behind it)
    TRY(n.finish_grad_or_defer(x2, (n.fin_mask & 2) && n.fin_consumer_ok(c2, B, H, W, true)));
    PaOperand g2 = n.gradop(x2);
    bool dz2_valid = false;
    TRY(n.conv_dgrad(c2, g2, B, H, W, pa_none(), pa_none(), n.final_ep(x1), x1.grad, dz2, &dz2_valid, x2.bn));
    if (dz2_valid) g2 = pa_plain(dz2);
    TRY(n.conv_wgrad(c2, g2, n.op(x1), B, H, W));
    // (x1's constants have two first readers -- conv1's weight gradient on the weight-gradient stream and conv1's data gradient in bwd_b:
    // queueing the weight gradient behind a data gradient that finalizes in its prologue measured +0.05 ms, round 4 -- a launch stays)
    TRY(n.finish_grad(x1));
    const PaOperand g1 = n.gradop(x1);
    TRY(n.conv_wgrad(c1, g1, n.op(in), B, H, W));
    if (has_adapter) TRY(n.conv_wgrad(ad, g3, n.op(in), B, H, W));
    if (!n.on_side && (++n.flush_ctr % n.flush_every) != 0) return 0;
    return n.flush_wgrads();                   // one event for the block's 3-4 weight gradients
}

int Residual::bwd_b(Net& n, const Act& in, const PaOperand& extra) {
    const int B = in.B, H = in.H, W = in.W;
    const PaOperand g3 = dz3_valid ? pa_plain(dz3) : n.gradop(x3), g1 = n.gradop(x1);
    if (has_adapter) {
        if (ad_forked) { TRY(n.wait_join(0)); ad_forked = false; }
        else TRY(n.conv_dgrad(ad, g3, B, H, W, extra, pa_none(), ep_plain(), adgrad));
        TRY(n.conv_dgrad(c1, g1, B, H, W, pa_plain(adgrad), pa_none(), n.final_ep(in), in.grad));
    } else {
        TRY(n.conv_dgrad(c1, g1, B, H, W, g3, extra, n.final_ep(in), in.grad, nullptr, nullptr, nullptr, low_of_in, &low_fused));
    }
    return 0;
}

int Residual::bwd(Net& n, const Act& in, const PaOperand& extra, bool in_needs_grad) {
    TRY(bwd_a(n, in, in_needs_grad ? &extra : nullptr));
    return in_needs_grad ? bwd_b(n, in, extra) : 0;
}

// ------------------------------------------------------------------------------------------------
// hourglass (reference :139-157 down path, :192-203 up path)
// The skip branch of level k (a full-resolution residual block) does not depend on the low-resolution path
// below it: it is enqueued on side stream k and joined where its result is consumed.

int Hourglass::encode(Net& n, const Act& in) {
    const Act* cur = &in;
    for (int k = 0; k < 4; ++k) {
        auto skip_branch = [&]() -> int {
            TRY(skip[k].fwd(n, *cur));
            if (n.drop_mask) TRY(pa_launch_cell_mask(n.op(skip[k].x3), n.drop_mask, ep_plain(), skipm[k].raw, cur->B, cur->H, cur->W, cur->C, n.st));
            return 0;
        };
        if (n.forks(k)) {
            TRY(n.fork_to(k));
            { StreamScope sc(n, n.side[k]); TRY(skip_branch()); }
            TRY(n.record_join(k));
        } else {
            TRY(skip_branch());
        }
        TRY(c_maxpool_fwd(n, n.op(*cur), pooled[k].raw, cur->B, cur->H, cur->W, cur->C, n.st));
        TRY(down[k].fwd(n, pooled[k]));
        cur = &down[k].x3;
    }
    TRY(neck.fwd(n, *cur));
    if (n.drop_mask) {
        const Act& x = neck.x3;
        TRY(pa_launch_cell_mask(n.op(x), n.drop_mask, ep_plain(), neckm.raw, x.B, x.H, x.W, x.C, n.st));
    }
    return 0;
}

int Hourglass::decode(Net& n) {
    const Act* low = n.drop_mask ? &neckm : &neck.x3;
    for (int k = 3; k >= 0; --k) {
        TRY(up[k].fwd(n, *low));
        const Act& m = merged[k];
        if (n.forks(k)) TRY(n.wait_join(k));
        TRY(c_upadd_fwd(n, n.op(up[k].x3), n.op(n.drop_mask ? skipm[k] : skip[k].x3), m.raw, m.B, m.H, m.W, m.C, n.st));
        low = &merged[k];
    }
    return 0;
}

bool Hourglass::low_fusable(const Net& n, int k) const {
    if (n.drop_mask || k < 0 || k > 3) return false;
    const Act& m = merged[k];
    return pa_upadd_bwd_splits(n.final_ep(up[k].x3), n.final_ep(skip[k].x3), m.B, m.H, m.W, m.C);
}

// precondition: merged[0].grad holds the finished (plain) gradient of the hourglass output.
// extra0: gradient reaching the hourglass INPUT from consumers outside the hourglass.
int Hourglass::bwd(Net& n, const Act& in, const PaOperand& extra0) {
    if (n.hold_level > 0 && n.multi_stream && n.wstream) n.hold = true;
    for (int k = 0; k < 4; ++k) {
        if (k == n.hold_level) TRY(n.release_held(k));
        const Act& m = merged[k];
        // the two outputs of the upsample-add backward as two launches: d low on the main chain (the low-resolution path waits for it), d skip
        // on the side stream in front of the skip block's backward pass, its only reader -- the main chain waits for 75 MB instead of 175
        static int split_env = -1;
        if (split_env < 0) { const char* e = pa_getenv("PA_UPADD_SPLIT"); split_env = e ? atoi(e) : PA_UPADD_SPLIT_DEFAULT; }
        if (split_env && !n.drop_mask && n.forks(k) && pa_upadd_bwd_splits(n.final_ep(up[k].x3), n.final_ep(skip[k].x3), m.B, m.H, m.W, m.C)) {
            const Act& upin_ = (k == 3) ? neck.x3 : merged[k + 1];
            const Act& x = (k == 0) ? in : down[k - 1].x3;
            TRY(n.fork_to(k));
            {
                StreamScope sc(n, n.side[k]);
                TRY(c_upadd_bwd(n, m.grad, n.final_ep(up[k].x3), up[k].x3.grad, n.final_ep(skip[k].x3), skip[k].x3.grad, m.B, m.H, m.W, m.C, n.st, nullptr, 2));
                TRY(n.finish_grad_or_defer(skip[k].x3, n.x3_fin_ok(skip[k], x)));
                TRY(skip[k].bwd_a(n, x));
            }
            TRY(n.record_join(k));
            // (d up[k].x3 may have come out of the data gradient that produced d merged[k]: low_done)
            if (!low_done[k]) TRY(c_upadd_bwd(n, m.grad, n.final_ep(up[k].x3), up[k].x3.grad, n.final_ep(skip[k].x3), skip[k].x3.grad, m.B, m.H, m.W, m.C, n.st, nullptr, 1));
            TRY(n.finish_grad_or_defer(up[k].x3, n.x3_fin_ok(up[k], upin_)));
            up[k].low_of_in = low_fusable(n, k + 1) ? &up[k + 1].x3 : nullptr;
            TRY(up[k].bwd(n, upin_, pa_none(), true));
            if (k < 3) low_done[k + 1] = up[k].low_fused;
            continue;
        }
        if (n.drop_mask) {              // the skip tensor entered the sum through the cell mask: d skip = mask * d (masked skip)
            TRY(c_upadd_bwd(n, m.grad, n.final_ep(up[k].x3), up[k].x3.grad, ep_plain(), skipm[k].grad, m.B, m.H, m.W, m.C, n.st));
            TRY(pa_launch_cell_mask(pa_plain(skipm[k].grad), n.drop_mask, n.final_ep(skip[k].x3), skip[k].x3.grad, m.B, m.H, m.W, m.C, n.st));
        } else {
            // (low_done: only d skip is left -- the launch that can do one half exists wherever the fused data gradient does)
            TRY(c_upadd_bwd(n, m.grad, n.final_ep(up[k].x3), up[k].x3.grad, n.final_ep(skip[k].x3), skip[k].x3.grad,
                                    m.B, m.H, m.W, m.C, n.st, nullptr, low_done[k] ? 2 : 3));
        }
        {
            const Act& upin_ = (k == 3) ? (n.drop_mask ? neckm : neck.x3) : merged[k + 1];
            const Act& skin_ = (k == 0) ? in : down[k - 1].x3;
            TRY(n.finish_grad2_or_defer(up[k].x3, n.x3_fin_ok(up[k], upin_), skip[k].x3, n.x3_fin_ok(skip[k], skin_)));
        }
        if (n.forks(k)) {          // parameter / inner gradients of the skip block next to the deeper levels
            const Act& x = (k == 0) ? in : down[k - 1].x3;
            TRY(n.fork_to(k));
            { StreamScope sc(n, n.side[k]); TRY(skip[k].bwd_a(n, x)); }
            TRY(n.record_join(k));
        }
        const Act& upin = (k == 3) ? (n.drop_mask ? neckm : neck.x3) : merged[k + 1];
        up[k].low_of_in = low_fusable(n, k + 1) ? &up[k + 1].x3 : nullptr;
        TRY(up[k].bwd(n, upin, pa_none(), true));
        if (k < 3) low_done[k + 1] = up[k].low_fused;
    }
    if (n.drop_mask) {
        const Act& x = neck.x3;
        TRY(pa_launch_cell_mask(pa_plain(neckm.grad), n.drop_mask, n.final_ep(x), x.grad, x.B, x.H, x.W, x.C, n.st));
    }
    TRY(n.release_held(4));
    TRY(n.finish_grad_or_defer(neck.x3, n.x3_fin_ok(neck, down[3].x3)));
    TRY(neck.bwd(n, down[3].x3, pa_none(), true));
    TRY(n.finish_grad_or_defer(down[3].x3, n.x3_fin_ok(down[3], pooled[3])));
    for (int k = 3; k >= 0; --k) {
        TRY(down[k].bwd(n, pooled[k], pa_none(), true));
        const Act& x = (k == 0) ? in : down[k - 1].x3;
        TRY(c_maxpool_bwd(n, pooled[k].grad, n.op(x), (k == 0) ? extra0 : pa_none(), ep_plain(), poolgrad[k],
                                  x.B, x.H, x.W, x.C, n.st));
        if (n.forks(k)) {
            TRY(n.wait_join(k));
            TRY(skip[k].bwd_b(n, x, pa_plain(poolgrad[k])));
        } else {
            TRY(skip[k].bwd(n, x, pa_plain(poolgrad[k]), true));
        }
        if (k >= 1) TRY(n.finish_grad_or_defer(x, n.x3_fin_ok(down[k - 1], pooled[k - 1])));      // x = down[k-1].x3: read next by down[k-1]'s backward pass
        else TRY(n.finish_grad(x));
    }
    return 0;
}

// A/B (tuning builds): a stream restricted to `n` of the 256 CUs, spread evenly (PA_WSTREAM_CUS / PA_SIDE_CUS)
// ... or with a queue priority (PA_WSTREAM_PRIO / PA_SIDE_PRIO: the runtime's numeric priority, lower = more urgent)
static hipError_t create_stream_cus(hipStream_t* s, const char* env, const char* prio_env) {
    const char* e = pa_getenv(env);
    const int n = e ? atoi(e) : 0;
    if (const char* p = pa_getenv(prio_env)) return hipStreamCreateWithPriority(s, hipStreamNonBlocking, atoi(p));
    if (n <= 0 || n >= 256) return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
    uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < n; ++i) { const int cu = (int)((long)i * 256 / n); mask[cu >> 5] |= 1u << (cu & 31); }
    return hipExtStreamCreateWithCUMask(s, 8, mask);
}

int Net::ensure_streams() {
    if (streams_ready) return 0;
    if (const char* e = pa_getenv("PA_FIN_MASK")) fin_mask = atoi(e);
    if (const char* e = pa_getenv("PA_FIN_PROLOGUE")) fin_rows_max = atoi(e) > PA_FIN_SMALL_ROWS ? PA_FIN_SMALL_ROWS : atoi(e);
    if (pa_getenv("PA_SINGLE_STREAM")) { multi_stream = false; streams_ready = true; return 0; }
    if (const char* e = pa_getenv("PA_FORK_LEVELS")) fork_mask = atoi(e);
    if (const char* e = pa_getenv("PA_SIDE_STREAMS")) n_side = atoi(e) < 1 ? 1 : (atoi(e) > 4 ? 4 : atoi(e));
    reduce_early = pa_getenv("PA_WREDUCE_LATE") == nullptr;
    if (const char* e = pa_getenv("PA_WFLUSH_EVERY")) flush_every = atoi(e) > 0 ? atoi(e) : 1;
    if (const char* e = pa_getenv("PA_WHOLD")) hold_level = atoi(e);
    for (int k = 0; k < 4; ++k) {
        if (k < n_side) PA_CHECK(create_stream_cus(&side[k], "PA_SIDE_CUS", "PA_SIDE_PRIO"));
        else side[k] = side[k % n_side];            // levels share streams (in-order per stream: fork/join events keep it correct)
        PA_CHECK(hipEventCreateWithFlags(&ev_fork[k], hipEventDisableTiming));
        PA_CHECK(hipEventCreateWithFlags(&ev_join[k], hipEventDisableTiming));
    }
    if (!pa_getenv("PA_NO_WSTREAM")) {
        PA_CHECK(create_stream_cus(&wstream, "PA_WSTREAM_CUS", "PA_WSTREAM_PRIO"));
        wstreams[0] = wstream;
        if (const char* e = pa_getenv("PA_WSTREAMS")) n_w = atoi(e) < 1 ? 1 : (atoi(e) > 4 ? 4 : atoi(e));
        for (int i = 1; i < n_w; ++i) PA_CHECK(hipStreamCreateWithFlags(&wstreams[i], hipStreamNonBlocking));
        for (int i = 0; i < n_w; ++i) PA_CHECK(hipEventCreateWithFlags(&ev_wdone_x[i], hipEventDisableTiming));
        for (int i = 0; i < 16; ++i) PA_CHECK(hipEventCreateWithFlags(&ev_w[i], hipEventDisableTiming));
        PA_CHECK(hipEventCreateWithFlags(&ev_wdone, hipEventDisableTiming));
    }
    streams_ready = true;
    return 0;
}
void Net::release_streams() {
    release_graph();
    if (bucket_events) { for (int i = 0; i < 16; ++i) (void)hipEventDestroy(ev_bucket[i]); (void)hipEventDestroy(ev_bucket_main); bucket_events = false; }
    if (mstream) { (void)hipStreamSynchronize(mstream); (void)hipStreamDestroy(mstream); (void)hipEventDestroy(ev_mfork); (void)hipEventDestroy(ev_meter); mstream = nullptr; meter_pending = false; }
    for (int k = 0; k < 4; ++k) {
        if (side[k]) {
            if (k < n_side) { (void)hipStreamSynchronize(side[k]); (void)hipStreamDestroy(side[k]); }
            (void)hipEventDestroy(ev_fork[k]); (void)hipEventDestroy(ev_join[k]); side[k] = nullptr;
        }
    }
    if (wstream) {
        for (int i = 1; i < n_w; ++i) { (void)hipStreamSynchronize(wstreams[i]); (void)hipStreamDestroy(wstreams[i]); wstreams[i] = nullptr; }
        for (int i = 0; i < n_w; ++i) (void)hipEventDestroy(ev_wdone_x[i]);
        (void)hipStreamSynchronize(wstream); (void)hipStreamDestroy(wstream); wstream = nullptr; wstreams[0] = nullptr;
        for (int i = 0; i < 16; ++i) (void)hipEventDestroy(ev_w[i]);
        (void)hipEventDestroy(ev_wdone);
    }
    for (ProfEntry& e : prof.entries) { (void)hipEventDestroy(e.e0); (void)hipEventDestroy(e.e1); }
    prof.entries.clear(); prof.used = 0;
    streams_ready = false;
}
int Net::fork_to(int k) { if (nosync() && (g_nosync & 1)) return 0; PA_CHECK(hipEventRecord(ev_fork[k], st)); PA_CHECK(hipStreamWaitEvent(side[k], ev_fork[k], 0)); return 0; }
int Net::record_join(int k) { if (nosync() && (g_nosync & 2)) return 0; PA_CHECK(hipEventRecord(ev_join[k], side[k])); return 0; }
int Net::wait_join(int k) { if (nosync() && (g_nosync & 2)) return 0; PA_CHECK(hipStreamWaitEvent(st, ev_join[k], 0)); return 0; }

// ------------------------------------------------------------------------------------------------
int Net::prepare_weights() { return pa_launch_weight_prep(prep_jobs, n_prep, prep_max, st); }

int Net::begin_step() {
    dbytes_rd = dbytes_wr = 0;
    if (!loss_self_clearing) PA_CHECK(hipMemsetAsync(stats_arena, 0, stats_arena_floats * sizeof(float), st));
    return 0;
}

// reference :282-342 (stem :283-289, stacks :292-334) + loss of stack-hg.py:156-159
int Net::meter_stream(hipStream_t* out) {
    *out = st;
    // a meter on the main stream may read what a pending meter-stream launch produced (the cached arg-max): order it behind them
    if (!meters_async || !multi_stream) return join_meters();
    if (!mstream) {
        PA_CHECK(hipStreamCreateWithFlags(&mstream, hipStreamNonBlocking));
        PA_CHECK(hipEventCreateWithFlags(&ev_mfork, hipEventDisableTiming));
        PA_CHECK(hipEventCreateWithFlags(&ev_meter, hipEventDisableTiming));
    }
    PA_CHECK(hipEventRecord(ev_mfork, st));
    PA_CHECK(hipStreamWaitEvent(mstream, ev_mfork, 0));
    *out = mstream;
    return 0;
}
int Net::meter_done(hipStream_t ms) {
    if (ms == st) return 0;
    PA_CHECK(hipEventRecord(ev_meter, ms));
    meter_pending = true;
    return 0;
}
int Net::join_meters() {
    if (!meter_pending) return 0;
    PA_CHECK(hipStreamWaitEvent(st, ev_meter, 0));
    meter_pending = false;
    return 0;
}

int Net::heat_argmax(int stack, const float** out, hipStream_t on) {
    const int H = res / 4;
    if (!heat_peak_valid[stack]) {
        TRY(pa_launch_argmax(heat[stack], (long)H * H * 16, 1, 16, B, 16, H, H, heat_peak[stack], nullptr, on ? on : st));
        heat_peak_valid[stack] = 1;
    }
    *out = heat_peak[stack];
    return 0;
}

int Net::forward_pose(const float* img_nchw, const bf16* img4_in, const double* pts, bool train, float* loss_out_dev) {
    const int rc = forward_pose_body(img_nchw, img4_in, pts, train, loss_out_dev);
    // the per-stack loss accumulators are cleared by the kernel that reads them at the END of a forward pass (loss_out_kernel): a pass that
    // fails after a head launch has added to them would leave its residue in every later loss -- clear them here then (self-healing)
    if (rc && pts && loss_dev && st) { (void)hipGetLastError(); (void)hipMemsetAsync(loss_dev, 0, 64 * sizeof(float), st); }
    return rc;
}

int Net::forward_pose_body(const float* img_nchw, const bf16* img4_in, const double* pts, bool train, float* loss_out_dev) {
    train_bn = train;
    heat_peak_valid.assign(stacks, 0);
    TRY(ensure_streams());
    TRY(join_meters());                 // (meters of the previous forward pass still read the heat maps this one overwrites)
    TRY(begin_step());
    if (!train) TRY(pa_launch_bn_eval(bneval_jobs, n_bneval, eps, st));
    const bf16* image = img4_in ? img4_in : img4;
    cur_image = image;
    if (!img4_in) TRY(pa_launch_nchw_to_nhwc4(img_nchw, img4, B, res, res, st));
    if (pts && pts != pts_dev) PA_CHECK(hipMemcpyAsync(pts_dev, pts, (size_t)B * classes * 2 * sizeof(double), hipMemcpyDeviceToDevice, st));
    TRY(conv_fwd(stem_conv, pa_plain(image), B, res / 2, res / 2, pa_none(), pa_none(), a0.raw, &stem_bn));
    TRY(res1.fwd(*this, a0));
    TRY(c_maxpool_fwd(*this, op(res1.x3), pool0.raw, B, res / 2, res / 2, 128, st));
    TRY(res2.fwd(*this, pool0));
    TRY(res3.fwd(*this, res2.x3));
    const int Hh = res / 4;
    for (int i = 0; i < stacks; ++i) {
        TRY(hg[i].encode(*this, xin[i]));
        TRY(hg[i].decode(*this));
        TRY(post[i].fwd(*this, hg[i].out()));
        TRY(conv_fwd(lin[i], op(post[i].x3), B, Hh, Hh, pa_none(), pa_none(), lin_out[i].raw, &lin_bn[i]));
        TRY(c_head_fwd(*this, op(lin_out[i]), outc[i].wf, params + outc[i].p_b, heat[i], heat64[i], pts ? pts_dev : nullptr,
                               pts ? loss_dev + i : nullptr, B, Hh, Hh, chan, st));
        if (i + 1 < stacks) {
            TRY(conv_fwd(forth[i], op(lin_out[i]), B, Hh, Hh, op(xin[i]), pa_none(), forth_tmp[i], nullptr));
            TRY(conv_fwd(inc[i], pa_plain(heat64[i]), B, Hh, Hh, pa_plain(forth_tmp[i]), pa_none(), xin[i + 1].raw, nullptr));
        }
    }
    if (pts) TRY(pa_launch_loss_out(loss_dev, loss_keep, loss_out_dev, loss_total_out, stacks, st));
    return 0;
}

// hand-written backward of the whole pose net; gradients land in the flat `grads` array
int Net::backward_stack(int i) {
    const int Hh = res / 4;
    const float gscale = PA_GRAD_SCALE / ((float)B * 16.f * (float)Hh * (float)Hh);      // (fp16 build: scaled gradients, common.h)
    const bool inner = i + 1 < stacks;
    if (inner) {
        const PaOperand gx = pa_plain(xin[i + 1].grad);
        TRY(conv_wgrad(inc[i], gx, pa_plain(heat64[i]), B, Hh, Hh));
        TRY(conv_dgrad(inc[i], gx, B, Hh, Hh, pa_none(), pa_none(), ep_plain(), dheat_in[i]));
        TRY(conv_wgrad(forth[i], gx, op(lin_out[i]), B, Hh, Hh));
    }
    TRY(c_heat_grad(*this, heat[i], pts_dev, inner ? dheat_in[i] : nullptr, dheat64[i], gscale, B, Hh, Hh, st));
    const PaOperand gh = pa_plain(dheat64[i]);
    TRY(conv_wgrad(outc[i], gh, op(lin_out[i]), B, Hh, Hh));
    if (inner) {
        TRY(conv_dgrad(outc[i], gh, B, Hh, Hh, pa_none(), pa_none(), ep_plain(), lgrad_tmp[i]));
        TRY(conv_dgrad(forth[i], pa_plain(xin[i + 1].grad), B, Hh, Hh, pa_plain(lgrad_tmp[i]), pa_none(),
                       final_ep(lin_out[i]), lin_out[i].grad));
    } else {
        TRY(conv_dgrad(outc[i], gh, B, Hh, Hh, pa_none(), pa_none(), final_ep(lin_out[i]), lin_out[i].grad));
    }
    TRY(finish_grad(lin_out[i]));
    PaOperand gl = gradop(lin_out[i]);
    bool gl_stored = false;                  // lgrad_tmp[i] is free again: the data gradient stores dz there for the weight gradient
    TRY(conv_dgrad(lin[i], gl, B, Hh, Hh, pa_none(), pa_none(), final_ep(post[i].x3), post[i].x3.grad, lgrad_tmp[i], &gl_stored));
    if (gl_stored) gl = pa_plain(lgrad_tmp[i]);
    TRY(conv_wgrad(lin[i], gl, op(post[i].x3), B, Hh, Hh));
    TRY(finish_grad(post[i].x3));
    post[i].low_of_in = hg[i].low_fusable(*this, 0) ? &hg[i].up[0].x3 : nullptr;          // (post_res reads merged[0]: its conv1 data gradient can emit d up1 too)
    TRY(post[i].bwd(*this, hg[i].out(), pa_none(), true));
    hg[i].low_done[0] = post[i].low_fused;
    TRY(hg[i].bwd(*this, xin[i], inner ? pa_plain(xin[i + 1].grad) : pa_none()));
    return 0;
}

// stem: residual3 <- residual2 <- maxpool <- residual1 <- 7x7 conv
int Net::backward_stem() {
    TRY(res3.bwd(*this, res2.x3, pa_none(), true));
    TRY(finish_grad(res2.x3));
    TRY(res2.bwd(*this, pool0, pa_none(), true));
    TRY(c_maxpool_bwd(*this, pool0.grad, op(res1.x3), pa_none(), final_ep(res1.x3), res1.x3.grad, B, res / 2, res / 2, 128, st));
    TRY(finish_grad(res1.x3));
    TRY(res1.bwd(*this, a0, pa_none(), true));
    TRY(finish_grad(a0));
    TRY(conv_wgrad(stem_conv, gradop(a0), pa_plain(cur_image), B, res / 2, res / 2));
    TRY(reduce_grads());
    return join_meters();               // the step's meters ran beside this pass: later readers on the main stream come behind them
}

int Net::backward_pose() {
    TRY(ensure_streams());
    for (int i = stacks - 1; i >= 0; --i) TRY(backward_stack(i));
    return backward_stem();
}

// every gradient of hourglass `stack` is final once the main chain has enqueued its backward pass (the side branches are joined
// inside it) AND the weight-gradient stream has reduced its slabs: an event on that stream behind one on the main stream
int Net::mark_bucket(int stack) {
    if (stack < 0 || stack >= 16) return 1;
    if (!bucket_events) {
        for (int i = 0; i < 16; ++i) PA_CHECK(hipEventCreateWithFlags(&ev_bucket[i], hipEventDisableTiming));
        PA_CHECK(hipEventCreateWithFlags(&ev_bucket_main, hipEventDisableTiming));
        bucket_events = true;
    }
    TRY(flush_wgrads());
    if (multi_stream && wstream) TRY(flush_red_stash(wstreams[(w_rr + n_w - 1) % n_w]));
    if (multi_stream && wstream && (reduce_early || immediate_reduce)) {
        PA_CHECK(hipEventRecord(ev_bucket_main, st));
        for (int i = 0; i < n_w; ++i) PA_CHECK(hipStreamWaitEvent(wstreams[i], ev_bucket_main, 0));
        for (int i = 1; i < n_w; ++i) {         // (more than one weight-gradient stream: fold them into the first)
            PA_CHECK(hipEventRecord(ev_wdone_x[i], wstreams[i]));
            PA_CHECK(hipStreamWaitEvent(wstreams[0], ev_wdone_x[i], 0));
        }
        PA_CHECK(hipEventRecord(ev_bucket[stack], wstreams[0]));
        return 0;
    }
    return -1;            // slabs are only reduced at the very end in this mode: no early bucket
}

// One training step (forward, loss, backward) as a HIP graph.  Inputs live in the engine's own buffers (img4, pts_dev:
// the caller copies into them on `st` before the launch), every pointer and shape inside is fixed by the layout, so the
// graph is captured once per (mode, bound memory) and replayed.  The capture runs the normal enqueue code: the side streams
// fork from / join into `st` by events, which the capture turns into graph dependencies.
int Net::train_step_graph(bool train) {
    TRY(ensure_streams());
    const int key = (train ? 1 : 0) | (multi_stream ? 2 : 0);
    if (drop_mask || prof.on) { pa_set_error_msg("train_step_graph: not with the occlusion branch / the launch profiler"); return 1; }
    // meters left on the meter stream by an eager step (pa_net_meters_async) read the heat maps / joints this launch overwrites; the
    // join must also happen OUTSIDE a capture (its event was recorded outside)
    TRY(join_meters());
    // (a replayed launch would carry the launch number of the capture: its barrier tags would match the previous replay's granules)
    if (!step_exec || step_key != key) {
        release_graph();
        // the caller's stream may be the legacy default stream, which cannot capture: capture on a stream of our own
        hipStream_t cap = nullptr, caller = st;
        PA_CHECK(hipStreamCreateWithFlags(&cap, hipStreamNonBlocking));
        st = cap;
        hipError_t e = hipStreamBeginCapture(cap, hipStreamCaptureModeRelaxed);
        capturing = e == hipSuccess;          // (extension launches -- any-order weight gradients -- stay out of a capture)
        if (e != hipSuccess) { st = caller; (void)hipStreamDestroy(cap); pa_set_error("hipStreamBeginCapture", e, __FILE__, __LINE__); return (int)e; }
        int rc = forward_pose(nullptr, img4, pts_dev, train, nullptr);
        if (!rc) rc = backward_pose();
        hipGraph_t g = nullptr;
        e = hipStreamEndCapture(cap, &g);
        capturing = false;
        st = caller;
        (void)hipStreamDestroy(cap);
        if (rc) { if (g) (void)hipGraphDestroy(g); return rc; }
        if (e != hipSuccess) { pa_set_error("hipStreamEndCapture", e, __FILE__, __LINE__); return (int)e; }
        step_graph = g;
        PA_CHECK(hipGraphInstantiate(&step_exec, step_graph, nullptr, nullptr, 0));
        step_key = key;
    }
    train_bn = train;
    cur_image = img4;
    heat_peak_valid.assign(stacks, 0);
    PA_CHECK(hipGraphLaunch(step_exec, st));
    return 0;
}

void Net::release_graph() {
    if (step_exec) { (void)hipGraphExecDestroy(step_exec); step_exec = nullptr; }
    if (step_graph) { (void)hipGraphDestroy(step_graph); step_graph = nullptr; }
    step_key = -1;
}

int Net::reduce_grads() {
    TRY(flush_wgrads());
    if (multi_stream && wstream) TRY(flush_red_stash(wstreams[(w_rr + n_w - 1) % n_w]));
    if (multi_stream && wstream) {              // all weight-gradient slabs are complete
        if (!(nosync() && (g_nosync & 8))) {
            for (int i = 0; i < n_w; ++i) {
                PA_CHECK(hipEventRecord(ev_wdone_x[i], wstreams[i]));
                PA_CHECK(hipStreamWaitEvent(st, ev_wdone_x[i], 0));
            }
        }
        if (reduce_early || immediate_reduce) return 0;
    }
    if (immediate_reduce) return 0;             // every layer was reduced right after its weight-gradient launch
    TRY(pa_launch_wgrad_reduce(red_jobs, n_red, red_max, st));
    if (!is_agent) {
        TRY(pa_launch_stem_wgrad_reduce(stem_conv.part, stem_conv.splits, grads + stem_conv.p_w, st, grads + stem_conv.p_b));
    }
    return 0;
}
