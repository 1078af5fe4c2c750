// Stand-alone timing harness of the fused low-resolution launch (csrc/lowres_fused.hip) on random data: the real program shape
// (9 residual blocks, 3 pools, 2 upsample-adds at 16 / 8 / 4 pixel maps, chan 256, B images), no parity (tests/ do that).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I ../../pose_adv_aug_amd/csrc lowres_bench.cpp -o lowres_bench
#include "lowres_fused.hip"
#include <vector>
#include <stdio.h>
#include <string.h>
void pa_set_error_msg(const char* m) { printf("error: %s\n", m); }
void pa_set_error(const char*, hipError_t, const char*, int) {}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static char* g_pool; static size_t g_off;
template <class T> T* take(size_t n) { g_off = (g_off + 255) & ~(size_t)255; T* p = reinterpret_cast<T*>(g_pool + g_off); g_off += n * sizeof(T); return p; }
struct Tn { bf16* p; float* k0; float* k1; int H, C; };
int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 24, C = 256, Cm = 128;
    CK(hipMalloc(&g_pool, (size_t)1 << 30)); CK(hipMemset(g_pool, 0, (size_t)1 << 30));
    std::vector<unsigned short> h((size_t)1 << 24);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned short)(0x3c00 + (i * 2654435761u >> 22) % 512);        // bf16 values around 0.01 .. 0.03
    auto tensor = [&](int H, int Ch, bool bn) { Tn t; t.H = H; t.C = Ch; t.p = take<bf16>((size_t)B * H * H * Ch);
        hipMemcpy(t.p, h.data(), std::min((size_t)B * H * H * Ch * 2, h.size() * 2), hipMemcpyHostToDevice);
        t.k0 = bn ? take<float>(Ch) : nullptr; t.k1 = bn ? take<float>(Ch) : nullptr; return t; };
    std::vector<LrOp> prog;
    auto conv = [&](Tn in, Tn* add, Tn out, int taps, int src_lds, int dst_lds) {
        LrOp o; memset(&o, 0, sizeof o); o.src_lds = src_lds; o.dst_lds = dst_lds; o.type = LR_CONV; o.H = o.W = out.H; o.Cin = in.C; o.Cout = out.C; o.taps = taps;
        o.in = in.p; o.in_k0 = in.k0; o.in_k1 = in.k1; if (add) { o.add = add->p; o.add_k0 = add->k0; o.add_k1 = add->k1; }
        bf16* w = take<bf16>((size_t)out.C * taps * in.C); hipMemcpy(w, h.data(), (size_t)out.C * taps * in.C * 2, hipMemcpyHostToDevice);
        o.w = w; o.bias = take<float>(out.C); o.out = out.p; o.has_bn = 1;
        o.bn.gamma = take<float>(out.C); o.bn.beta = take<float>(out.C); o.bn.rmean = take<float>(out.C); o.bn.rvar = take<float>(out.C);
        o.bn.scale = out.k0; o.bn.shift = out.k1; o.bn.mean = take<float>(out.C); o.bn.invstd = take<float>(out.C);
        prog.push_back(o); };
    auto block = [&](Tn in) { Tn x1 = tensor(in.H, Cm, true), x2 = tensor(in.H, Cm, true), x3 = tensor(in.H, C, true);
        conv(in, nullptr, x1, 1, -1, 1); conv(x1, nullptr, x2, 9, 1, 0); conv(x2, &in, x3, 1, 0, -1); return x3; };
    auto pool = [&](Tn in) { Tn out = tensor(in.H / 2, C, false); LrOp o; memset(&o, 0, sizeof o); o.type = LR_POOL; o.H = o.W = out.H; o.Cin = o.Cout = C; o.taps = 1;
        o.in = in.p; o.in_k0 = in.k0; o.in_k1 = in.k1; o.out = out.p; prog.push_back(o); return out; };
    auto upadd = [&](Tn low, Tn sk) { Tn out = tensor(sk.H, C, false); LrOp o; memset(&o, 0, sizeof o); o.type = LR_UPADD; o.H = o.W = out.H; o.Cin = o.Cout = C; o.taps = 1;
        o.in = low.p; o.in_k0 = low.k0; o.in_k1 = low.k1; o.add = sk.p; o.add_k0 = sk.k0; o.add_k1 = sk.k1; o.out = out.p; prog.push_back(o); return out; };
    Tn p1 = tensor(16, C, false), d1 = block(p1), s2 = block(d1), p2 = pool(d1), d2 = block(p2), s3 = block(d2), p3 = pool(d2), d3 = block(p3), nk = block(d3), u3 = block(nk);
    Tn m3 = upadd(u3, s3), u2 = block(m3), m2 = upadd(u2, s2), u1 = block(m2);
    (void)u1;
    LrOp* ops = take<LrOp>(prog.size()); CK(hipMemcpy(ops, prog.data(), prog.size() * sizeof(LrOp), hipMemcpyHostToDevice));
    LrLaunch L; L.rows = take<float2>(2 * 256 * 256 * 2); L.launch_id = 0; L.counter = take<unsigned>(64); L.batch = (float)B; L.momentum = 0.1f; L.eps = 1e-5f; L.update_running = 1;
    long long* timing = take<long long>(24); L.timing = nullptr;
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int N = 20;
    for (int pass = 0; pass < 3; ++pass) {
        L.timing = pass == 2 ? timing : nullptr;
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < N; ++i) { ++L.launch_id; if (pa_launch_lowres_fwd(ops, (int)prog.size(), L, B, 256, st)) return 1; }
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (pass) printf("%s: %.1f us per launch (B = %d, %zu steps)\n", pass == 2 ? "timed" : "plain", 1e3 * ms / N, B, prog.size());
    }
    long long t[24]; CK(hipMemcpy(t, timing, sizeof t, hipMemcpyDeviceToHost));
    const char* names[8] = {"consts", "stage/bn", "kloop", "epilogue", "sync+pub", "wout+wait", "collect", "pool/up"};
    printf("%-6s", "map"); for (int p = 0; p < 8; ++p) printf("%10s", names[p]); printf("%10s\n", "sum");
    double tot = 0;
    for (int l = 0; l < 3; ++l) { double s = 0; printf("%-6s", l == 0 ? "16x16" : (l == 1 ? "8x8" : "4x4")); for (int p = 0; p < 8; ++p) { printf("%10.0f", (double)t[l * 8 + p] / N); s += (double)t[l * 8 + p] / N; } printf("%10.0f\n", s); tot += s; }
    printf("total %.0f cycles per launch\n", tot);
    return 0;
}
