// Stand-alone timing of single conv_s launches (csrc/unet_tail.hip) on the <= 8x8 layer shapes at batch 64, with the
// s_memtime phase marks of workgroup 0.  Profiling aid, built by tools/ubench/build.sh; not part of the product.
#include "../../bndm_amd/csrc/unet_kernels.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
using namespace bndm;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Case { const char *name; int B, H, C1, C2, Cout, kind, qkv, nreq, gs; };

static void run(const Case &c) {
    const int HW = c.H * c.H, M = c.B * HW, NB = c.qkv ? 3 : 1, D = tail_ring_depth(NB), rows = c.qkv ? 3 * c.Cout : c.Cout;
    std::vector<TailSeg> segs{TailSeg{c.kind, c.C1}};
    if (c.C2) segs.push_back(TailSeg{c.kind, c.C2});
    const TailPlan plan = build_tail_plan(segs, rows, NB, D, [](int, int r, int ch, int t) { return 0.01f * ((r * 7 + ch * 3 + t) % 13 - 6); },
                                          [&](int nt, int nb, int n) { return c.qkv ? nb * c.Cout + nt * 32 + n : nt * 32 + n; });
    std::vector<_Float16> w16(plan.wgt.size());
    for (size_t i = 0; i < w16.size(); ++i) w16[i] = (_Float16)plan.wgt[i];
    const int ncopy = 24;
    char *W; CK(hipMalloc(&W, w16.size() * 2 * ncopy));
    for (int k = 0; k < ncopy; ++k) CK(hipMemcpy(W + w16.size() * 2 * k, w16.data(), w16.size() * 2, hipMemcpyHostToDevice));
    void *A1, *A2 = nullptr, *out, *n1, *n2, *desc, *rounds; float *gamma; unsigned long long *dbg;
    const int srows = c.kind == TAIL_SEG_3x3_S2 ? 4 * M : (c.kind == TAIL_SEG_3x3_UP ? M / 4 : M);
    CK(hipMalloc(&A1, (size_t)srows * c.C1 * 2)); CK(hipMemset(A1, 0x2c, (size_t)srows * c.C1 * 2));
    if (c.C2) { CK(hipMalloc(&A2, (size_t)srows * c.C2 * 2)); CK(hipMemset(A2, 0x2c, (size_t)srows * c.C2 * 2)); }
    CK(hipMalloc(&out, (size_t)M * rows * 2)); CK(hipMalloc(&n1, (size_t)M * rows * 2)); CK(hipMalloc(&n2, (size_t)M * rows * 2));
    CK(hipMalloc(&gamma, 4096 * 4)); CK(hipMemset(gamma, 0, 4096 * 4));
    CK(hipMalloc(&desc, plan.desc.size() * 4)); CK(hipMemcpy(desc, plan.desc.data(), plan.desc.size() * 4, hipMemcpyHostToDevice));
    std::vector<TailRound> rt(plan.nrounds);
    for (int i = 0; i < plan.nrounds; ++i) {
        const TailPlanRound &p = plan.rounds[i];
        rt[i] = TailRound{p.seg ? A2 : A1, (p.seg ? c.C2 : c.C1) * 2, p.c0 * 2, p.mode, p.phase, p.nsub, 0};
    }
    CK(hipMalloc(&rounds, rt.size() * sizeof(TailRound))); CK(hipMemcpy(rounds, rt.data(), rt.size() * sizeof(TailRound), hipMemcpyHostToDevice));
    CK(hipMalloc(&dbg, 32 * 8)); CK(hipMemset(dbg, 0, 32 * 8));
    TailArgs a{};
    a.desc = (const uint32_t *)desc; a.rounds = (const TailRound *)rounds; a.nrounds = plan.nrounds; a.maxsteps = plan.maxsteps;
    a.r0 = rt[0]; if (plan.nrounds > 1) a.r1 = rt[1];
    for (int i = 0; i < 8; ++i) a.nuse[i] = plan.nuse[i];
    a.tile_bytes = (long long)plan.tile_elems * 2; a.wave_bytes = (int)(plan.wave_elems * 2);
    a.B = c.B; a.hwlog = 31 - __builtin_clz(HW); a.wlog = 31 - __builtin_clz(c.H); a.Cout = c.Cout; a.ntn = c.Cout / 32;
    a.bias = gamma; a.eps = 1e-5f; a.epi = c.qkv ? TAIL_EPI_ATTN : TAIL_EPI_CONV;
    if (c.qkv) a.attn_out = out; else { a.raw_out = out; a.nreq = c.nreq; a.req[0] = TailNorm{n1, gamma, gamma, c.gs, 1}; a.req[1] = TailNorm{n2, gamma, gamma, 2 * c.gs > 32 ? 32 : 2 * c.gs, 1}; }
    int TM = 64;
    if (!c.qkv && HW == 64 && (long long)c.B * HW / 128 * a.ntn >= 192) TM = 128;
    if (getenv("TM64")) TM = 64;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 48;
    float best = 1e9f;
    for (int hot = 1; hot >= 0; --hot) {
        best = 1e9f;
        for (int pass = 0; pass < 6; ++pass) {
            CK(hipEventRecord(e0, 0));
            for (int r = 0; r < reps; ++r) {
                TailArgs q = a;
                q.wgt = W + w16.size() * 2 * (hot ? 0 : r % ncopy);
                if (launch_conv_tail(BNDM_DTYPE_F16, TM, NB, q, 0)) { printf("launch failed: %s\n", bndm_last_error()); exit(1); }
            }
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (pass >= 3 && ms < best) best = ms;
        }
        const double us = best * 1e3 / reps, fl = 2.0 * M * rows * 1.0 * (c.kind == TAIL_SEG_1x1 ? 1 : 9) * (c.C1 + c.C2);
        printf("%-30s TM=%3d M=%5d N=%4d K=%5d grid=%4d rounds=%2d steps/wave=%3d %s: %7.2f us/launch %6.1f TF/s  W %5.2f MB\n", c.name, TM, M, rows,
               (c.kind == TAIL_SEG_1x1 ? 1 : 9) * (c.C1 + c.C2), (M + TM - 1) / TM * a.ntn, plan.nrounds, plan.maxsteps, hot ? "hot " : "cold", us, fl / us * 1e-6, w16.size() * 2 / 1e6);
    }
    // phase marks (one launch, cold weights)
    TailArgs q = a; q.wgt = W + w16.size() * 2 * 7; q.dbg = dbg;
    launch_conv_tail(BNDM_DTYPE_F16, TM, NB, q, 0);
    CK(hipDeviceSynchronize());
    unsigned long long h[32]; CK(hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost));
    for (int wv = 0; wv < 2; ++wv) {
        printf("    marks wave %d (clk since start): ", wv ? 7 : 0);
        for (int k = 1; k < 8; ++k) printf(" %6lld", h[16 * wv + k] ? (long long)(h[16 * wv + k] - h[0]) : -1LL);
        printf("   [issue | landed+barrier | loop | drain | reduce | items | norm]   prologue:");
        for (int k = 8; k < 12; ++k) printf(" %5lld", h[16 * wv + k] ? (long long)(h[16 * wv + k] - h[0]) : -1LL);
        printf(" [rows req | dma0 | weights | dma1]\n");
    }
    CK(hipFree(W)); CK(hipFree(A1)); if (A2) CK(hipFree(A2)); CK(hipFree(out)); CK(hipFree(n1)); CK(hipFree(n2)); CK(hipFree(gamma));
    CK(hipFree(desc)); CK(hipFree(rounds)); CK(hipFree(dbg));
}

int main(int argc, char **argv) {
    const Case cases[] = {
        {"8x8 conv 256->256 (2 norms)", 64, 8, 256, 0, 256, TAIL_SEG_3x3, 0, 2, 8},
        {"8x8 conv 768->256", 64, 8, 768, 0, 256, TAIL_SEG_3x3, 0, 1, 8},
        {"4x4 conv 512->512 (2 norms)", 64, 4, 512, 0, 512, TAIL_SEG_3x3, 0, 2, 16},
        {"4x4 conv 1024->512", 64, 4, 512, 512, 512, TAIL_SEG_3x3, 0, 1, 16},
        {"2x2 conv 512->512 (2 norms)", 64, 2, 512, 0, 512, TAIL_SEG_3x3, 0, 2, 16},
        {"2x2 conv 1024->512 (1 norm)", 64, 2, 512, 512, 512, TAIL_SEG_3x3, 0, 1, 16},
        {"4x4 1x1 512->512 (to_out, 2n)", 64, 4, 512, 0, 512, TAIL_SEG_1x1, 0, 2, 16},
        {"4x4 1x1 512->512 (no norm)", 64, 4, 512, 0, 512, TAIL_SEG_1x1, 0, 0, 16},
        {"4x4 qkv+attn 512", 64, 4, 512, 0, 512, TAIL_SEG_1x1, 1, 0, 16},
        {"2x2 qkv+attn 512", 64, 2, 512, 0, 512, TAIL_SEG_1x1, 1, 0, 16},
        {"4x4->2x2 stride-2 512", 64, 2, 512, 0, 512, TAIL_SEG_3x3_S2, 0, 2, 16},
        {"16x16->8x8 stride-2 256", 64, 8, 256, 0, 256, TAIL_SEG_3x3_S2, 0, 2, 8},
        {"4x4->8x8 up 512", 64, 8, 512, 0, 512, TAIL_SEG_3x3_UP, 0, 1, 16},
    };
    const int only = argc > 1 ? atoi(argv[1]) : -1;
    for (int i = 0; i < (int)(sizeof(cases) / sizeof(cases[0])); ++i)
        if (only < 0 || i == only) run(cases[i]);
    return 0;
}
