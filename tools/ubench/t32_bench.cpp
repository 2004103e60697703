// Stand-alone timing of single conv_t32 launches (csrc/unet_conv32.hip) as a function of the GRID: one layer shape at
// batch 1 .. 64.  Profiling aid, built by tools/ubench/build.sh; not part of the product.
//
// Why: a workgroup's life is serial (prologue P, K loop L, epilogue E) and both workgroups of a CU run in phase, so a
// launch is ~ rounds x (P + L + E).  The sweep separates the terms without instrumentation:
//     64x64, TH=16: 16 tiles per sample ->  B=16: 256 workgroups = one per CU, alone;  B=32: two per CU, one round;
//     B=64: two rounds;  B=1..8: a fraction of the chip (no memory-system contention: "P + L + E alone").
// If T(B=16) ~ T(B=1) the fixed parts are latency chains; if T(B=16) >> T(B=1) they are launch-wide memory bursts.
// K is varied too (Cin = 128 / 256 = K 1152 / 2304): dT/dK is the loop, the intercept is P + E.
//
//   t32_bench [TH] [normed 0/1] [resid 0/1]        (env T32_H: resolution, default 64)
#include "../../bndm_amd/csrc/unet_kernels.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
using namespace bndm;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static uint16_t f16_bits(float f) {
    _Float16 h = (_Float16)f;
    uint16_t u;
    memcpy(&u, &h, 2);
    return u;
}

static double run(int TH, int B, int H, int Cin, int Cout, bool normed, bool resid) {
    const int W = H, HW = H * W;
    FusedArgs a{};
    a.nseg = 1;
    a.seg[0].C = Cin;
    a.seg[0].taps = 9;
    a.seg[0].up = 0;
    a.seg[0].ss_off = normed ? 0 : -1;
    a.ssC = normed ? Cin : 0;
    a.silu = 1;
    a.H = H;
    a.W = W;
    a.Cout = Cout;
    a.B = B;
    a.nco = 128;
    a.Ktot = 9 * Cin;
    const std::vector<float> wp = pack_weights_t32(a.seg, 1, Cout, [](int, int co, int c, int t) { return 0.02f * ((co * 5 + c * 3 + t) % 11 - 5); });
    std::vector<uint16_t> w16(wp.size());
    for (size_t i = 0; i < wp.size(); ++i) w16[i] = f16_bits(wp[i]);
    void *Wd, *x, *out, *res = nullptr, *zeros;
    float *bias, *stats_in, *stats_out, *gamma;
    CK(hipMalloc(&Wd, w16.size() * 2));
    CK(hipMemcpy(Wd, w16.data(), w16.size() * 2, hipMemcpyHostToDevice));
    const size_t xin = (size_t)B * HW * Cin, xout = (size_t)B * HW * Cout;
    std::vector<uint16_t> hx(xin);
    unsigned s = 12345u;
    for (size_t i = 0; i < xin; ++i) {
        s = s * 1664525u + 1013904223u;
        hx[i] = f16_bits(((int)(s >> 16) % 2001 - 1000) * 1e-3f);          // uniform [-1, 1): random data, not zeros (DVFS)
    }
    CK(hipMalloc(&x, xin * 2));
    CK(hipMemcpy(x, hx.data(), xin * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&out, xout * 2));
    if (resid) {
        CK(hipMalloc(&res, xout * 2));
        CK(hipMemset(res, 0x2c, xout * 2));
    }
    CK(hipMalloc(&zeros, 256));
    CK(hipMemset(zeros, 0, 256));
    CK(hipMalloc(&bias, Cout * 4));
    CK(hipMemset(bias, 0, Cout * 4));
    const int tps_in = conv_t32_tiles_per_sample(16, H, W), tps_out = conv_t32_tiles_per_sample(TH, H, W);
    // input statistics as a producer with 256-pixel tiles would have left them: sum 0, sum of squares = pixels / 3 per
    // channel (variance 1/3 of the uniform data above) -- sized for the per-channel layout (the per-pair layout needs half)
    std::vector<float> hs((size_t)B * tps_in * Cin * 2);
    for (size_t i = 0; i < hs.size(); i += 2) {
        hs[i] = 0.f;
        hs[i + 1] = (float)(HW / tps_in) / 3.f;
    }
    CK(hipMalloc(&stats_in, hs.size() * 4));
    CK(hipMemcpy(stats_in, hs.data(), hs.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&stats_out, (size_t)B * tps_out * Cout * 2 * 4));
    std::vector<float> ones(1024, 1.f);
    CK(hipMalloc(&gamma, 1024 * 4));
    CK(hipMemcpy(gamma, ones.data(), 1024 * 4, hipMemcpyHostToDevice));
    a.seg[0].src = x;
    a.Wgt = Wd;
    a.bias = bias;
    a.resid = res;
    a.out = out;
    a.stats = stats_out;
    a.zeros = zeros;
    if (normed) {
        a.ss = (const float *)zeros;
        a.gn_p1 = stats_in;
        a.gn_ns1 = tps_in;
        a.gn_C1 = Cin;
        a.gn_HW = HW;
        a.gn_gamma = gamma;
        a.gn_beta = bias;          // zeros (Cout >= ... only the first Cin entries are read; both are >= 128 floats)
        a.gn_eps = 1e-5f;
    }
    if (!conv_t32_supports(a)) {
        printf("unsupported\n");
        exit(1);
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int reps = 20;
    float best = 1e9f;
    for (int pass = 0; pass < 5; ++pass) {
        CK(hipEventRecord(e0, 0));
        for (int r = 0; r < reps; ++r)
            if (launch_conv_t32(BNDM_DTYPE_F16, TH, a, 0)) {
                printf("launch failed: %s\n", bndm_last_error());
                exit(1);
            }
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (pass && ms / reps < best) best = ms / reps;
    }
    for (void *p : {Wd, x, out, res, zeros, (void *)bias, (void *)stats_in, (void *)stats_out, (void *)gamma})
        if (p) CK(hipFree(p));
    return best * 1e3;      // us per launch (back to back: includes the kernel boundary)
}

int main(int argc, char **argv) {
    const int TH = argc > 1 ? atoi(argv[1]) : 16;
    const bool normed = argc > 2 ? atoi(argv[2]) != 0 : true, resid = argc > 3 ? atoi(argv[3]) != 0 : false;
    const int H = getenv("T32_H") ? atoi(getenv("T32_H")) : 64;
    printf("conv_t32<TH=%d> %dx%d, Cout 128, GroupNorm %d, residual %d: us per launch (back to back), TFLOP/s\n", TH, H, H, normed, resid);
    printf("%6s %8s | %12s %12s | %12s\n", "B", "tiles", "K=1152", "K=2304", "dT per 1152");
    for (int B : {1, 2, 4, 8, 16, 32, 64}) {
        const int tiles = B * conv_t32_tiles_per_sample(TH, H, H);
        const double t1 = run(TH, B, H, 128, 128, normed, resid), t2 = run(TH, B, H, 256, 128, normed, resid);
        const double fl = 2.0 * B * H * H * 128.0;
        printf("%6d %8d | %7.1f %4.0f | %7.1f %4.0f | %7.1f\n", B, tiles, t1, fl * 1152 / (t1 * 1e-6) / 1e12, t2, fl * 2304 / (t2 * 1e-6) / 1e12,
               t2 - t1);
    }
    return 0;
}
