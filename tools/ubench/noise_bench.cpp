// Stand-alone timing of bndm_bluenoise (C ABI) in its HBM regime (B = 2, 5, 10 RGB images of 64 px): back-to-back calls
// with caller-owned buffers, per-call wall time -> GB/s of L's lower triangle.  Also checks the result against a plain
// device-side reference product on a few entries.  Profiling aid (tools/ubench/build.sh); not part of the product.
#include "../../include/bndm_hip.h"
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main() {
    const int N = 4096;
    std::vector<float> hL((size_t)N * N, 0.f);
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
    for (int i = 0; i < N; ++i)
        for (int j = 0; j <= i; ++j) hL[(size_t)i * N + j] = rnd() * 0.05f;
    float *L; CK(hipMalloc(&L, hL.size() * 4)); CK(hipMemcpy(L, hL.data(), hL.size() * 4, hipMemcpyHostToDevice));
    const double tri_bytes = 4.0 * N * (N + 1) / 2;
    for (int B : {2, 5, 10}) {
        const int C = 3, n = B * C;
        std::vector<float> hz((size_t)n * N), ha(B, 0.f);
        for (auto &v : hz) v = rnd() * 2.f;
        float *z, *alpha, *o0, *o1, *o2; void *ws;
        const size_t wsb = bndm_bluenoise_workspace_bytes(B, C, 64);
        CK(hipMalloc(&z, hz.size() * 4)); CK(hipMemcpy(z, hz.data(), hz.size() * 4, hipMemcpyHostToDevice));
        CK(hipMalloc(&alpha, B * 4)); CK(hipMemcpy(alpha, ha.data(), B * 4, hipMemcpyHostToDevice));
        CK(hipMalloc(&o0, hz.size() * 4)); CK(hipMalloc(&o1, hz.size() * 4)); CK(hipMalloc(&o2, hz.size() * 4));
        CK(hipMalloc(&ws, wsb));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const int reps = 100;
        float best = 1e9f;
        for (int pass = 0; pass < 4; ++pass) {
            CK(hipEventRecord(e0, 0));
            for (int r = 0; r < reps; ++r)
                if (bndm_bluenoise(L, 0, z, BNDM_Z_COLUMNS, alpha, o0, o1, o2, B, 0, B, C, 64, BNDM_NOISE_BLEND, ws, wsb, nullptr)) {
                    printf("error: %s\n", bndm_last_error()); return 1;
                }
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (pass && ms < best) best = ms;
        }
        // check: noise_bn[col][i] = sum_j L[i][j] z[col][j] (alpha = 0 -> noise == noise_bn), fp64 reference on 64 entries
        std::vector<float> ho(hz.size());
        CK(hipMemcpy(ho.data(), o0, ho.size() * 4, hipMemcpyDeviceToHost));
        double maxerr = 0, maxref = 0;
        for (int t = 0; t < 64; ++t) {
            const int col = t % n, i = (t * 977 + 4095 * (t & 1)) % N;
            double ref = 0;
            for (int j = 0; j <= i; ++j) ref += (double)hL[(size_t)i * N + j] * hz[(size_t)col * N + j];
            maxerr = fmax(maxerr, fabs(ref - ho[(size_t)col * N + i]));
            maxref = fmax(maxref, fabs(ref));
        }
        const double us = best * 1e3 / reps;
        printf("B=%2d (%2d columns): %6.2f us/call  %7.1f GB/s of L   max|err| %.2e (max|ref| %.2f)\n", B, n, us, tri_bytes / us * 1e-3, maxerr, maxref);
        CK(hipFree(z)); CK(hipFree(alpha)); CK(hipFree(o0)); CK(hipFree(o1)); CK(hipFree(o2)); CK(hipFree(ws));
    }
    return 0;
}
