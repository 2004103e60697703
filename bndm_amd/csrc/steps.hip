// Sampler-step kernels (HBM-bound elementwise work, float4-vectorised, grid-stride).
//   iadb_step  : utils.py:216-226 / iadb_bn.py:323-344 / latent_iadb_bn_diffusers.py:107-117
//   ddim_step  : DDIMScheduler.step as called at ddim_diffusers.py:680
//   export_u8  : iadb_bn.py:815-816 (truncate) / ddim_diffusers.py:687-688 (round half even)
//   train_targets : iadb_bn.py:915,946-956 / latent_iadb_bn_diffusers.py:606-621 (forward blend + regression targets)
#include "common.hpp"

// bit-exact parity with torch's separate mul/add kernels: products and sums are written as plain
// operators and fma contraction is forbidden in this file (header intrinsics would still fuse)
#pragma clang fp contract(off)

namespace bndm {
namespace {

// x[b,c,:] += da*d[b,c,:] + dg*d[b,C+c,:]; operation order as torch evaluates
// x + da*d1 + dg*d2 == (x + da*d1) + dg*d2, products rounded separately (no fma contraction).
__global__ __launch_bounds__(256) void iadb_step_kernel(float *__restrict__ x, const float *__restrict__ d,
                                                        float da, float dg, int C, int Cout, int HW4,
                                                        size_t total4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4;
         i += (size_t)gridDim.x * blockDim.x) {
        const size_t bc = i / HW4;
        const int p = (int)(i - bc * HW4);
        const size_t b = bc / C;
        const int c = (int)(bc - b * C);
        f32x4 xv = reinterpret_cast<f32x4 *>(x)[i];
        const f32x4 d1 = reinterpret_cast<const f32x4 *>(d)[(b * Cout + c) * HW4 + p];
#pragma unroll
        for (int e = 0; e < 4; ++e) xv[e] = xv[e] + da * d1[e];
        if (Cout == 2 * C) {
            const f32x4 d2 = reinterpret_cast<const f32x4 *>(d)[(b * Cout + C + c) * HW4 + p];
#pragma unroll
            for (int e = 0; e < 4; ++e) xv[e] = xv[e] + dg * d2[e];
        }
        store_wt(reinterpret_cast<f32x4 *>(x) + i, xv);
    }
}

__global__ __launch_bounds__(256) void ddim_step_kernel(float *__restrict__ x, const float *__restrict__ eps,
                                                        float sqrt_at, float sqrt_1m_at, float sqrt_ap,
                                                        float sqrt_1m_ap, float clip, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (size_t)gridDim.x * blockDim.x) {
        const float e = eps[i];
        float x0 = (x[i] - sqrt_1m_at * e) / sqrt_at;
        if (clip > 0.f) x0 = fminf(fmaxf(x0, -clip), clip);
        x[i] = sqrt_ap * x0 + sqrt_1m_ap * e;
    }
}

__global__ __launch_bounds__(256) void export_u8_kernel(const float *__restrict__ x, uint8_t *__restrict__ out,
                                                        int C, int HW, int rounding, size_t total) {
    // one thread per output byte [b][p][c]  (NHWC)
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const size_t bp = i / C;
        const int p = (int)(bp % HW);
        const size_t b = bp / HW;
        const float v = x[(b * C + c) * HW + p];
        float y;
        if (rounding == 0) {
            y = (v + 1.0f) / 2.0f;                               // (x + 1) / 2.0
            y = fminf(fmaxf(y, 0.f), 1.f);
            y = y * 255.0f;                                      // numpy f32 * 255 -> astype(uint8)
            out[i] = (uint8_t)(int)y;
        } else {
            y = v / 2.0f + 0.5f;                                 // x / 2 + 0.5
            y = fminf(fmaxf(y, 0.f), 1.f);
            y = rintf(y * 255.0f);                               // .round() is half-to-even
            out[i] = (uint8_t)(int)y;
        }
    }
}

// Training-time noise injection, everything after get_noise_v2: per sample b (alpha = a[b], alpha_{t-1} = ap[b])
//   x_alpha = a*x0 + (1-a)*x1      tar1 = x1 - x0      tar2 = ap*(bn - wn)      tar = tar1 + tar2
// in torch's evaluation order with separately rounded products (x1 is the data, x0 the noise).
__global__ __launch_bounds__(256) void train_targets_kernel(const float *__restrict__ x0, const float *__restrict__ x1,
                                                            const float *__restrict__ bn, const float *__restrict__ wn,
                                                            const float *__restrict__ a, const float *__restrict__ ap,
                                                            float *__restrict__ x_alpha, float *__restrict__ tar1,
                                                            float *__restrict__ tar2, float *__restrict__ tar,
                                                            size_t per4, size_t total4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4;
         i += (size_t)gridDim.x * blockDim.x) {
        const size_t b = i / per4;
        const float al = a[b], om = 1.0f - al;
        const f32x4 n0 = reinterpret_cast<const f32x4 *>(x0)[i], d1 = reinterpret_cast<const f32x4 *>(x1)[i];
        f32x4 t1, t2 = {0.f, 0.f, 0.f, 0.f};
        if (x_alpha) {
            f32x4 xa;
#pragma unroll
            for (int e = 0; e < 4; ++e) xa[e] = al * n0[e] + om * d1[e];
            reinterpret_cast<f32x4 *>(x_alpha)[i] = xa;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) t1[e] = d1[e] - n0[e];
        if (tar1) reinterpret_cast<f32x4 *>(tar1)[i] = t1;
        if (bn) {
            const float apv = ap[b];
            const f32x4 nb = reinterpret_cast<const f32x4 *>(bn)[i], nw = reinterpret_cast<const f32x4 *>(wn)[i];
#pragma unroll
            for (int e = 0; e < 4; ++e) t2[e] = apv * (nb[e] - nw[e]);
            if (tar2) reinterpret_cast<f32x4 *>(tar2)[i] = t2;
        }
        if (tar) {
            f32x4 ts;
#pragma unroll
            for (int e = 0; e < 4; ++e) ts[e] = bn ? t1[e] + t2[e] : t1[e];
            reinterpret_cast<f32x4 *>(tar)[i] = ts;
        }
    }
}

inline int grid_for(size_t n) {
    size_t b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

}  // namespace
}  // namespace bndm

using namespace bndm;

extern "C" int bndm_iadb_step(float *x, const float *d, float da, float dg, int B, int C, int Cout,
                              int HW, void *stream) {
    BNDM_REQUIRE(x && d, "bndm_iadb_step: NULL tensor");
    BNDM_REQUIRE(Cout == C || Cout == 2 * C,
                 "bndm_iadb_step: out_channel %d with %d image channels (reference raises "
                 "NotImplementedError, utils.py:223)", Cout, C);
    BNDM_REQUIRE(HW % 4 == 0, "bndm_iadb_step: H*W must be a multiple of 4");
    if (B <= 0) return 0;
    const size_t total4 = (size_t)B * C * (HW / 4);
    hipLaunchKernelGGL(iadb_step_kernel, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)stream, x, d, da,
                       dg, C, Cout, HW / 4, total4);
    return launch_status("iadb_step");
}

extern "C" int bndm_ddim_step(float *x, const float *eps, float sqrt_at, float sqrt_1m_at, float sqrt_ap,
                              float sqrt_1m_ap, float clip, size_t n, void *stream) {
    BNDM_REQUIRE(x && eps, "bndm_ddim_step: NULL tensor");
    if (n == 0) return 0;
    hipLaunchKernelGGL(ddim_step_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, eps, sqrt_at,
                       sqrt_1m_at, sqrt_ap, sqrt_1m_ap, clip, n);
    return launch_status("ddim_step");
}

extern "C" int bndm_export_u8(const float *x, uint8_t *out, int B, int C, int HW, int rounding,
                              void *stream) {
    BNDM_REQUIRE(x && out, "bndm_export_u8: NULL tensor");
    BNDM_REQUIRE(rounding == 0 || rounding == 1, "bndm_export_u8: rounding must be 0 or 1");
    const size_t total = (size_t)B * C * HW;
    if (total == 0) return 0;
    hipLaunchKernelGGL(export_u8_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, out, C,
                       HW, rounding, total);
    return launch_status("export_u8");
}

extern "C" int bndm_iadb_train_targets(const float *x0, const float *x1, const float *noise_bn, const float *noise_wn,
                                       const float *alpha, const float *alpha_prev, float *x_alpha, float *tar1,
                                       float *tar2, float *tar, int B, size_t per_sample, void *stream) {
    BNDM_REQUIRE(x0 && x1 && alpha, "bndm_iadb_train_targets: NULL tensor");
    BNDM_REQUIRE((noise_bn == nullptr) == (noise_wn == nullptr), "bndm_iadb_train_targets: noise_bn and noise_wn go together");
    BNDM_REQUIRE(!noise_bn || alpha_prev, "bndm_iadb_train_targets: alpha_prev is required with noise_bn / noise_wn");
    BNDM_REQUIRE(!tar2 || noise_bn, "bndm_iadb_train_targets: tar2 needs noise_bn / noise_wn");
    BNDM_REQUIRE(B >= 0 && per_sample % 4 == 0, "bndm_iadb_train_targets: C*H*W must be a multiple of 4");
    const size_t per4 = per_sample / 4, total4 = per4 * (size_t)B;
    if (total4 == 0) return 0;
    hipLaunchKernelGGL(bndm::train_targets_kernel, dim3(bndm::grid_for(total4)), dim3(256), 0, (hipStream_t)stream, x0,
                       x1, noise_bn, noise_wn, alpha, alpha_prev, x_alpha, tar1, tar2, tar, per4, total4);
    return bndm::launch_status("train_targets");
}
