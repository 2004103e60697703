// GroupNorm helpers of the fused path:
//   gn_finalize2  per-(sample, channel) scale / shift of GroupNorm(32)(cat(x1, x2)) from per-tile partial sums (the
//                 producing convolutions' epilogues write them), consumed by conv_t32's in-patch normalisation;
//   gn_small      statistics + apply (+SiLU) in ONE launch for the <= 8x8 layers, optionally summing the producing
//                 convolution's split-K slabs on the way in (the sum it stores is bit-identical to splitk_reduce's).
#include "unet_kernels.hpp"
#include "unet_types.hpp"

namespace bndm {
namespace {

// cat(x1, x2) statistics from per-tensor partial sums; a block owns 4 groups of one sample: grid (8, B)
__global__ __launch_bounds__(128) void gn_finalize2_kernel(const float *__restrict__ p1, int nslab1, int C1,
                                                           const float *__restrict__ p2, int nslab2, int C2, int HW,
                                                           int groups, float eps, const float *__restrict__ gamma,
                                                           const float *__restrict__ beta,
                                                           float *__restrict__ scale_shift) {
    __shared__ float cs[256], css[256];
    const int sub = blockIdx.x, b = blockIdx.y, C = C1 + C2, Cb = C >> 3;
    for (int cl = threadIdx.x; cl < Cb; cl += blockDim.x) {
        const int c = sub * Cb + cl;
        const float *p;
        int ns, Cs, cc;
        if (c < C1) { p = p1; ns = nslab1; Cs = C1; cc = c; } else { p = p2; ns = nslab2; Cs = C2; cc = c - C1; }
        // loads in batches of 8 (independent, in flight together); the sums keep the slab order
        float s = 0, q = 0;
        for (int k0 = 0; k0 < ns; k0 += 8) {
            float2 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (k0 + k < ns) v[k] = *reinterpret_cast<const float2 *>(p + ((size_t)(b * ns + k0 + k) * Cs + cc) * 2);
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (k0 + k < ns) {
                    s += v[k].x;
                    q += v[k].y;
                }
        }
        cs[cl] = s;
        css[cl] = q;
    }
    __syncthreads();
    const int Cg = C / groups;
    for (int cl = threadIdx.x; cl < Cb; cl += blockDim.x) {
        const int g0 = (cl / Cg) * Cg, c = sub * Cb + cl;
        double s = 0, q = 0;
        for (int k = 0; k < Cg; ++k) {
            s += cs[g0 + k];
            q += css[g0 + k];
        }
        const double n = (double)Cg * HW;
        const double mean = s / n;
        double var = q / n - mean * mean;
        var = var > 0 ? var : 0;
        const float rstd = (float)(1.0 / sqrt(var + (double)eps));
        const float sc = rstd * gamma[c];
        scale_shift[((size_t)b * 2 + 0) * C + c] = sc;
        scale_shift[((size_t)b * 2 + 1) * C + c] = beta[c] - (float)mean * sc;
    }
}

// GroupNorm(32) (+SiLU) of cat(x1, x2) for the low-resolution layers (a sample's tensor is at most
// ~100 KB and stays in L2): statistics and application in ONE launch.  Groups are independent, so a
// block owns 4 groups (C/8 channels) of one sample: grid (8, B).
constexpr int GN_SMALL_KEEP = 4;      // rows of its 8-channel chunk a thread keeps in registers between the two passes

template <typename T>
__global__ __launch_bounds__(256) void gn_small_kernel(const T *__restrict__ x1, int C1, const T *__restrict__ x2,
                                                       int C2, int HW, int groups, float eps,
                                                       const float *__restrict__ gamma, const float *__restrict__ beta,
                                                       int silu, T *__restrict__ out, const GnSlabSrc sl) {
    using v8 = typename TT<T>::v8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int C = C1 + C2, Cb = C >> 3, CHb = Cb >> 3, RP = 256 / CHb;   // channels / 16-B chunks of this block
    float *red = reinterpret_cast<float *>(smem);                 // [RP][Cb][2]
    float *ss = red + (size_t)RP * Cb * 2;                        // [2][Cb]
    const int sub = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int chunk = tid % CHb, prow = tid / CHb;
    const int cl = chunk * 8, c0 = sub * Cb + cl;                 // local / global first channel of this thread
    const bool from_slabs = sl.part != nullptr && c0 < C1;        // x1 arrives as split-K partial sums
    if (sl.part) x1 = (const T *)sl.raw_out;
    const T *src = c0 < C1 ? x1 + (size_t)b * HW * C1 + c0 : x2 + (size_t)b * HW * C2 + (c0 - C1);
    const int Cs = c0 < C1 ? C1 : C2;
    float s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
    v8 keep[GN_SMALL_KEEP];
    if (prow < RP) {
        f32x4 add0 = {0.f, 0.f, 0.f, 0.f}, add1 = {0.f, 0.f, 0.f, 0.f};
        if (from_slabs) {
            if (sl.bias) {
                add0 = *reinterpret_cast<const f32x4 *>(sl.bias + c0);
                add1 = *reinterpret_cast<const f32x4 *>(sl.bias + c0 + 4);
            }
        }
        auto load_row = [&](int p) __attribute__((always_inline)) {
            v8 v;
            if (from_slabs) {
                // same operation order as splitk_reduce_kernel: slabs in z order, + bias, + temb, + resid, round
                const size_t m = (size_t)b * HW + p, slab = (size_t)gridDim.y * HW * C1;
                const float *pp = sl.part + m * C1 + c0;
                f32x4 a0 = *reinterpret_cast<const f32x4 *>(pp), a1 = *reinterpret_cast<const f32x4 *>(pp + 4);
                for (int z0 = 1; z0 < sl.splitk; z0 += 4) {        // four slabs in flight; sums stay in z order
                    f32x4 u0[4], u1[4];
#pragma unroll
                    for (int z = 0; z < 4; ++z)
                        if (z0 + z < sl.splitk) {
                            u0[z] = *reinterpret_cast<const f32x4 *>(pp + (size_t)(z0 + z) * slab);
                            u1[z] = *reinterpret_cast<const f32x4 *>(pp + (size_t)(z0 + z) * slab + 4);
                        }
#pragma unroll
                    for (int z = 0; z < 4; ++z)
                        if (z0 + z < sl.splitk) {
                            a0 += u0[z];
                            a1 += u1[z];
                        }
                }
                if (sl.bias) {
                    a0 += add0;
                    a1 += add1;
                }
                if (sl.temb) {
                    const float *tp = sl.temb + (size_t)b * sl.temb_bstride + sl.temb_off + c0;
                    a0 += *reinterpret_cast<const f32x4 *>(tp);
                    a1 += *reinterpret_cast<const f32x4 *>(tp + 4);
                }
                if (sl.resid) {
                    const v8 rv = *reinterpret_cast<const v8 *>((const T *)sl.resid + m * C1 + c0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        a0[e] += (float)rv[e];
                        a1[e] += (float)rv[4 + e];
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = (T)a0[e];
                    v[4 + e] = (T)a1[e];
                }
                *reinterpret_cast<v8 *>((T *)sl.raw_out + m * C1 + c0) = v;     // re-read below by this same thread
            } else {
                v = *reinterpret_cast<const v8 *>(src + (size_t)p * Cs);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float f = (float)v[e];
                s1[e] += f;
                s2[e] = fmaf(f, f, s2[e]);
            }
            return v;
        };
        // a thread owns at most HW / RP rows (<= 4 for C <= 1024): they stay in registers for the second pass
#pragma unroll
        for (int k = 0; k < GN_SMALL_KEEP; ++k) {
            const int p = prow + k * RP;
            if (p < HW) keep[k] = load_row(p);
        }
        for (int p = prow + GN_SMALL_KEEP * RP; p < HW; p += RP) (void)load_row(p);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            red[((size_t)prow * Cb + cl + e) * 2 + 0] = s1[e];
            red[((size_t)prow * Cb + cl + e) * 2 + 1] = s2[e];
        }
    }
    __syncthreads();
    for (int i = tid; i < 2 * Cb; i += 256) {                      // per-channel totals into row 0
        float t = red[i];
        for (int r = 1; r < RP; ++r) t += red[(size_t)r * Cb * 2 + i];
        red[i] = t;
    }
    __syncthreads();
    const int Cg = C / groups;
    for (int c = tid; c < Cb; c += 256) {
        const int g0 = (c / Cg) * Cg;
        double sm = 0, q = 0;
        for (int k = 0; k < Cg; ++k) {
            sm += red[(g0 + k) * 2];
            q += red[(g0 + k) * 2 + 1];
        }
        const double n = (double)Cg * HW, mean = sm / n;
        double var = q / n - mean * mean;
        var = var > 0 ? var : 0;
        const float sc = (float)(1.0 / sqrt(var + (double)eps)) * gamma[sub * Cb + c];
        ss[c] = sc;
        ss[Cb + c] = beta[sub * Cb + c] - (float)mean * sc;
    }
    __syncthreads();
    if (prow < RP) {
        auto apply_row = [&](int p, const v8 &v) __attribute__((always_inline)) {
            v8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float f = fmaf((float)v[e], ss[cl + e], ss[Cb + cl + e]);
                if (silu) f = f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * f));
                o[e] = (T)f;
            }
            store_wt(reinterpret_cast<v8 *>(out + ((size_t)b * HW + p) * C + c0), o);
        };
#pragma unroll
        for (int k = 0; k < GN_SMALL_KEEP; ++k) {
            const int p = prow + k * RP;
            if (p < HW) apply_row(p, keep[k]);
        }
        for (int p = prow + GN_SMALL_KEEP * RP; p < HW; p += RP)
            apply_row(p, *reinterpret_cast<const v8 *>(src + (size_t)p * Cs));
    }
}

}  // namespace

int launch_gn_small(int dtype, const void *x1, int C1, const void *x2, int C2, int B, int HW, int groups, float eps,
                    const float *gamma, const float *beta, int silu, void *out, hipStream_t st, const GnSlabSrc *slab) {
    const GnSlabSrc sl = slab ? *slab : GnSlabSrc{};
    const int C = C1 + C2;
    // a block owns C/8 channels = 4 groups; chunks of 8 channels must not straddle x1 | x2
    if (groups != 32 || C % 64 || C1 % 8 || C > 2048) {
        set_error("gn_small: unsupported channels %d+%d (groups %d)", C1, C2, groups);
        return BNDM_E_ARG;
    }
    const int Cb = C / 8, CHb = Cb / 8, RP = 256 / CHb;
    const size_t smem = (size_t)RP * Cb * 2 * 4 + (size_t)2 * Cb * 4;
    const dim3 grid(8, B);
    if (dtype == BNDM_DTYPE_F16)
        hipLaunchKernelGGL(gn_small_kernel<_Float16>, grid, dim3(256), smem, st, (const _Float16 *)x1, C1,
                           (const _Float16 *)x2, C2, HW, groups, eps, gamma, beta, silu, (_Float16 *)out, sl);
    else
        hipLaunchKernelGGL(gn_small_kernel<__bf16>, grid, dim3(256), smem, st, (const __bf16 *)x1, C1,
                           (const __bf16 *)x2, C2, HW, groups, eps, gamma, beta, silu, (__bf16 *)out, sl);
    return launch_status("gn_small");
}

int launch_gn_finalize2(const float *p1, int nslab1, int C1, const float *p2, int nslab2, int C2, int B, int HW,
                        int groups, float eps, const float *gamma, const float *beta, float *scale_shift,
                        hipStream_t st) {
    if (C1 + C2 > 2048 || groups != 32 || (C1 + C2) % 64) {
        set_error("gn_finalize2: C=%d groups=%d unsupported", C1 + C2, groups);
        return BNDM_E_ARG;
    }
    hipLaunchKernelGGL(gn_finalize2_kernel, dim3(8, B), dim3(128), 0, st, p1, nslab1, C1, p2, nslab2, C2, HW, groups,
                       eps, gamma, beta, scale_shift);
    return launch_status("gn_finalize2");
}

}  // namespace bndm
