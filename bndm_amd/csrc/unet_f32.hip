// fp32-compute mode of the UNet2DModel engine (BNDM_DTYPE_F32): the reference samples in fp32 with no autocast
// (iadb_bn.py:304-344), and SURVEY.md 8d asks for "an fp32-compute HIP mode at B <= 4 that must hold rel-L2 <= 1e-4 per
// forward".  This is that mode: the same network (diffusers.UNet2DModel as built at iadb_bn.py:205-282) evaluated layer
// by layer on fp32 NCHW tensors with plain fp32 FMA kernels -- no MFMA, no 16-bit storage, no fusion.  It shares nothing
// with the 16-bit engine but the parameter registry, so it doubles as an independent check of that engine's packing and
// fusion logic (tests/test_gpu_f32.py: oracle vs fp32 mode <= 1e-4, fp32 mode vs f16 engine <= 2e-3).  Speed is not
// a goal (about 100x slower than the 16-bit engine); correctness and readability are.
#include "unet_f32.hpp"

#include <cmath>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

namespace bndm {
namespace {

constexpr int GROUPS = 32;
constexpr float GN_EPS = 1e-5f;
constexpr int CO_T = 16;          // output channels per thread of the direct convolution

__device__ __forceinline__ float silu32(float v) { return v / (1.0f + expf(-v)); }

// Direct convolution, KS x KS taps (3 or 1), stride 1 or 2, zero padding KS/2, optional nearest-2x upsampling of the
// input (the taps then address the upsampled grid).  One thread = one output pixel x CO_T output channels; the weights
// of a (co block, ci) pair are block-uniform, so they come through the scalar cache.
//   out[b][co][y][x] = bias[co] + addbc[b][co] + resid[b][co][y][x] + sum_ci,t w[co][ci][t] * in[b][ci][..]
template <int KS>
__global__ __launch_bounds__(256) void conv_f32_kernel(const float *__restrict__ in, const float *__restrict__ w,
                                                       const float *__restrict__ bias, const float *__restrict__ addbc,
                                                       const float *__restrict__ resid, float *__restrict__ out, int Cin,
                                                       int Hi, int Wi, int Cout, int Ho, int Wo, int stride, int up) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    const int co0 = blockIdx.y * CO_T, b = blockIdx.z;
    const bool live = p < Ho * Wo;
    const int oy = live ? p / Wo : 0, ox = live ? p - oy * Wo : 0;
    const int He = up ? Hi * 2 : Hi, We = up ? Wi * 2 : Wi;     // extent of the (upsampled) input grid
    constexpr int T = KS * KS, PAD = KS / 2;
    int off[T];
    bool ok[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int iy = oy * stride + t / KS - PAD, ix = ox * stride + t % KS - PAD;
        ok[t] = live && (unsigned)iy < (unsigned)He && (unsigned)ix < (unsigned)We;
        off[t] = ok[t] ? (up ? (iy >> 1) * Wi + (ix >> 1) : iy * Wi + ix) : 0;
    }
    float acc[CO_T];
#pragma unroll
    for (int k = 0; k < CO_T; ++k) acc[k] = 0.f;
    const float *inb = in + (size_t)b * Cin * Hi * Wi;
    for (int ci = 0; ci < Cin; ++ci) {
        float v[T];
#pragma unroll
        for (int t = 0; t < T; ++t) v[t] = ok[t] ? inb[(size_t)ci * Hi * Wi + off[t]] : 0.f;
#pragma unroll
        for (int k = 0; k < CO_T; ++k) {
            const int co = co0 + k < Cout ? co0 + k : Cout - 1;          // (clamped: the store below is guarded)
            const float *wk = w + ((size_t)co * Cin + ci) * T;
#pragma unroll
            for (int t = 0; t < T; ++t) acc[k] = fmaf(wk[t], v[t], acc[k]);
        }
    }
    if (!live) return;
#pragma unroll
    for (int k = 0; k < CO_T; ++k) {
        const int co = co0 + k;
        if (co >= Cout) break;
        float r = acc[k] + (bias ? bias[co] : 0.f);
        if (addbc) r += addbc[(size_t)b * Cout + co];
        const size_t o = ((size_t)b * Cout + co) * Ho * Wo + p;
        if (resid) r += resid[o];
        out[o] = r;
    }
}

// GroupNorm(32, eps) over [B][C][HW] (+ optional SiLU); one block per (group, sample), statistics in double
__global__ __launch_bounds__(256) void gn_f32_kernel(const float *__restrict__ x, const float *__restrict__ gamma,
                                                     const float *__restrict__ beta, float *__restrict__ y, int C, int HW,
                                                     float eps, int silu) {
    __shared__ double red[2][256];
    const int g = blockIdx.x, b = blockIdx.y, cpg = C / GROUPS;
    const size_t base = ((size_t)b * C + (size_t)g * cpg) * HW;
    const int n = cpg * HW;
    double s = 0, ss = 0;
    for (int i = threadIdx.x; i < n; i += 256) {
        const double v = x[base + i];
        s += v;
        ss += v * v;
    }
    red[0][threadIdx.x] = s;
    red[1][threadIdx.x] = ss;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if (threadIdx.x < k) {
            red[0][threadIdx.x] += red[0][threadIdx.x + k];
            red[1][threadIdx.x] += red[1][threadIdx.x + k];
        }
        __syncthreads();
    }
    const double mean = red[0][0] / n;
    const double var = fmax(red[1][0] / n - mean * mean, 0.0);
    const float m = (float)mean, rstd = (float)(1.0 / sqrt(var + (double)eps));
    for (int i = threadIdx.x; i < n; i += 256) {
        const int c = g * cpg + i / HW;
        float v = (x[base + i] - m) * rstd * gamma[c] + beta[c];
        y[base + i] = silu ? silu32(v) : v;
    }
}

// out[b][o] = bias[o] + sum_i w[o][i] * f(x[b][i]),  f = SiLU when silu_in
__global__ __launch_bounds__(256) void linear_f32_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                         const float *__restrict__ bias, float *__restrict__ out, int I,
                                                         int O, int silu_in) {
    const int o = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (o >= O) return;
    float acc = 0.f;
    for (int i = 0; i < I; ++i) {
        float v = x[(size_t)b * I + i];
        if (silu_in) v = silu32(v);
        acc = fmaf(w[(size_t)o * I + i], v, acc);
    }
    out[(size_t)b * O + o] = acc + bias[o];
}

// Timesteps(dim, flip_sin_to_cos=True, freq_shift=0): [cos | sin], freqs exp(-ln(1e4) k / half)
__global__ void timestep_f32_kernel(const float *__restrict__ t, float *__restrict__ out, int dim) {
    const int k = threadIdx.x, b = blockIdx.x, half = dim / 2;
    if (k >= half) return;
    const float f = expf(-logf(10000.0f) * (float)k / (float)half);
    const float ang = t[b] * f;
    out[(size_t)b * dim + k] = cosf(ang);
    out[(size_t)b * dim + half + k] = sinf(ang);
}

// softmax(q k^T / sqrt(8)) v for 8-wide heads; q, k, v, out are [B][C][T]; one thread per (sample, head, query)
__global__ __launch_bounds__(64) void attn_f32_kernel(const float *__restrict__ q, const float *__restrict__ k,
                                                      const float *__restrict__ v, float *__restrict__ out, int C, int T) {
    const int t = blockIdx.x * 64 + threadIdx.x, hd = blockIdx.y, b = blockIdx.z;
    if (t >= T) return;
    const size_t base = ((size_t)b * C + (size_t)hd * 8) * T;
    float qv[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) qv[d] = q[base + (size_t)d * T + t];
    const float scale = 0.35355339059327373f;           // 8^-0.5
    float mx = -INFINITY;
    for (int s = 0; s < T; ++s) {
        float dot = 0.f;
#pragma unroll
        for (int d = 0; d < 8; ++d) dot = fmaf(qv[d], k[base + (size_t)d * T + s], dot);
        mx = fmaxf(mx, dot * scale);
    }
    float den = 0.f, o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < T; ++s) {
        float dot = 0.f;
#pragma unroll
        for (int d = 0; d < 8; ++d) dot = fmaf(qv[d], k[base + (size_t)d * T + s], dot);
        const float p = expf(dot * scale - mx);
        den += p;
#pragma unroll
        for (int d = 0; d < 8; ++d) o[d] = fmaf(p, v[base + (size_t)d * T + s], o[d]);
    }
#pragma unroll
    for (int d = 0; d < 8; ++d) out[base + (size_t)d * T + t] = o[d] / den;
}

// dst[b][c_off + c][hw] = src[b][c][hw]
__global__ __launch_bounds__(256) void put_channels_f32_kernel(const float *__restrict__ src, float *__restrict__ dst, int C,
                                                               int Ctot, int c_off, int HW, size_t total) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const size_t per = (size_t)C * HW;
    const size_t b = i / per, r = i - b * per;
    dst[(b * Ctot + c_off) * HW + r] = src[i];
}

struct Tensor {
    float *p = nullptr;
    int C = 0, H = 0, W = 0;
};

}  // namespace

struct F32Model {
    bndm_unet_config cfg{};
    std::unordered_map<std::string, float *> par;       // device copies of the state dict, PyTorch layouts
    std::vector<void *> owned;
    char *arena = nullptr;
    size_t arena_bytes = 0;

    // per-forward state
    size_t off = 0;
    bool dry = false;
    int B = 0;
    hipStream_t st = nullptr;
    int err = 0;

    float *alloc(size_t n) {
        const size_t bytes = (n * 4 + 255) & ~(size_t)255;
        float *p = reinterpret_cast<float *>(arena + off);
        off += bytes;
        if (!dry && off > arena_bytes && !err) {
            set_error("fp32 mode: workspace overflow (%zu > %zu bytes)", off, arena_bytes);
            err = BNDM_E_STATE;
        }
        return p;
    }
    Tensor tensor(int C, int H, int W) { return Tensor{alloc((size_t)B * C * H * W), C, H, W}; }
    const float *P(const std::string &n) const { return par.at(n); }
    bool has(const std::string &n) const { return par.count(n) != 0; }
    bool go() const { return !dry && !err; }

    Tensor conv(const Tensor &x, const std::string &name, int Cout, int ks, int stride, int up, const float *addbc,
                const Tensor *resid) {
        const int He = up ? x.H * 2 : x.H, We = up ? x.W * 2 : x.W;
        const int Ho = stride == 2 ? He / 2 : He, Wo = stride == 2 ? We / 2 : We;
        Tensor y = tensor(Cout, Ho, Wo);
        if (go()) {
            dim3 grid((Ho * Wo + 255) / 256, (Cout + CO_T - 1) / CO_T, B);
            if (ks == 3)
                hipLaunchKernelGGL(conv_f32_kernel<3>, grid, dim3(256), 0, st, x.p, P(name + ".weight"), P(name + ".bias"),
                                   addbc, resid ? resid->p : nullptr, y.p, x.C, x.H, x.W, Cout, Ho, Wo, stride, up);
            else
                hipLaunchKernelGGL(conv_f32_kernel<1>, grid, dim3(256), 0, st, x.p, P(name + ".weight"), P(name + ".bias"),
                                   addbc, resid ? resid->p : nullptr, y.p, x.C, x.H, x.W, Cout, Ho, Wo, stride, up);
        }
        return y;
    }
    Tensor gn(const Tensor &x, const std::string &name, bool silu) {
        Tensor y = tensor(x.C, x.H, x.W);
        if (go())
            hipLaunchKernelGGL(gn_f32_kernel, dim3(GROUPS, B), dim3(256), 0, st, x.p, P(name + ".weight"), P(name + ".bias"),
                               y.p, x.C, x.H * x.W, GN_EPS, silu ? 1 : 0);
        return y;
    }
    float *linear(const float *x, const std::string &name, int I, int O, bool silu_in) {
        float *y = alloc((size_t)B * O);
        if (go())
            hipLaunchKernelGGL(linear_f32_kernel, dim3((O + 255) / 256, B), dim3(256), 0, st, x, P(name + ".weight"),
                               P(name + ".bias"), y, I, O, silu_in ? 1 : 0);
        return y;
    }
    Tensor cat(const Tensor &a, const Tensor &b) {
        Tensor y = tensor(a.C + b.C, a.H, a.W);
        if (go()) {
            const int HW = a.H * a.W;
            const size_t ta = (size_t)B * a.C * HW, tb = (size_t)B * b.C * HW;
            hipLaunchKernelGGL(put_channels_f32_kernel, dim3((unsigned)((ta + 255) / 256)), dim3(256), 0, st, a.p, y.p, a.C,
                               y.C, 0, HW, ta);
            hipLaunchKernelGGL(put_channels_f32_kernel, dim3((unsigned)((tb + 255) / 256)), dim3(256), 0, st, b.p, y.p, b.C,
                               y.C, a.C, HW, tb);
        }
        return y;
    }
    // ResnetBlock2D (oracle/unet_oracle.py::_resnet)
    Tensor resnet(const Tensor &x, const float *emb, const std::string &name, int Cout, int D) {
        Tensor h = gn(x, name + ".norm1", true);
        float *tp = linear(emb, name + ".time_emb_proj", D, Cout, true);
        h = conv(h, name + ".conv1", Cout, 3, 1, 0, tp, nullptr);
        h = gn(h, name + ".norm2", true);
        Tensor sc = x;
        if (has(name + ".conv_shortcut.weight")) sc = conv(x, name + ".conv_shortcut", Cout, 1, 1, 0, nullptr, nullptr);
        return conv(h, name + ".conv2", Cout, 3, 1, 0, nullptr, &sc);
    }
    // Attention block (oracle/unet_oracle.py::_attn): linear layers over channels are 1x1 convolutions on [B][C][T]
    Tensor attn(const Tensor &x, const std::string &name) {
        const int C = x.C, T = x.H * x.W;
        Tensor h = gn(x, name + ".group_norm", false);
        Tensor q = conv(h, name + ".to_q", C, 1, 1, 0, nullptr, nullptr);
        Tensor k = conv(h, name + ".to_k", C, 1, 1, 0, nullptr, nullptr);
        Tensor v = conv(h, name + ".to_v", C, 1, 1, 0, nullptr, nullptr);
        Tensor o = tensor(C, x.H, x.W);
        if (go())
            hipLaunchKernelGGL(attn_f32_kernel, dim3((T + 63) / 64, C / 8, B), dim3(64), 0, st, q.p, k.p, v.p, o.p, C, T);
        return conv(o, name + ".to_out.0", C, 1, 1, 0, nullptr, &x);
    }

    static std::string S(const char *fmt, int a, int b = 0) {
        char buf[96];
        snprintf(buf, sizeof(buf), fmt, a, b);
        return buf;
    }

    // UNet2DModel.forward (oracle/unet_oracle.py::forward); `sample` is [B][Cin][R][R] (or x and extra concatenated)
    void forward(const float *sample, const float *extra, const float *timesteps, float *out) {
        const bndm_unet_config &c = cfg;
        const int *boc = c.block_out_channels;
        const int n = c.num_levels, R = c.resolution, D = boc[0] * 4;
        off = 0;
        float *e0 = alloc((size_t)B * boc[0]);
        if (go()) hipLaunchKernelGGL(timestep_f32_kernel, dim3(B), dim3(boc[0] / 2), 0, st, timesteps, e0, boc[0]);
        float *e1 = linear(e0, "time_embedding.linear_1", boc[0], D, false);
        float *emb = linear(e1, "time_embedding.linear_2", D, D, true);

        Tensor in{const_cast<float *>(sample), c.in_channels, R, R};
        if (extra) {                                   // conditional sampler: cat([x, x_c], 1) (iadb_bn.py:406)
            Tensor a{const_cast<float *>(sample), c.in_channels / 2, R, R}, bx{const_cast<float *>(extra), c.in_channels / 2, R, R};
            in = cat(a, bx);
        }
        Tensor h = conv(in, "conv_in", boc[0], 3, 1, 0, nullptr, nullptr);
        std::vector<Tensor> skips{h};
        for (int i = 0; i < n; ++i) {
            for (int j = 0; j < c.layers_per_block; ++j) {
                h = resnet(h, emb, S("down_blocks.%d.resnets.%d", i, j), boc[i], D);
                if (c.down_attn[i]) h = attn(h, S("down_blocks.%d.attentions.%d", i, j));
                skips.push_back(h);
            }
            if (i != n - 1) {
                h = conv(h, S("down_blocks.%d.downsamplers.0.conv", i), boc[i], 3, 2, 0, nullptr, nullptr);
                skips.push_back(h);
            }
        }
        h = resnet(h, emb, "mid_block.resnets.0", boc[n - 1], D);
        h = attn(h, "mid_block.attentions.0");
        h = resnet(h, emb, "mid_block.resnets.1", boc[n - 1], D);
        for (int i = 0; i < n; ++i) {
            const int oc = boc[n - 1 - i];
            for (int j = 0; j < c.layers_per_block + 1; ++j) {
                h = cat(h, skips.back());
                skips.pop_back();
                h = resnet(h, emb, S("up_blocks.%d.resnets.%d", i, j), oc, D);
                if (c.up_attn[i]) h = attn(h, S("up_blocks.%d.attentions.%d", i, j));
            }
            if (i != n - 1) h = conv(h, S("up_blocks.%d.upsamplers.0.conv", i), oc, 3, 1, 1, nullptr, nullptr);
        }
        h = gn(h, "conv_norm_out", true);
        // conv_out straight into the caller's tensor
        const size_t keep = off;
        Tensor y = conv(h, "conv_out", c.out_channels, 3, 1, 0, nullptr, nullptr);
        if (go()) (void)hipMemcpyAsync(out, y.p, (size_t)B * c.out_channels * R * R * 4, hipMemcpyDeviceToDevice, st);
        (void)keep;
    }
};

int f32_model_create(const bndm_unet_config &cfg, const std::vector<std::string> &names,
                     const std::vector<std::vector<float>> &values, F32Model **out) {
    F32Model *m = new (std::nothrow) F32Model();
    if (!m) return BNDM_E_NOMEM;
    m->cfg = cfg;
    for (size_t i = 0; i < names.size(); ++i) {
        void *p = nullptr;
        if (hipMalloc(&p, values[i].size() * 4 + 16) != hipSuccess ||
            hipMemcpy(p, values[i].data(), values[i].size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
            set_error("fp32 mode: upload of '%s' failed", names[i].c_str());
            f32_model_destroy(m);
            return BNDM_E_NOMEM;
        }
        m->owned.push_back(p);
        m->par[names[i]] = (float *)p;
    }
    // size the workspace with a dry run at max_batch
    m->dry = true;
    m->B = cfg.max_batch;
    m->forward(nullptr, cfg.in_channels % 2 == 0 ? (const float *)16 : nullptr, nullptr, nullptr);
    m->arena_bytes = m->off + 4096;
    m->dry = false;
    if (hipMalloc((void **)&m->arena, m->arena_bytes) != hipSuccess) {
        set_error("fp32 mode: %zu-byte workspace allocation failed", m->arena_bytes);
        f32_model_destroy(m);
        return BNDM_E_NOMEM;
    }
    *out = m;
    return 0;
}

void f32_model_destroy(F32Model *m) {
    if (!m) return;
    for (void *p : m->owned) (void)hipFree(p);
    if (m->arena) (void)hipFree(m->arena);
    delete m;
}

int f32_model_forward(F32Model *m, const float *sample, const float *extra, const float *timesteps, float *out, int B,
                      hipStream_t st) {
    m->B = B;
    m->st = st;
    m->err = 0;
    m->forward(sample, extra, timesteps, out);
    if (m->err) return m->err;
    return launch_status("fp32 forward");
}

}  // namespace bndm
