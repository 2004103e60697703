// UNet2DModel engine behind the C ABI (include/bndm_hip.h, bndm_unet_*).
//
// Mirrors what the reference obtains from diffusers.UNet2DModel(...) (iadb_bn.py:205-282,
// utils.py:7-84, ddim_diffusers.py:377-453, latent_iadb_bn_diffusers.py:337-372):
//   conv_in -> [ResnetBlock2D x2 (+Attention) -> Downsample2D] per level -> mid (Res, Attn, Res)
//   -> [ResnetBlock2D x3 on cat(h, skip) (+Attention) -> Upsample2D] per level -> GN -> SiLU -> conv_out
// The handle owns packed 16-bit weights ([Cout][K] with K ordered segment -> tap -> channel), fp32
// norm/bias tables and activation workspaces sized for max_batch.  forward() is a fixed list of
// kernel launches built once in finalize(); nothing is allocated or synchronised per call.
#include "unet_kernels.hpp"
#include "unet_f32.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <functional>
#include <list>
#include <map>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

using namespace bndm;

namespace {

constexpr int GROUPS = 32;
constexpr float GN_EPS = 1e-5f;

uint16_t to_f16_bits(float f) {
    _Float16 h = (_Float16)f;
    uint16_t u;
    memcpy(&u, &h, 2);
    return u;
}
uint16_t to_bf16_bits(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // NaN
    u += 0x7fffu + ((u >> 16) & 1u);                                             // round to nearest even
    return (uint16_t)(u >> 16);
}

struct ParamSpec {
    std::string name;
    std::vector<int> shape;
    int64_t numel;
};

struct Buf {
    size_t bytes = 0;
    void *ptr = nullptr;
};

// tile / split-K choice of the generic conv path (shared by the conv launch and by the consumer of deferred slabs)
struct ConvPlan {
    int tile, splitk;
};
static ConvPlan plan_conv(int M, int Cout, int ksteps, size_t splitk_bytes) {
    int tile = TILE_256x128;
    int nblk = ceil_div(M, 256) * ceil_div(Cout, 128);
    if (nblk < 192) {
        tile = TILE_128x128;
        nblk = ceil_div(M, 128) * ceil_div(Cout, 128);
    }
    int splitk = 1;
    constexpr int sk_target = 256, sk_minsteps = 8;      // workgroups aimed at, K-steps a slice keeps at least
    if (nblk < 192 && ksteps >= 8) {
        splitk = std::min(std::min(ceil_div(sk_target, nblk), ksteps / sk_minsteps), 32);
        if (splitk >= 2) {
            const int per = ceil_div(ksteps, splitk);
            splitk = ceil_div(ksteps, per);          // no empty slices
        }
        if (splitk < 2) splitk = 1;
    }
    if ((size_t)splitk * M * Cout * 4 > splitk_bytes) splitk = 1;
    return ConvPlan{tile, splitk};
}

// What the conv_s launch that produces a <= 8x8 tensor writes besides the raw tensor: one GroupNorm(+SiLU) normalised copy
// per consuming GroupNorm.  Consumers are built after their producer, so they append to this record and the producer's
// launch closure reads it at run time.
struct TailOut {
    struct Req {
        int slot;
        const float *gamma, *beta;
        int gs, silu;
    };
    std::vector<Req> reqs;
    bool raw = true;     // the raw 16-bit tensor is written (false: every consumer reads a normalised copy)
};

struct Act {       // NHWC 16-bit activation living in buffer slot `slot`
    int slot;
    int C, H, W;
    std::shared_ptr<TailOut> to;   // produced by conv_s (unet_tail.hip)
};

struct RunCtx {
    int B;
    hipStream_t st;
    const float *sample;
    const float *extra;
    const float *timesteps;
    float *out;
    // time-embedding projections precomputed for the whole schedule (sampler loops): row of this step,
    // shared by every sample (batch stride 0); nullptr -> computed per forward from `timesteps`
    const float *tp_row = nullptr;
    // profiling
    bool prof = false;
    std::vector<hipEvent_t> *ev = nullptr;
};

enum OpClass { OPC_CONV = 0, OPC_OTHER = 1 };

struct Op {
    int cls;
    double flops_per_sample;   // algorithmic 2*MAC per batch element (convs only)
    std::function<int(RunCtx &)> run;
    std::string name;
    bool dominant = false;          // conv_t32 with 256-pixel tiles
    double bytes_per_sample = 0;    // algorithmic HBM bytes per batch element (activations)
    double bytes_fixed = 0;         // weights
    std::string kernel;             // kernel family (and tile variant) this op launches, e.g. "conv_t32<TH=16>"
};

}  // namespace

struct bndm_unet {
    bndm_unet_config cfg{};
    int kind = 0;                      // 0: UNet2DModel, 1: AutoencoderKL decoder (cfg.resolution = latent H = W)
    F32Model *f32 = nullptr;           // BNDM_DTYPE_F32: the whole forward runs in csrc/unet_f32.hip
    int temb_dim = 0;
    std::vector<ParamSpec> params;
    std::unordered_map<std::string, int> pindex;
    std::vector<std::vector<float>> host;
    std::vector<char> loaded;
    bool finalized = false;

    std::vector<Buf> bufs;             // activation / scratch slots (sized for max_batch)
    std::vector<void *> weights;       // packed parameter allocations
    std::vector<Op> ops;
    int ntemb = 0;                     // total time_emb_proj columns
    void *zeros = nullptr;
    // per-schedule time-embedding table (K10): [cap][ntemb] fp32, [cap] fp32 t, [cap][temb_dim] 16-bit
    float *tp_table = nullptr, *t_steps = nullptr;
    void *act_steps = nullptr;
    int tp_cap = 0;
    float *t_pinned = nullptr;           // host staging of the step times (pinned: the upload is truly asynchronous)
    hipEvent_t t_uploaded = nullptr;     // recorded after that upload; the next call waits for it before rewriting
    std::vector<void *> retired;         // outgrown tables, released with the handle (earlier launches may still read them)
    std::function<int(int, const float *, void *, float *, hipStream_t)> temb_table_fn;

    // fixed scratch slots
    int s_y = -1, s_h1 = -1, s_y2 = -1, s_part = -1, s_ss = -1, s_qkv = -1, s_att = -1, s_splitk = -1;
    int s_actemb = -1, s_tp = -1, s_d = -1, s_t = -1;

    int dtype() const { return cfg.dtype; }
    int new_slot(size_t bytes) {
        bufs.push_back(Buf{bytes, nullptr});
        return (int)bufs.size() - 1;
    }
    void grow(int slot, size_t bytes) {
        if (bufs[slot].bytes < bytes) bufs[slot].bytes = bytes;
    }
    void *P(int slot) const { return bufs[slot].ptr; }
    const std::vector<float> &hp(const std::string &n) const { return host[pindex.at(n)]; }
};

namespace {

// ------------------------------------------------------------------------------------------------
// parameter registry (diffusers state-dict naming)
// ------------------------------------------------------------------------------------------------
struct SpecBuilder {
    bndm_unet *h;
    void add(const std::string &n, std::vector<int> shape) {
        int64_t ne = 1;
        for (int d : shape) ne *= d;
        h->pindex[n] = (int)h->params.size();
        h->params.push_back(ParamSpec{n, shape, ne});
    }
    void conv(const std::string &n, int ci, int co, int k) {
        add(n + ".weight", {co, ci, k, k});
        add(n + ".bias", {co});
    }
    void lin(const std::string &n, int ci, int co) {
        add(n + ".weight", {co, ci});
        add(n + ".bias", {co});
    }
    void norm(const std::string &n, int c) {
        add(n + ".weight", {c});
        add(n + ".bias", {c});
    }
    void resnet(const std::string &n, int ci, int co) {
        norm(n + ".norm1", ci);
        conv(n + ".conv1", ci, co, 3);
        lin(n + ".time_emb_proj", h->temb_dim, co);
        norm(n + ".norm2", co);
        conv(n + ".conv2", co, co, 3);
        if (ci != co) conv(n + ".conv_shortcut", ci, co, 1);
    }
    void attn(const std::string &n, int c) {
        norm(n + ".group_norm", c);
        lin(n + ".to_q", c, c);
        lin(n + ".to_k", c, c);
        lin(n + ".to_v", c, c);
        lin(n + ".to_out.0", c, c);
    }
};

std::string S(const char *fmt, ...) {
    char buf[160];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    return buf;
}

// AutoencoderKL decoder (+ post_quant_conv); cfg.block_out_channels holds the ENCODER order (128, 256, 512, 512)
void build_specs_vae(bndm_unet *h) {
    const bndm_unet_config &c = h->cfg;
    const int n = c.num_levels, L = c.in_channels;
    SpecBuilder sb{h};
    auto resnet = [&](const std::string &nm, int ci, int co) {
        sb.norm(nm + ".norm1", ci);
        sb.conv(nm + ".conv1", ci, co, 3);
        sb.norm(nm + ".norm2", co);
        sb.conv(nm + ".conv2", co, co, 3);
        if (ci != co) sb.conv(nm + ".conv_shortcut", ci, co, 1);
    };
    const int top = c.block_out_channels[n - 1];
    sb.conv("post_quant_conv", L, L, 1);
    sb.conv("decoder.conv_in", L, top, 3);
    resnet("decoder.mid_block.resnets.0", top, top);
    sb.attn("decoder.mid_block.attentions.0", top);
    resnet("decoder.mid_block.resnets.1", top, top);
    int prev = top;
    for (int i = 0; i < n; ++i) {
        const int oc = c.block_out_channels[n - 1 - i];
        for (int j = 0; j < c.layers_per_block + 1; ++j)
            resnet(S("decoder.up_blocks.%d.resnets.%d", i, j), j == 0 ? prev : oc, oc);
        if (i != n - 1) sb.conv(S("decoder.up_blocks.%d.upsamplers.0.conv", i), oc, oc, 3);
        prev = oc;
    }
    sb.norm("decoder.conv_norm_out", c.block_out_channels[0]);
    sb.conv("decoder.conv_out", c.block_out_channels[0], c.out_channels, 3);
}

void build_specs(bndm_unet *h) {
    const bndm_unet_config &c = h->cfg;
    const int *boc = c.block_out_channels;
    const int n = c.num_levels;
    SpecBuilder sb{h};
    sb.conv("conv_in", c.in_channels, boc[0], 3);
    sb.lin("time_embedding.linear_1", boc[0], h->temb_dim);
    sb.lin("time_embedding.linear_2", h->temb_dim, h->temb_dim);
    int out_c = boc[0];
    for (int i = 0; i < n; ++i) {
        const int in_c = out_c;
        out_c = boc[i];
        for (int j = 0; j < c.layers_per_block; ++j) {
            sb.resnet(S("down_blocks.%d.resnets.%d", i, j), j == 0 ? in_c : out_c, out_c);
            if (c.down_attn[i]) sb.attn(S("down_blocks.%d.attentions.%d", i, j), out_c);
        }
        if (i != n - 1) sb.conv(S("down_blocks.%d.downsamplers.0.conv", i), out_c, out_c, 3);
    }
    const int mid = boc[n - 1];
    sb.resnet("mid_block.resnets.0", mid, mid);
    sb.attn("mid_block.attentions.0", mid);
    sb.resnet("mid_block.resnets.1", mid, mid);
    out_c = boc[n - 1];
    for (int i = 0; i < n; ++i) {
        const int prev = out_c;
        out_c = boc[n - 1 - i];
        const int in_c = boc[n - 1 - std::min(i + 1, n - 1)];
        const int nl = c.layers_per_block + 1;
        for (int j = 0; j < nl; ++j) {
            const int skip = (j == nl - 1) ? in_c : out_c;
            const int rin = (j == 0) ? prev : out_c;
            sb.resnet(S("up_blocks.%d.resnets.%d", i, j), rin + skip, out_c);
            if (c.up_attn[i]) sb.attn(S("up_blocks.%d.attentions.%d", i, j), out_c);
        }
        if (i != n - 1) sb.conv(S("up_blocks.%d.upsamplers.0.conv", i), out_c, out_c, 3);
    }
    sb.norm("conv_norm_out", boc[0]);
    sb.conv("conv_out", boc[0], c.out_channels, 3);
}

// ------------------------------------------------------------------------------------------------
// device uploads
// ------------------------------------------------------------------------------------------------
int upload(bndm_unet *h, const void *src, size_t bytes, void **out) {
    void *p = nullptr;
    BNDM_CHECK_HIP(hipMalloc(&p, bytes ? bytes : 16));
    h->weights.push_back(p);
    if (bytes) BNDM_CHECK_HIP(hipMemcpy(p, src, bytes, hipMemcpyHostToDevice));
    *out = p;
    return 0;
}

int upload_f32(bndm_unet *h, const std::vector<float> &v, const float **out) {
    void *p;
    int rc = upload(h, v.data(), v.size() * 4, &p);
    *out = (const float *)p;
    return rc;
}

int upload_16(bndm_unet *h, const std::vector<float> &v, const void **out) {
    std::vector<uint16_t> q(v.size());
    if (h->dtype() == BNDM_DTYPE_F16)
        for (size_t i = 0; i < v.size(); ++i) q[i] = to_f16_bits(v[i]);
    else
        for (size_t i = 0; i < v.size(); ++i) q[i] = to_bf16_bits(v[i]);
    void *p;
    int rc = upload(h, q.data(), q.size() * 2, &p);
    *out = p;
    return rc;
}

// One K-segment of a packed conv weight: rows of `w` ([Cout][CinTot][k][k], PyTorch OIHW),
// input channels [c_begin, c_begin + C), taps = k*k.
struct WSeg {
    const std::vector<float> *w;
    int cin_total, c_begin, C, taps;
};

// Wp[co][koff + tap*C + c] = w[co][c_begin + c][tap]; rows padded with zeros to a multiple of `row_pad`
int pack_conv_weight(bndm_unet *h, const std::vector<WSeg> &segs, int Cout, int row_pad, const void **out,
                     int *Ktot_out) {
    int Ktot = 0;
    for (const WSeg &s : segs) Ktot += s.taps * s.C;
    const int rows = ceil_div(Cout, row_pad) * row_pad;
    std::vector<float> wp((size_t)rows * Ktot, 0.f);
    int koff = 0;
    for (const WSeg &s : segs) {
        for (int co = 0; co < Cout; ++co)
            for (int t = 0; t < s.taps; ++t)
                for (int c = 0; c < s.C; ++c)
                    wp[(size_t)co * Ktot + koff + t * s.C + c] =
                        (*s.w)[((size_t)co * s.cin_total + s.c_begin + c) * s.taps + t];
        koff += s.taps * s.C;
    }
    *Ktot_out = Ktot;
    if (row_pad == 128 && Ktot % 64 == 0) {
        // tile-contiguous, pre-swizzled layout of the 128-row igemm tiles (ConvArgs::wtiled)
        std::vector<float> wt(wp.size());
        const int ks = Ktot / 64;
        for (int nt = 0; nt < rows / 128; ++nt)
            for (int s = 0; s < ks; ++s)
                for (int r = 0; r < 128; ++r)
                    for (int j = 0; j < 8; ++j) {
                        const float *src = &wp[(size_t)(nt * 128 + r) * Ktot + s * 64 + (j ^ ((r >> 1) & 7)) * 8];
                        float *dst = &wt[(((size_t)nt * ks + s) * 128 + r) * 64 + j * 8];
                        for (int e = 0; e < 8; ++e) dst[e] = src[e];
                    }
        return upload_16(h, wt, out);
    }
    return upload_16(h, wp, out);
}

// ------------------------------------------------------------------------------------------------
// graph construction
// ------------------------------------------------------------------------------------------------
struct StatRef {
    int pslot = -1;   // buffer slot of partial sums [B][nslab][C][2]
    int nslab = 0;
};

struct Builder {
    bndm_unet *h;
    int rc = 0;
    int temb_cursor = 0;
    bool use_fused = true;
    bool use_gn_small = true;
    bool use_defer = true;                 // split-K slabs summed by the consuming gn_small
    float gn_eps = GN_EPS;                 // 1e-5 (UNet2DModel), 1e-6 (AutoencoderKL)
    bool has_temb = true;                  // ResnetBlock2D with a time_emb_proj (UNet) or without (VAE)
    int fused_min = 16, fused_max = 1 << 20;   // resolutions (H) handled by the fused conv path
    std::unordered_map<int, StatRef> stats_of;   // activation slot -> cached GroupNorm partial sums
    std::vector<float> tp_w, tp_b;      // concatenated time_emb_proj [ntemb][temb_dim], [ntemb]

    size_t act_bytes(int C, int H, int W) const { return (size_t)h->cfg.max_batch * H * W * C * 2; }
    Act new_act(int C, int H, int W) { return Act{h->new_slot(act_bytes(C, H, W)), C, H, W}; }
    Act scratch(int slot, int C, int H, int W) {
        h->grow(slot, act_bytes(C, H, W));
        return Act{slot, C, H, W};
    }

    std::string cur_name;
    void push(int cls, double flops, std::function<int(RunCtx &)> fn) {
        h->ops.push_back(Op{cls, flops, std::move(fn), cur_name});
        // kernel family from the op label's four-letter tag; conv_fused() overwrites it with the tile variant
        static const std::pair<const char *, const char *> fam[] = {
            {"cnvF", "conv_t32"}, {"conv", "conv_igemm"}, {"gnst", "gn_stats"}, {"gnfn", "gn_finalize2"},
            {"gnsm", "gn_small"}, {"gnap", "gn_apply"}, {"rdce", "splitk_reduce"}, {"attn", "attention"},
            {"temb", "temb_mlp"}, {"post", "pointwise_f32"}, {"deco", "conv_in"}, {"cnvS", "conv_s"}};
        Op &op = h->ops.back();
        op.kernel = cur_name.substr(0, cur_name.find(' '));
        for (const auto &f : fam)
            if (cur_name.compare(0, 4, f.first) == 0) op.kernel = f.second;
        if (cur_name.compare(0, 7, "conv_in") == 0 || cur_name == "decoder.conv_in") op.kernel = "conv_in";
    }

    // per-(sample, channel) partial sums of x: produced by the fused conv epilogue when possible,
    // otherwise by one gn_stats launch right here (once per tensor, reused by every consumer)
    StatRef ensure_stats(const Act &x) {
        auto it = stats_of.find(x.slot);
        if (it != stats_of.end()) return it->second;
        bndm_unet *hh = h;
        const int HW = x.H * x.W, nslab = gn_num_slabs(HW), C = x.C;
        StatRef sr{h->new_slot((size_t)h->cfg.max_batch * nslab * C * 2 * 4), nslab};
        const int sx = x.slot, sp = sr.pslot;
        cur_name = S("gnst %-44s C=%-4d %dx%d", "", C, x.H, x.W);
        push(OPC_OTHER, 0, [=](RunCtx &r) {
            return launch_gn_stats(hh->dtype(), hh->P(sx), C, nullptr, 0, r.B, HW, (float *)hh->P(sp), nslab, r.st);
        });
        stats_of[x.slot] = sr;
        return sr;
    }
    StatRef new_stats(const Act &x, int nslab) {
        StatRef sr{h->new_slot((size_t)h->cfg.max_batch * nslab * x.C * 2 * 4), nslab};
        stats_of[x.slot] = sr;
        return sr;
    }

    // GroupNorm(32)(cat(x1, x2)) for a fused consumer: either the consumer finalises the statistics itself from the
    // producers' per-tile partial sums (<= 16 slabs each: conv_t32's prologue), or a gn_finalize2 launch writes the
    // scale / shift table to scratch slot s_ss first
    struct GnSpec {
        StatRef a1, a2;
        int C1 = 0, C2 = 0, HW = 0;
        const float *gamma = nullptr, *beta = nullptr;
        bool inkernel = false;
    };
    bool use_gn_inkernel = true;
    GnSpec gn_table(const Act &x1, const Act *x2, const std::string &pname) {
        materialize();
        bndm_unet *hh = h;
        GnSpec g;
        const int C1 = x1.C, C2 = x2 ? x2->C : 0, C = C1 + C2, HW = x1.H * x1.W;
        const StatRef a1 = ensure_stats(x1);
        const StatRef a2 = x2 ? ensure_stats(*x2) : StatRef{};
        const float *gamma, *beta;
        if ((rc = upload_f32(h, h->hp(pname + ".weight"), &gamma))) return g;
        if ((rc = upload_f32(h, h->hp(pname + ".bias"), &beta))) return g;
        g.a1 = a1;
        g.a2 = a2;
        g.C1 = C1;
        g.C2 = C2;
        g.HW = HW;
        g.gamma = gamma;
        g.beta = beta;
        g.inkernel = use_gn_inkernel && a1.nslab <= 32 && (!x2 || a2.nslab <= 32) && C <= 512;
        if (g.inkernel) return g;
        h->grow(h->s_ss, (size_t)h->cfg.max_batch * 2 * C * 4);
        cur_name = S("gnfn %-44s C=%-4d %dx%d", pname.c_str(), C, x1.H, x1.W);
        const float eps_ = gn_eps;
        push(OPC_OTHER, 0, [=](RunCtx &r) {
            return launch_gn_finalize2((const float *)hh->P(a1.pslot), a1.nslab, C1,
                                       a2.pslot >= 0 ? (const float *)hh->P(a2.pslot) : nullptr, a2.nslab, C2, r.B, HW,
                                       GROUPS, eps_, gamma, beta, (float *)hh->P(hh->s_ss), r.st);
        });
        return g;
    }

    // GroupNorm(32) of cat(x1, x2) followed by optional SiLU, materialised -> out (unfused path)
    void group_norm(const Act &x1, const Act *x2, const std::string &pname, bool silu, const Act &out) {
        bndm_unet *hh = h;
        const int C1 = x1.C, C2 = x2 ? x2->C : 0, HW = x1.H * x1.W;
        if (HW <= 64 && use_gn_small) {
            // whole sample fits in L2 many times over: statistics + apply in one launch
            const float *gamma, *beta;
            if ((rc = upload_f32(h, h->hp(pname + ".weight"), &gamma))) return;
            if ((rc = upload_f32(h, h->hp(pname + ".bias"), &beta))) return;
            const int s1 = x1.slot, s2 = x2 ? x2->slot : -1, so = out.slot;
            const bool fused_reduce = pend.slot == x1.slot;
            const Pending q = pend;
            if (fused_reduce) pend.slot = -1;
            else materialize();
            cur_name = S("gnsm %-44s C=%-4d %dx%d", pname.c_str(), C1 + C2, x1.H, x1.W);
            const float eps_ = gn_eps;
            push(OPC_OTHER, 0, [=](RunCtx &r) {
                GnSlabSrc sl;
                if (fused_reduce) {
                    const ConvPlan pl = plan_conv(r.B * HW, C1, q.ksteps, hh->bufs[hh->s_splitk].bytes);
                    if (pl.splitk > 1) {
                        sl.part = (const float *)hh->P(hh->s_splitk);
                        sl.splitk = pl.splitk;
                        sl.bias = q.bias;
                        if (q.temb_off >= 0) {
                            sl.temb = r.tp_row ? r.tp_row : (const float *)hh->P(hh->s_tp);
                            sl.temb_bstride = r.tp_row ? 0 : hh->ntemb;
                            sl.temb_off = q.temb_off;
                        }
                        sl.resid = q.rs >= 0 ? hh->P(q.rs) : nullptr;
                        sl.raw_out = hh->P(s1);
                    }
                }
                return launch_gn_small(hh->dtype(), hh->P(s1), C1, s2 >= 0 ? hh->P(s2) : nullptr, C2, r.B, HW, GROUPS,
                                       eps_, gamma, beta, silu ? 1 : 0, hh->P(so), r.st, sl.part ? &sl : nullptr);
            });
            return;
        }
        materialize();
        {
            const bool keep = use_gn_inkernel;
            use_gn_inkernel = false;                     // gn_apply reads the table from s_ss
            gn_table(x1, x2, pname);
            use_gn_inkernel = keep;
        }
        if (rc) return;
        const int s1 = x1.slot, s2 = x2 ? x2->slot : -1, so = out.slot;
        cur_name = S("gnap %-44s C=%-4d %dx%d", pname.c_str(), C1 + C2, x1.H, x1.W);
        push(OPC_OTHER, 0, [=](RunCtx &r) {
            return launch_gn_apply(hh->dtype(), hh->P(s1), C1, s2 >= 0 ? hh->P(s2) : nullptr, C2,
                                   (const float *)hh->P(hh->s_ss), r.B, HW, silu ? 1 : 0, hh->P(so), r.st);
        });
    }

    // fused 3x3 conv launch (csrc/unet_conv32.hip)
    struct FIn {
        Act a;
        int taps, up, ss_off;
    };
    bool can_fuse(int H, int W, int Cout) const {
        return use_fused && H >= 16 && W >= 16 && H >= fused_min && H <= fused_max && Cout % 128 == 0;
    }
    // what conv_t32 can take (conv_t32_supports): a scale / shift table of at most 512 channels, at most 32 chunks of 32
    // channels per launch, source tensors below 2 GiB.  Layers outside (e.g. an up-block concat of 2 x 512 channels at
    // >= 16x16) run on the implicit-GEMM convolution + materialised GroupNorm instead.
    bool fits_t32(int Cin, int Cout, int H, int W) const {
        const long long px = (long long)h->cfg.max_batch * H * W;
        const int chunks2 = Cout / 32 + (Cin != Cout ? Cin / 32 : 0);
        return Cin <= 512 && Cout <= 512 && Cin / 32 <= 32 && chunks2 <= 32 && px * std::max(Cin, Cout) * 2 < (1LL << 31);
    }
    void conv_fused(const std::vector<FIn> &ins, const std::vector<WSeg> &ws, const GnSpec *gs, bool silu,
                    const float *bias, int temb_off, const Act *resid, const Act &out, bool want_stats,
                    const std::string &label, bool head = false) {
        // head: out.slot < 0 -- the result is the caller's fp32 NCHW tensor (conv_out), out.C <= 32 real channels
        const bool normed = gs != nullptr;
        const int ssC = gs ? gs->C1 + gs->C2 : 0;
        const GnSpec g = gs ? *gs : GnSpec{};
        const float eps_ = gn_eps;
        materialize();
        bndm_unet *hh = h;
        const void *Wp = nullptr;
        int Ktot = 0;
        for (const WSeg &w : ws) Ktot += w.taps * w.C;
        FusedArgs a{};
        a.nseg = (int)ins.size();
        std::vector<int> slots;
        double mac = 0;
        for (int i = 0; i < a.nseg; ++i) {
            a.seg[i].C = ins[i].a.C;
            a.seg[i].taps = ins[i].taps;
            a.seg[i].up = ins[i].up;
            a.seg[i].ss_off = ins[i].ss_off;
            slots.push_back(ins[i].a.slot);
            mac += (double)ins[i].taps * ins[i].a.C;
        }
        a.ssC = ssC;
        a.silu = silu ? 1 : 0;
        a.bias = bias;
        a.temb_off = temb_off >= 0 ? temb_off : 0;
        a.H = out.H;
        a.W = out.W;
        a.Cout = out.C;
        a.zeros = h->zeros;
        // 256-pixel tiles unless that leaves workgroup slots idle at this handle's batch size
        int TH = out.H >= 32 ? 16 : 8;
        const long long tiles16 = (long long)h->cfg.max_batch * (out.H / 16) * (out.W / 16) * (head ? 1 : out.C / 128);
        static const int th16_min = getenv("BNDM_TH16_MIN") ? atoi(getenv("BNDM_TH16_MIN")) : 192;
        if (TH == 16 && tiles16 < th16_min) TH = 8;
        {
            a.Ktot = Ktot;
            a.out_nchw32 = head ? 1 : 0;
            a.nco = 128;     // (64-channel tiles for the small grids were tried: the doubled patch DMA + normalisation loses)
            a.B = h->cfg.max_batch;
            a.ss = normed ? (const float *)h->zeros : nullptr;
            if (!conv_t32_supports(a)) {
                set_error("conv_t32 cannot run %s (segment list / ssC=%d / tensor size)", label.c_str(), ssC);
                rc = BNDM_E_ARG;
                return;
            }
            // conv_t32 normalises to log2(e) * silu(.) (one multiplication less per element, unet_conv32.hip: norm2): the
            // weights of a normalised segment carry the ln 2
            const std::vector<float> wp = pack_weights_t32(a.seg, a.nseg, out.C, [&](int si, int co, int c, int t) {
                const WSeg &w = ws[si];
                const float v = (*w.w)[((size_t)co * w.cin_total + w.c_begin + c) * w.taps + t];
                return a.seg[si].taps == 9 && a.seg[si].ss_off >= 0 ? 0.6931471805599453f * v : v;   // (1x1 chunks are read raw)
            }, head ? 32 : 128);
            if ((rc = upload_16(h, wp, &Wp))) return;
        }
        a.Wgt = Wp;
        a.Ktot = Ktot;
        const int rs = resid ? resid->slot : -1, so = out.slot;
        int pst = -1;
        if (want_stats) pst = new_stats(out, conv_t32_tiles_per_sample(TH, out.H, out.W)).pslot;
        else if (!head) stats_of.erase(out.slot);
        cur_name = S("cnvF %-44s K=%-5d N=%-4d %dx%d", label.c_str(), Ktot, out.C, out.H, out.W);
        double abytes = 2.0 * out.C * out.H * out.W * (resid ? 2 : 1);
        for (const FIn &f : ins) abytes += 2.0 * f.a.C * f.a.H * f.a.W;
        const double wbytes = 2.0 * Ktot * out.C;
        const size_t op_index = h->ops.size();
        push(OPC_CONV, 2.0 * mac * out.C * out.H * out.W, [=](RunCtx &r) {
            FusedArgs c = a;
            for (int i = 0; i < c.nseg; ++i) c.seg[i].src = hh->P(slots[i]);
            c.B = r.B;
            c.ss = normed ? (const float *)hh->P(hh->s_ss) : nullptr;
            if (normed && g.inkernel) {
                c.gn_p1 = (const float *)hh->P(g.a1.pslot);
                c.gn_p2 = g.a2.pslot >= 0 ? (const float *)hh->P(g.a2.pslot) : nullptr;
                c.gn_ns1 = g.a1.nslab;
                c.gn_ns2 = g.a2.nslab;
                c.gn_C1 = g.C1;
                c.gn_HW = g.HW;
                c.gn_gamma = g.gamma;
                c.gn_beta = g.beta;
                c.gn_eps = eps_;
            }
            c.temb = temb_off >= 0 ? (r.tp_row ? r.tp_row : (const float *)hh->P(hh->s_tp)) : nullptr;
            c.temb_bstride = r.tp_row ? 0 : hh->ntemb;
            c.resid = rs >= 0 ? hh->P(rs) : nullptr;
            c.out = head ? (void *)r.out : hh->P(so);
            c.stats = pst >= 0 ? (float *)hh->P(pst) : nullptr;
            return launch_conv_t32(hh->dtype(), TH, c, r.st);
        });
        h->ops[op_index].dominant = TH == 16 && !head;
        h->ops[op_index].kernel = S(head ? "conv_t32<TH=%d,N=32>" : "conv_t32<TH=%d>", TH);
        h->ops[op_index].bytes_per_sample = abytes;
        h->ops[op_index].bytes_fixed = wbytes;
    }

    struct SegIn {
        Act a;
        int taps, up;
    };

    // generic NHWC16 conv: out = sum over segments + bias (+temb) (+resid)
    // A split-K conv whose first consumer is a small GroupNorm leaves its fp32 slabs un-reduced (`pend`): gn_small sums
    // them while it reads its input (and stores the 16-bit tensor as a side product) -- one launch less per conv.
    // Any other consumer, and any conv that would reuse the slab workspace, calls materialize() first.
    struct Pending {
        int slot = -1, C = 0, H = 0, W = 0, ksteps = 0, temb_off = -1, rs = -1;
        const float *bias = nullptr;
    } pend;
    void materialize() {
        if (pend.slot < 0) return;
        bndm_unet *hh = h;
        const Pending q = pend;
        pend.slot = -1;
        cur_name = S("rdce %-44s C=%-4d %dx%d", "", q.C, q.H, q.W);
        push(OPC_OTHER, 0, [=](RunCtx &r) {
            const int M = r.B * q.H * q.W;
            const ConvPlan pl = plan_conv(M, q.C, q.ksteps, hh->bufs[hh->s_splitk].bytes);
            if (pl.splitk == 1) return 0;            // the conv wrote the tensor itself
            ConvArgs c{};
            c.B = r.B;
            c.H = q.H;
            c.W = q.W;
            c.Cout = q.C;
            c.bias = q.bias;
            if (q.temb_off >= 0) {
                c.temb = r.tp_row ? r.tp_row : (const float *)hh->P(hh->s_tp);
                c.temb_bstride = r.tp_row ? 0 : hh->ntemb;
                c.temb_off = q.temb_off;
            }
            c.resid = q.rs >= 0 ? hh->P(q.rs) : nullptr;
            c.out = hh->P(q.slot);
            return launch_splitk_reduce(hh->dtype(), (const float *)hh->P(hh->s_splitk), pl.splitk, c, r.st);
        });
    }

    void conv(const std::vector<SegIn> &ins, const void *Wp, int Ktot, const float *bias, int temb_off,
              const Act *resid, const Act &out, int stride, const std::string &label = "", bool defer = false) {
        materialize();
        bndm_unet *hh = h;
        cur_name = S("conv %-44s K=%-5d N=%-4d %dx%d", label.c_str(), Ktot, out.C, out.H, out.W);
        ConvArgs a{};
        a.nseg = (int)ins.size();
        std::vector<int> slots;
        double mac = 0;
        for (int i = 0; i < a.nseg; ++i) {
            a.seg[i].C = ins[i].a.C;
            a.seg[i].taps = ins[i].taps;
            a.seg[i].up = ins[i].up;
            slots.push_back(ins[i].a.slot);
            mac += (double)ins[i].taps * ins[i].a.C;
        }
        a.Wgt = Wp;
        a.bias = bias;
        a.temb_off = temb_off;
        a.H = out.H;
        a.W = out.W;
        a.stride = stride;
        a.Cout = out.C;
        a.Ktot = Ktot;
        a.zeros = h->zeros;
        {
            double act = 0;
            for (const SegIn &f : ins) act += (double)h->cfg.max_batch * f.a.H * f.a.W * f.a.C;
            a.wmajor = (double)out.C * Ktot > act ? 1 : 0;
            a.wtiled = Ktot % 64 == 0 ? 1 : 0;           // pack_conv_weight(..., 128, ...) tiles exactly then
        }
        {
            bool any_up = false;
            for (int i = 0; i < a.nseg; ++i) any_up = any_up || a.seg[i].up;
            if (!any_up) {
                const std::vector<int> tab = build_conv_steps(a.seg, a.nseg, out.W, stride);
                void *dtab;
                if ((rc = upload(h, tab.data(), tab.size() * sizeof(int), &dtab))) return;
                a.steps = dtab;
            }
        }
        const int rs = resid ? resid->slot : -1, so = out.slot;
        stats_of.erase(out.slot);
        const int ksteps = Ktot / 64;
        const double flops = 2.0 * mac * out.C * out.H * out.W;
        // split-K workspace is sized at launch-independent worst case (max_batch)
        {
            const int M = h->cfg.max_batch * out.H * out.W;
            const int nblk = ceil_div(M, 128) * ceil_div(out.C, 128);
            if (nblk < 192) h->grow(h->s_splitk, (size_t)32 * M * out.C * 4);
        }
        const bool can_defer = defer && use_defer && use_gn_small && out.H * out.W <= 64;
        push(OPC_CONV, flops, [=](RunCtx &r) mutable {
            ConvArgs c = a;
            for (int i = 0; i < c.nseg; ++i) c.seg[i].src = hh->P(slots[i]);
            c.B = r.B;
            c.resid = rs >= 0 ? hh->P(rs) : nullptr;
            if (temb_off >= 0) {
                c.temb = r.tp_row ? r.tp_row : (const float *)hh->P(hh->s_tp);
                c.temb_bstride = r.tp_row ? 0 : hh->ntemb;
            } else {
                c.temb = nullptr;
                c.temb_off = 0;
            }
            const int M = r.B * c.H * c.W;
            const ConvPlan pl = plan_conv(M, c.Cout, ksteps, hh->bufs[hh->s_splitk].bytes);
            const int tile = pl.tile, splitk = pl.splitk;
            if (splitk == 1) {
                c.splitk = 1;
                c.out = hh->P(so);
                return launch_conv(hh->dtype(), tile, EPI_NHWC16, c, r.st);
            }
            ConvArgs p = c;
            p.splitk = splitk;
            p.out = hh->P(hh->s_splitk);
            int e = launch_conv(hh->dtype(), tile, EPI_F32_ROWS, p, r.st);
            if (e || can_defer) return e;             // deferred: the consumer (or materialize()) sums the slabs
            c.out = hh->P(so);
            return launch_splitk_reduce(hh->dtype(), (const float *)hh->P(hh->s_splitk), splitk, c, r.st);
        });
        if (can_defer) {
            pend.slot = out.slot;
            pend.C = out.C;
            pend.H = out.H;
            pend.W = out.W;
            pend.ksteps = ksteps;
            pend.temb_off = temb_off;
            pend.rs = rs;
            pend.bias = bias;
        }
    }

    // ------------------------------------------------------------------------------------------------
    // <= 8x8 levels: conv_s (unet_tail.hip), one launch per convolution
    // ------------------------------------------------------------------------------------------------
    bool use_tail = true;
    std::vector<std::function<int()>> post_alloc;        // run by finalize once the activation buffers exist
    bool tail_ok(int H, int W) const {
        return use_tail && h->kind == 0 && H == W && (H == 8 || H == 4 || H == 2);
    }
    struct TSrc {
        Act a;
        int kind;                          // TailSegKind
        const std::vector<float> *w;       // OIHW weight the source multiplies
        int cin_total, c_begin;            // its input-channel range inside that weight
    };
    // GroupNorm(32) over Ctot channels, the part that covers x (channels [off, off + x.C)): written by x's producer when
    // its groups are whole 8 / 16 / 32-channel blocks of x
    bool norm_fits(const Act &x, int off, int Ctot) const {
        const int gs = Ctot / 32;
        return x.to && Ctot % 32 == 0 && (gs == 8 || gs == 16 || gs == 32) && off % gs == 0 && x.C % gs == 0 &&
               x.to->reqs.size() < 3;
    }
    Act request_norm(const Act &x, const std::string &pname, int off, int Ctot, bool silu) {
        const std::vector<float> &g = h->hp(pname + ".weight"), &b = h->hp(pname + ".bias");
        const std::vector<float> gs_(g.begin() + off, g.begin() + off + x.C), bs_(b.begin() + off, b.begin() + off + x.C);
        const float *gd = nullptr, *bd = nullptr;
        if ((rc = upload_f32(h, gs_, &gd)) || (rc = upload_f32(h, bs_, &bd))) return x;
        Act n = new_act(x.C, x.H, x.W);
        x.to->reqs.push_back(TailOut::Req{n.slot, gd, bd, Ctot / 32, silu ? 1 : 0});
        return n;
    }
    // normalised inputs of a convolution over GroupNorm(cat(x1, x2)): the producers' copies, or one gn_small launch when
    // the groups straddle the tensors (e.g. 512 + 256 channels: groups of 24)
    std::vector<TSrc> normed_sources(const Act &x1, const Act *x2, const std::string &pname, bool silu, int kind,
                                     const std::vector<float> *w) {
        const int C1 = x1.C, C2 = x2 ? x2->C : 0, C = C1 + C2;
        std::vector<TSrc> out;
        if (norm_fits(x1, 0, C) && (!x2 || norm_fits(*x2, C1, C))) {
            out.push_back(TSrc{request_norm(x1, pname, 0, C, silu), kind, w, C, 0});
            if (x2 && !rc) out.push_back(TSrc{request_norm(*x2, pname, C1, C, silu), kind, w, C, C1});
            return out;
        }
        materialize();                      // (this gn_small reads the 16-bit tensors, not a deferred conv's slabs)
        bndm_unet *hh = h;
        Act y = new_act(C, x1.H, x1.W);
        const float *gamma, *beta;
        if ((rc = upload_f32(h, h->hp(pname + ".weight"), &gamma)) || (rc = upload_f32(h, h->hp(pname + ".bias"), &beta)))
            return out;
        if (x1.to) x1.to->raw = true;
        if (x2 && x2->to) x2->to->raw = true;
        const int s1 = x1.slot, s2 = x2 ? x2->slot : -1, so = y.slot, HW = x1.H * x1.W;
        const float eps_ = gn_eps;
        cur_name = S("gnsm %-44s C=%-4d %dx%d", pname.c_str(), C, x1.H, x1.W);
        push(OPC_OTHER, 0, [=](RunCtx &r) {
            return launch_gn_small(hh->dtype(), hh->P(s1), C1, s2 >= 0 ? hh->P(s2) : nullptr, C2, r.B, HW, GROUPS, eps_,
                                   gamma, beta, silu ? 1 : 0, hh->P(so), r.st, nullptr);
        });
        out.push_back(TSrc{y, kind, w, C, 0});
        return out;
    }
    // one conv_s launch: out[H x W x Cout] = sum over sources (+ bias / time embedding row, + residual); qkv: the q|k|v
    // projection (rows [Wq; Wk; Wv]) + attention of C / 8 heads -> out is the attention output (before to_out)
    Act conv_tail(const std::vector<TSrc> &srcs, int Cout, int H, int W, const float *bias, int temb_off, const Act *resid,
                  const std::string &label, bool qkv = false) {
        materialize();                      // conv_s reads its sources' 16-bit tensors: a deferred split-K conv is summed first
        bndm_unet *hh = h;
        const int HW = H * W, NB = qkv ? 3 : 1, D = tail_ring_depth(NB), rows = qkv ? 3 * Cout : Cout;
        if (Cout % 32 || (int)srcs.size() > 4) {
            set_error("conv_s cannot run %s (Cout=%d, %d sources)", label.c_str(), Cout, (int)srcs.size());
            rc = BNDM_E_ARG;
            return srcs[0].a;
        }
        // conv_s addresses its sources and outputs with 32-bit byte offsets
        for (const TSrc &t : srcs)
            if ((long long)h->cfg.max_batch * t.a.H * t.a.W * t.a.C * 2 >= (1LL << 31) ||
                (long long)h->cfg.max_batch * HW * rows * 2 >= (1LL << 31)) {
                set_error("conv_s: %s exceeds 2 GiB per tensor at max_batch=%d (32-bit offsets); lower max_batch",
                          label.c_str(), h->cfg.max_batch);
                rc = BNDM_E_ARG;
                return srcs[0].a;
            }
        const int ntn = Cout / 32;
        int TM = 64;
        if (!qkv && HW == 64 && (long long)h->cfg.max_batch * HW / 128 * ntn >= 192) TM = 128;
        std::vector<TailSeg> segs;
        double mac = 0;
        for (const TSrc &t : srcs) {
            if (t.a.C % 32) {
                set_error("conv_s: %d input channels in %s", t.a.C, label.c_str());
                rc = BNDM_E_ARG;
                return srcs[0].a;
            }
            segs.push_back(TailSeg{t.kind, t.a.C});
            mac += (double)(t.kind == TAIL_SEG_1x1 ? 1 : 9) * t.a.C;
        }
        const TailPlan plan = build_tail_plan(
            segs, rows, NB, D,
            [&](int si, int row, int c, int tap) {
                const TSrc &t = srcs[si];
                const int taps = t.kind == TAIL_SEG_1x1 ? 1 : 9;
                return (*t.w)[((size_t)row * t.cin_total + t.c_begin + c) * taps + tap];
            },
            [&](int nt, int nb, int n) { return qkv ? nb * Cout + nt * 32 + n : nt * 32 + n; });
        const void *dW = nullptr;
        void *dDesc = nullptr, *dRounds = nullptr;
        if ((rc = upload_16(h, plan.wgt, &dW))) return srcs[0].a;
        if ((rc = upload(h, plan.desc.data(), plan.desc.size() * 4, &dDesc))) return srcs[0].a;
        std::vector<TailRound> rt(plan.nrounds);
        if ((rc = upload(h, rt.data(), rt.size() * sizeof(TailRound), &dRounds))) return srcs[0].a;
        {
            std::vector<int> slots;
            for (const TSrc &t : srcs) slots.push_back(t.a.slot);
            const std::vector<TailPlanRound> pr = plan.rounds;
            std::vector<int> cs;
            for (const TSrc &t : srcs) cs.push_back(t.a.C);
            post_alloc.push_back([=]() {
                std::vector<TailRound> tab(pr.size());
                for (size_t i = 0; i < pr.size(); ++i) {
                    tab[i].src = hh->P(slots[pr[i].seg]);
                    tab[i].row_bytes = cs[pr[i].seg] * 2;
                    tab[i].cbyte = pr[i].c0 * 2;
                    tab[i].mode = pr[i].mode;
                    tab[i].phase = pr[i].phase;
                    tab[i].nsub = pr[i].nsub;
                    tab[i].pad = 0;
                }
                BNDM_CHECK_HIP(hipMemcpy(dRounds, tab.data(), tab.size() * sizeof(TailRound), hipMemcpyHostToDevice));
                return 0;
            });
        }
        Act out = new_act(Cout, H, W);
        if (!qkv) out.to = std::make_shared<TailOut>();
        for (const TSrc &t : srcs)
            if (t.a.to && t.kind != TAIL_SEG_3x3 && t.kind != TAIL_SEG_1x1) t.a.to->raw = true;   // resampling reads the raw tensor
        if (resid && resid->to) resid->to->raw = true;
        TailArgs a{};
        a.wgt = dW;
        a.desc = (const uint32_t *)dDesc;
        a.rounds = (const TailRound *)dRounds;
        a.nrounds = plan.nrounds;
        a.maxsteps = plan.maxsteps;
        for (int i = 0; i < 8; ++i) a.nuse[i] = plan.nuse[i];
        a.tile_bytes = (long long)plan.tile_elems * 2;
        a.wave_bytes = (int)(plan.wave_elems * 2);
        a.hwlog = 31 - __builtin_clz(HW);
        a.wlog = 31 - __builtin_clz(W);
        a.Cout = Cout;
        a.ntn = ntn;
        a.bias = bias;
        a.temb_off = temb_off >= 0 ? temb_off : 0;
        a.eps = gn_eps;
        a.epi = qkv ? TAIL_EPI_ATTN : TAIL_EPI_CONV;
        const int rs = resid ? resid->slot : -1, so = out.slot;
        const std::shared_ptr<TailOut> to = out.to;
        cur_name = S("cnvS %-44s K=%-5d N=%-4d %dx%d", label.c_str(), (int)mac, rows, H, W);
        const size_t op_index = h->ops.size();
        std::vector<int> rslots, rcs;
        for (const TSrc &t : srcs) {
            rslots.push_back(t.a.slot);
            rcs.push_back(t.a.C);
        }
        const std::vector<TailPlanRound> prs(plan.rounds.begin(), plan.rounds.begin() + std::min(2, plan.nrounds));
        push(OPC_CONV, 2.0 * mac * rows * HW + (qkv ? 4.0 * HW * HW * Cout : 0.0), [=](RunCtx &r) {
            TailArgs c = a;
            c.B = r.B;
            for (size_t i = 0; i < prs.size(); ++i) {
                TailRound &d = i ? c.r1 : c.r0;
                d = TailRound{hh->P(rslots[prs[i].seg]), rcs[prs[i].seg] * 2, prs[i].c0 * 2, prs[i].mode, prs[i].phase,
                              prs[i].nsub, 0};
            }
            c.temb = temb_off >= 0 ? (r.tp_row ? r.tp_row : (const float *)hh->P(hh->s_tp)) : nullptr;
            c.temb_bstride = r.tp_row ? 0 : hh->ntemb;
            c.resid = rs >= 0 ? hh->P(rs) : nullptr;
            if (qkv) {
                c.attn_out = hh->P(so);
            } else {
                c.raw_out = to->raw ? hh->P(so) : nullptr;
                c.nreq = (int)to->reqs.size();
                for (int i = 0; i < c.nreq; ++i)
                    c.req[i] = TailNorm{hh->P(to->reqs[i].slot), to->reqs[i].gamma, to->reqs[i].beta, to->reqs[i].gs,
                                        to->reqs[i].silu};
            }
            return launch_conv_tail(hh->dtype(), TM, NB, c, r.st);
        });
        h->ops[op_index].kernel = qkv ? "conv_s<qkv+attention>" : S("conv_s<TM=%d>", TM);
        h->ops[op_index].bytes_fixed = 2.0 * mac * rows;
        return out;
    }

    Act resnet_tail(const Act &x1, const Act *x2, int Cout, const std::string &name, int temb_off) {
        const int C1 = x1.C, C2 = x2 ? x2->C : 0, Cin = C1 + C2, H = x1.H, W = x1.W;
        const std::vector<TSrc> s1 =
            normed_sources(x1, x2, name + ".norm1", true, TAIL_SEG_3x3, &h->hp(name + ".conv1.weight"));
        if (rc) return x1;
        const float *bias1 = has_temb ? nullptr : bias_of(name + ".conv1");
        Act h1 = conv_tail(s1, Cout, H, W, bias1, temb_off, nullptr, name + ".conv1");
        if (rc) return x1;
        const bool h1_local = norm_fits(h1, 0, Cout);
        std::vector<TSrc> s2 = normed_sources(h1, nullptr, name + ".norm2", true, TAIL_SEG_3x3, &h->hp(name + ".conv2.weight"));
        if (rc) return x1;
        if (h1_local) h1.to->raw = false;                 // conv2 reads the normalised copy only
        const float *b2;
        if (Cin != Cout) {
            const std::string sc = name + ".conv_shortcut";
            s2.push_back(TSrc{x1, TAIL_SEG_1x1, &h->hp(sc + ".weight"), Cin, 0});
            if (x2) s2.push_back(TSrc{*x2, TAIL_SEG_1x1, &h->hp(sc + ".weight"), Cin, C1});
            if (x1.to) x1.to->raw = true;
            if (x2 && x2->to) x2->to->raw = true;
            b2 = bias_of(name + ".conv2", &sc);
        } else {
            b2 = bias_of(name + ".conv2");
        }
        if (rc) return x1;
        return conv_tail(s2, Cout, H, W, b2, -1, Cin == Cout ? &x1 : nullptr, name + (Cin != Cout ? ".conv2+sc" : ".conv2"));
    }

    // Attention block (iadb_bn.py:209-228 AttnDownBlock2D / AttnUpBlock2D, heads of 8 channels): GroupNorm by the
    // producer, q|k|v + softmax(q k^T / sqrt(8)) v in one launch, to_out + residual in the second
    Act attention_tail(const Act &x, const std::string &name) {
        const int C = x.C;
        wcat_store.emplace_back();
        std::vector<float> &wcat = wcat_store.back();
        std::vector<float> bcat;
        for (const char *p : {".to_q", ".to_k", ".to_v"}) {
            const std::vector<float> &w = h->hp(name + p + ".weight"), &b = h->hp(name + p + ".bias");
            wcat.insert(wcat.end(), w.begin(), w.end());
            bcat.insert(bcat.end(), b.begin(), b.end());
        }
        const std::vector<TSrc> sq = normed_sources(x, nullptr, name + ".group_norm", false, TAIL_SEG_1x1, &wcat);
        if (rc) return x;
        const float *bq;
        if ((rc = upload_f32(h, bcat, &bq))) return x;
        Act att = conv_tail(sq, C, x.H, x.W, bq, -1, nullptr, name + ".qkv+attn", true);
        if (rc) return x;
        return conv_tail({TSrc{att, TAIL_SEG_1x1, &h->hp(name + ".to_out.0.weight"), C, 0}}, C, x.H, x.W,
                         bias_of(name + ".to_out.0"), -1, &x, name + ".to_out");
    }
    std::list<std::vector<float>> wcat_store;

    const float *bias_of(const std::string &n, const std::string *plus = nullptr) {
        std::vector<float> b = h->hp(n + ".bias");
        if (plus) {
            const std::vector<float> &q = h->hp(*plus + ".bias");
            for (size_t i = 0; i < b.size(); ++i) b[i] += q[i];
        }
        const float *d = nullptr;
        if (!rc) rc = upload_f32(h, b, &d);
        return d;
    }

    // ResnetBlock2D on cat(x1, x2): returns the block output (new unique activation)
    Act resnet(const Act &x1, const Act *x2, int Cout, const std::string &name) {
        const int C1 = x1.C, C2 = x2 ? x2->C : 0, Cin = C1 + C2, H = x1.H, W = x1.W;
        // time_emb_proj rows appended to the shared projection matrix
        const int temb_off = has_temb ? temb_cursor : -1;
        if (has_temb) {
            const std::vector<float> &w = h->hp(name + ".time_emb_proj.weight");
            const std::vector<float> &b = h->hp(name + ".time_emb_proj.bias");
            tp_w.insert(tp_w.end(), w.begin(), w.end());
            tp_b.insert(tp_b.end(), b.begin(), b.end());
            // conv1's bias rides in the projection's bias: the conv epilogue then adds ONE row (temb) instead of two
            const std::vector<float> &cb = h->hp(name + ".conv1.bias");
            for (int c = 0; c < Cout; ++c) tp_b[temb_cursor + c] += cb[c];
            temb_cursor += Cout;
        }
        if (tail_ok(H, W)) return resnet_tail(x1, x2, Cout, name, temb_off);
        const float *bias1 = has_temb ? nullptr : bias_of(name + ".conv1");
        if (can_fuse(H, W, Cout) && fits_t32(Cin, Cout, H, W)) {
            // conv1: GN(norm1)+SiLU applied to cat(x1, x2) inside the conv prologue
            const GnSpec g1 = gn_table(x1, x2, name + ".norm1");
            if (rc) return x1;
            std::vector<WSeg> w1{WSeg{&h->hp(name + ".conv1.weight"), Cin, 0, C1, 9}};
            std::vector<FIn> in1{FIn{x1, 9, 0, 0}};
            if (x2) {
                w1.push_back(WSeg{&h->hp(name + ".conv1.weight"), Cin, C1, C2, 9});
                in1.push_back(FIn{*x2, 9, 0, C1});
            }
            Act h1 = scratch(h->s_h1, Cout, H, W);
            conv_fused(in1, w1, &g1, true, bias1, temb_off, nullptr, h1, true, name + ".conv1");
            if (rc) return x1;
            // conv2 (+ 1x1 conv_shortcut on the raw inputs, or identity residual)
            const GnSpec g2 = gn_table(h1, nullptr, name + ".norm2");
            if (rc) return x1;
            Act out = new_act(Cout, H, W);
            std::vector<WSeg> w2{WSeg{&h->hp(name + ".conv2.weight"), Cout, 0, Cout, 9}};
            std::vector<FIn> in2{FIn{h1, 9, 0, 0}};
            const float *b2;
            if (Cin != Cout) {
                const std::string sc = name + ".conv_shortcut";
                w2.push_back(WSeg{&h->hp(sc + ".weight"), Cin, 0, C1, 1});
                in2.push_back(FIn{x1, 1, 0, -1});
                if (x2) {
                    w2.push_back(WSeg{&h->hp(sc + ".weight"), Cin, C1, C2, 1});
                    in2.push_back(FIn{*x2, 1, 0, -1});
                }
                b2 = bias_of(name + ".conv2", &sc);
            } else {
                b2 = bias_of(name + ".conv2");
            }
            if (rc) return x1;
            conv_fused(in2, w2, &g2, true, b2, -1, Cin == Cout ? &x1 : nullptr, out, true,
                       name + (Cin != Cout ? ".conv2+sc" : ".conv2"));
            return out;
        }
        Act y1 = scratch(h->s_y, Cin, H, W);
        group_norm(x1, x2, name + ".norm1", true, y1);
        if (rc) return x1;
        const void *W1;
        int K1;
        rc = pack_conv_weight(h, {WSeg{&h->hp(name + ".conv1.weight"), Cin, 0, Cin, 9}}, Cout, 128, &W1, &K1);
        if (rc) return x1;
        Act h1 = scratch(h->s_h1, Cout, H, W);
        conv({SegIn{y1, 9, 0}}, W1, K1, bias1, temb_off, nullptr, h1, 1, name + ".conv1", true);
        Act y2 = scratch(h->s_y2, Cout, H, W);
        group_norm(h1, nullptr, name + ".norm2", true, y2);
        if (rc) return x1;
        Act out = new_act(Cout, H, W);
        const void *W2;
        int K2;
        if (Cin != Cout) {
            const std::string sc = name + ".conv_shortcut";
            std::vector<WSeg> ws{WSeg{&h->hp(name + ".conv2.weight"), Cout, 0, Cout, 9},
                                 WSeg{&h->hp(sc + ".weight"), Cin, 0, C1, 1}};
            std::vector<SegIn> ins{SegIn{y2, 9, 0}, SegIn{x1, 1, 0}};
            if (x2) {
                ws.push_back(WSeg{&h->hp(sc + ".weight"), Cin, C1, C2, 1});
                ins.push_back(SegIn{*x2, 1, 0});
            }
            rc = pack_conv_weight(h, ws, Cout, 128, &W2, &K2);
            if (rc) return x1;
            conv(ins, W2, K2, bias_of(name + ".conv2", &sc), -1, nullptr, out, 1, name + ".conv2+sc", true);
        } else {
            rc = pack_conv_weight(h, {WSeg{&h->hp(name + ".conv2.weight"), Cout, 0, Cout, 9}}, Cout, 128, &W2, &K2);
            if (rc) return x1;
            conv({SegIn{y2, 9, 0}}, W2, K2, bias_of(name + ".conv2"), -1, &x1, out, 1, name + ".conv2", true);
        }
        return out;
    }

    Act attention(const Act &x, const std::string &name) {
        bndm_unet *hh = h;
        const int C = x.C, H = x.H, W = x.W, T = H * W;
        if (tail_ok(H, W) && C % 32 == 0) return attention_tail(x, name);
        Act yn = scratch(h->s_y, C, H, W);
        group_norm(x, nullptr, name + ".group_norm", false, yn);
        if (rc) return x;
        // fused q/k/v projection: rows [Wq; Wk; Wv]
        std::vector<float> wcat, bcat;
        for (const char *p : {".to_q", ".to_k", ".to_v"}) {
            const std::vector<float> &w = h->hp(name + p + ".weight");
            const std::vector<float> &b = h->hp(name + p + ".bias");
            wcat.insert(wcat.end(), w.begin(), w.end());
            bcat.insert(bcat.end(), b.begin(), b.end());
        }
        const void *Wqkv;
        int Kq;
        rc = pack_conv_weight(h, {WSeg{&wcat, C, 0, C, 1}}, 3 * C, 128, &Wqkv, &Kq);
        if (rc) return x;
        const float *bq;
        if ((rc = upload_f32(h, bcat, &bq))) return x;
        Act qkv = scratch(h->s_qkv, 3 * C, H, W);
        conv({SegIn{yn, 1, 0}}, Wqkv, Kq, bq, -1, nullptr, qkv, 1, name + ".qkv");
        Act att = scratch(h->s_att, C, H, W);
        const int sq = qkv.slot, sa = att.slot;
        cur_name = S("attn %s T=%d", name.c_str(), T);
        push(OPC_OTHER, 0, [=](RunCtx &r) {
            return launch_attention(hh->dtype(), hh->P(sq), hh->P(sa), r.B, T, C, r.st);
        });
        const void *Wo;
        int Ko;
        rc = pack_conv_weight(h, {WSeg{&h->hp(name + ".to_out.0.weight"), C, 0, C, 1}}, C, 128, &Wo, &Ko);
        if (rc) return x;
        Act out = new_act(C, H, W);
        conv({SegIn{att, 1, 0}}, Wo, Ko, bias_of(name + ".to_out.0"), -1, &x, out, 1, name + ".to_out");
        return out;
    }

    // AutoencoderKL mid-block attention: ONE head of dim C over T = H*W tokens (4096 at a 64x64 latent).  Built from
    // the GEMM kernel: q, k projections; V^T = Wv . yn_b^T (the projection computed transposed, its bias added after
    // P.V because softmax rows sum to 1); S = q_b . k_b^T; row softmax in place; O = P . V; output projection + x.
    Act attention_big(const Act &x, const std::string &name) {
        bndm_unet *hh = h;
        const int C = x.C, H = x.H, W = x.W, T = H * W, MB = h->cfg.max_batch;
        if ((C & (C - 1)) || C < 128 || T % 128 || T > 4096) {
            set_error("attention_big: C=%d T=%d unsupported (C a power of two >= 128, T a multiple of 128 <= 4096)", C, T);
            rc = BNDM_E_ARG;
            return x;
        }
        Act yn = scratch(h->s_y, C, H, W);
        group_norm(x, nullptr, name + ".group_norm", false, yn);
        if (rc) return x;
        const void *Wq, *Wk, *Wo, *Wv;
        int K;
        if ((rc = pack_conv_weight(h, {WSeg{&h->hp(name + ".to_q.weight"), C, 0, C, 1}}, C, 128, &Wq, &K))) return x;
        if ((rc = pack_conv_weight(h, {WSeg{&h->hp(name + ".to_k.weight"), C, 0, C, 1}}, C, 128, &Wk, &K))) return x;
        if ((rc = pack_conv_weight(h, {WSeg{&h->hp(name + ".to_out.0.weight"), C, 0, C, 1}}, C, 128, &Wo, &K))) return x;
        if ((rc = upload_16(h, h->hp(name + ".to_v.weight"), &Wv))) return x;       // [C out][C in]: the A operand of V^T
        const float *bqd = bias_of(name + ".to_q"), *bkd = bias_of(name + ".to_k"), *bvd = bias_of(name + ".to_v");
        if (rc) return x;
        Act q = Act{h->new_slot(act_bytes(C, H, W)), C, H, W}, k = Act{h->new_slot(act_bytes(C, H, W)), C, H, W};
        conv({SegIn{yn, 1, 0}}, Wq, K, bqd, -1, nullptr, q, 1, name + ".to_q");
        conv({SegIn{yn, 1, 0}}, Wk, K, bkd, -1, nullptr, k, 1, name + ".to_k");
        if (rc) return x;
        materialize();
        const int s_vt = h->new_slot((size_t)C * T * 2), s_p = h->new_slot((size_t)T * T * 2);
        Act att = Act{h->new_slot(act_bytes(C, H, W)), C, H, W};
        const int sq = q.slot, sk = k.slot, sy = yn.slot, sa = att.slot;
        int vh = 1, vw = C;                            // V^T rows (= channels) viewed as a vh x vw "image"
        while (vh < vw) {
            vh <<= 1;
            vw >>= 1;
        }
        (void)MB;
        cur_name = S("attn %s T=%d C=%d", name.c_str(), T, C);
        push(OPC_CONV, 2.0 * T * C * C + 2.0 * 2 * T * (double)T * C, [=](RunCtx &r) {
            auto gemm = [&](const void *A, int Ah, int Aw, int Kd, const void *Wt, int N, const float *bias, void *out) {
                ConvArgs c{};
                c.nseg = 1;
                c.seg[0].src = A;
                c.seg[0].C = Kd;
                c.seg[0].taps = 1;
                c.seg[0].up = 0;
                c.Wgt = Wt;
                c.bias = bias;
                c.H = Ah;
                c.W = Aw;
                c.stride = 1;
                c.Cout = N;
                c.Ktot = Kd;
                c.zeros = hh->zeros;
                c.B = 1;
                c.splitk = 1;
                c.out = out;
                return launch_conv(hh->dtype(), Ah * Aw >= 2048 ? TILE_256x128 : TILE_128x128, EPI_NHWC16, c, r.st);
            };
            const size_t per = (size_t)T * C * 2;
            for (int b = 0; b < r.B; ++b) {
                const char *qb = (const char *)hh->P(sq) + b * per, *kb = (const char *)hh->P(sk) + b * per;
                const char *yb = (const char *)hh->P(sy) + b * per;
                int e;
                if ((e = gemm(Wv, vh, vw, C, yb, T, nullptr, hh->P(s_vt)))) return e;               // V^T  [C][T]
                if ((e = gemm(qb, H, W, C, kb, T, nullptr, hh->P(s_p)))) return e;                  // S    [T][T]
                if ((e = launch_softmax_rows(hh->dtype(), hh->P(s_p), T, T, 1.0f / sqrtf((float)C), r.st))) return e;
                if ((e = gemm(hh->P(s_p), H, W, T, hh->P(s_vt), C, bvd, (char *)hh->P(sa) + b * per))) return e;   // O
            }
            return 0;
        });
        Act out = new_act(C, H, W);
        conv({SegIn{att, 1, 0}}, Wo, K, bias_of(name + ".to_out.0"), -1, &x, out, 1, name + ".to_out");
        return out;
    }

    Act resample(const Act &x, const std::string &name, bool down) {
        if (tail_ok(down ? x.H / 2 : x.H * 2, down ? x.W / 2 : x.W * 2))
            return conv_tail({TSrc{x, down ? TAIL_SEG_3x3_S2 : TAIL_SEG_3x3_UP, &h->hp(name + ".weight"), x.C, 0}}, x.C,
                             down ? x.H / 2 : x.H * 2, down ? x.W / 2 : x.W * 2, bias_of(name), -1, nullptr, name);
        Act out = down ? new_act(x.C, x.H / 2, x.W / 2) : new_act(x.C, x.H * 2, x.W * 2);
        if (!down && can_fuse(out.H, out.W, out.C) && fits_t32(x.C, x.C, out.H, out.W)) {
            conv_fused({FIn{x, 9, 1, -1}}, {WSeg{&h->hp(name + ".weight"), x.C, 0, x.C, 9}}, nullptr, false, bias_of(name),
                       -1, nullptr, out, true, name);
            return out;
        }
        const void *Wp;
        int K;
        rc = pack_conv_weight(h, {WSeg{&h->hp(name + ".weight"), x.C, 0, x.C, 9}}, x.C, 128, &Wp, &K);
        if (rc) return x;
        conv({SegIn{x, 9, down ? 0 : 1}}, Wp, K, bias_of(name), -1, nullptr, out, down ? 2 : 1, name, true);
        return out;
    }

    // AutoencoderKL.decode: post_quant_conv -> Decoder (conv_in, mid block, up blocks, GN + SiLU + conv_out)
    int build_vae() {
        bndm_unet *hh = h;
        const bndm_unet_config &c = h->cfg;
        const int n = c.num_levels, R = c.resolution, L = c.in_channels, top = c.block_out_channels[n - 1];
        const int MB = c.max_batch;
        has_temb = false;
        gn_eps = 1e-6f;
        h->s_y = h->new_slot(0);
        h->s_h1 = h->new_slot(0);
        h->s_y2 = h->new_slot(0);
        h->s_part = h->new_slot(0);
        h->s_ss = h->new_slot(0);
        h->s_qkv = h->new_slot(0);
        h->s_att = h->new_slot(0);
        h->s_splitk = h->new_slot((size_t)64 << 20);
        h->s_tp = h->new_slot(0);
        h->s_d = h->new_slot((size_t)MB * L * R * R * 4);          // post_quant_conv output, fp32 NCHW
        {
            void *z;
            std::vector<char> zz(256, 0);
            if ((rc = upload(h, zz.data(), zz.size(), &z))) return rc;
            h->zeros = z;
        }
        {
            const float *wq, *bq;
            if ((rc = upload_f32(h, h->hp("post_quant_conv.weight"), &wq))) return rc;
            if ((rc = upload_f32(h, h->hp("post_quant_conv.bias"), &bq))) return rc;
            cur_name = "post_quant_conv";
            push(OPC_OTHER, 0, [=](RunCtx &r) {
                return launch_pointwise_f32(r.sample, wq, bq, (float *)hh->P(hh->s_d), r.B, L, L, R * R, r.st);
            });
        }
        Act x = new_act(top, R, R);
        {
            const int KP = ceil_div(9 * L, 16) * 16;
            const std::vector<float> &w = h->hp("decoder.conv_in.weight");
            std::vector<float> wp((size_t)top * KP, 0.f);
            for (int co = 0; co < top; ++co)
                for (int k = 0; k < 9 * L; ++k) wp[(size_t)co * KP + k] = w[(size_t)co * 9 * L + k];
            const void *dW;
            if ((rc = upload_16(h, wp, &dW))) return rc;
            const float *db = bias_of("decoder.conv_in");
            if (rc) return rc;
            const int so = x.slot;
            const int sst = new_stats(x, R * R / 128).pslot;
            cur_name = "decoder.conv_in";
            push(OPC_OTHER, 0, [=](RunCtx &r) {
                return launch_conv_in(hh->dtype(), (const float *)hh->P(hh->s_d), L, nullptr, 0, dW, db, hh->P(so),
                                      (float *)hh->P(sst), r.B, R, R, top, KP, r.st);
            });
        }
        x = resnet(x, nullptr, top, "decoder.mid_block.resnets.0");
        if (rc) return rc;
        x = attention_big(x, "decoder.mid_block.attentions.0");
        if (rc) return rc;
        x = resnet(x, nullptr, top, "decoder.mid_block.resnets.1");
        if (rc) return rc;
        for (int i = 0; i < n; ++i) {
            const int oc = c.block_out_channels[n - 1 - i];
            for (int j = 0; j < c.layers_per_block + 1; ++j) {
                x = resnet(x, nullptr, oc, S("decoder.up_blocks.%d.resnets.%d", i, j));
                if (rc) return rc;
            }
            if (i != n - 1) {
                x = resample(x, S("decoder.up_blocks.%d.upsamplers.0.conv", i), false);
                if (rc) return rc;
            }
        }
        {
            // decoder.conv_norm_out + SiLU + decoder.conv_out in one conv_t32 launch (fp32 NCHW images)
            const int C0 = x.C, RO = x.H;
            const GnSpec g = gn_table(x, nullptr, "decoder.conv_norm_out");
            if (rc) return rc;
            conv_fused({FIn{x, 9, 0, 0}}, {WSeg{&h->hp("decoder.conv_out.weight"), C0, 0, C0, 9}}, &g, true,
                       bias_of("decoder.conv_out"), -1, nullptr, Act{-1, c.out_channels, RO, RO}, false,
                       "decoder.conv_norm_out + conv_out", true);
            if (rc) return rc;
        }
        materialize();
        return rc;
    }

    int build() {
        bndm_unet *hh = h;
        const bndm_unet_config &c = h->cfg;
        const int *boc = c.block_out_channels;
        const int n = c.num_levels, R = c.resolution, D = h->temb_dim, C0 = boc[0];
        const int MB = c.max_batch;

        // fixed scratch slots
        h->s_y = h->new_slot(0);
        h->s_h1 = h->new_slot(0);
        h->s_y2 = h->new_slot(0);
        h->s_part = h->new_slot(0);
        h->s_ss = h->new_slot(0);
        h->s_qkv = h->new_slot(0);
        h->s_att = h->new_slot(0);
        h->s_splitk = h->new_slot((size_t)64 << 20);
        h->s_actemb = h->new_slot((size_t)MB * D * 2);
        h->s_tp = h->new_slot(0);   // sized once ntemb is known
        h->s_d = h->new_slot((size_t)MB * c.out_channels * R * R * 4);
        h->s_t = h->new_slot((size_t)MB * 4);
        {
            void *z;
            std::vector<char> zz(256, 0);
            if ((rc = upload(h, zz.data(), zz.size(), &z))) return rc;
            h->zeros = z;
        }

        // ---- time embedding MLP (fp32, transposed weights for coalesced reads) ----------------------
        {
            const std::vector<float> &w1 = h->hp("time_embedding.linear_1.weight");   // [D][C0]
            const std::vector<float> &w2 = h->hp("time_embedding.linear_2.weight");   // [D][D]
            std::vector<float> w1t((size_t)C0 * D), w2t((size_t)D * D);
            for (int o = 0; o < D; ++o)
                for (int k = 0; k < C0; ++k) w1t[(size_t)k * D + o] = w1[(size_t)o * C0 + k];
            for (int o = 0; o < D; ++o)
                for (int k = 0; k < D; ++k) w2t[(size_t)k * D + o] = w2[(size_t)o * D + k];
            const float *dw1, *dw2, *db1, *db2;
            if ((rc = upload_f32(h, w1t, &dw1))) return rc;
            if ((rc = upload_f32(h, w2t, &dw2))) return rc;
            if ((rc = upload_f32(h, h->hp("time_embedding.linear_1.bias"), &db1))) return rc;
            if ((rc = upload_f32(h, h->hp("time_embedding.linear_2.bias"), &db2))) return rc;
            cur_name = "temb_mlp";
            push(OPC_OTHER, 0, [=](RunCtx &r) {
                if (r.tp_row) return 0;
                return launch_temb_mlp(hh->dtype(), r.timesteps, r.B, C0, D, dw1, db1, dw2, db2, hh->P(hh->s_actemb),
                                       r.st);
            });
            hh->temb_table_fn = [=](int n, const float *t_dev, void *act, float *table, hipStream_t st) {
                return launch_temb_mlp(hh->dtype(), t_dev, n, C0, D, dw1, db1, dw2, db2, act, st);
            };
        }
        const size_t temb_proj_op = h->ops.size();
        cur_name = "conv time_emb_proj (all resnets)";
        push(OPC_CONV, 0, [](RunCtx &) { return 0; });   // placeholder: all time_emb_proj as one GEMM

        // ---- conv_in ------------------------------------------------------------------------------
        Act x = new_act(C0, R, R);
        {
            const int Cin = c.in_channels, KP = ceil_div(9 * Cin, 16) * 16;
            if (KP > 64) {
                set_error("conv_in: in_channels=%d not supported (9*Cin must be <= 64)", Cin);
                return BNDM_E_ARG;
            }
            const std::vector<float> &w = h->hp("conv_in.weight");   // [C0][Cin][3][3] -> k = ci*9 + t
            std::vector<float> wp((size_t)C0 * KP, 0.f);
            for (int co = 0; co < C0; ++co)
                for (int k = 0; k < 9 * Cin; ++k) wp[(size_t)co * KP + k] = w[(size_t)co * 9 * Cin + k];
            const void *dW;
            if ((rc = upload_16(h, wp, &dW))) return rc;
            const float *db = bias_of("conv_in");
            if (rc) return rc;
            const int so = x.slot;
            const int sst = new_stats(x, R * R / 128).pslot;
            cur_name = "conv_in";
            push(OPC_OTHER, 0, [=](RunCtx &r) {
                const int Ce = r.extra ? Cin / 2 : 0;     // conditional sampler: x and x_c have equal channels
                return launch_conv_in(hh->dtype(), r.sample, Cin - Ce, r.extra, Ce, dW, db, hh->P(so),
                                      (float *)hh->P(sst), r.B, R, R, C0, KP, r.st);
            });
        }

        // ---- down path ----------------------------------------------------------------------------
        std::vector<Act> skips{x};
        for (int i = 0; i < n; ++i) {
            for (int j = 0; j < c.layers_per_block; ++j) {
                x = resnet(x, nullptr, boc[i], S("down_blocks.%d.resnets.%d", i, j));
                if (rc) return rc;
                if (c.down_attn[i]) x = attention(x, S("down_blocks.%d.attentions.%d", i, j));
                if (rc) return rc;
                skips.push_back(x);
            }
            if (i != n - 1) {
                x = resample(x, S("down_blocks.%d.downsamplers.0.conv", i), true);
                if (rc) return rc;
                skips.push_back(x);
            }
        }
        // ---- mid ----------------------------------------------------------------------------------
        x = resnet(x, nullptr, boc[n - 1], "mid_block.resnets.0");
        if (rc) return rc;
        x = attention(x, "mid_block.attentions.0");
        if (rc) return rc;
        x = resnet(x, nullptr, boc[n - 1], "mid_block.resnets.1");
        if (rc) return rc;
        // ---- up path ------------------------------------------------------------------------------
        for (int i = 0; i < n; ++i) {
            const int oc = boc[n - 1 - i];
            for (int j = 0; j < c.layers_per_block + 1; ++j) {
                Act sk = skips.back();
                skips.pop_back();
                x = resnet(x, &sk, oc, S("up_blocks.%d.resnets.%d", i, j));
                if (rc) return rc;
                if (c.up_attn[i]) x = attention(x, S("up_blocks.%d.attentions.%d", i, j));
                if (rc) return rc;
            }
            if (i != n - 1) {
                x = resample(x, S("up_blocks.%d.upsamplers.0.conv", i), false);
                if (rc) return rc;
            }
        }
        // ---- head ---------------------------------------------------------------------------------
        if (use_fused && R >= 16 && C0 <= 512) {
            // conv_norm_out + SiLU + conv_out in one conv_t32 launch (32-channel tiles, fp32 NCHW output)
            const GnSpec g = gn_table(x, nullptr, "conv_norm_out");
            if (rc) return rc;
            conv_fused({FIn{x, 9, 0, 0}}, {WSeg{&h->hp("conv_out.weight"), C0, 0, C0, 9}}, &g, true, bias_of("conv_out"), -1,
                       nullptr, Act{-1, c.out_channels, R, R}, false, "conv_norm_out + conv_out", true);
            if (rc) return rc;
        } else {
            Act y = scratch(h->s_y, C0, R, R);
            group_norm(x, nullptr, "conv_norm_out", true, y);
            if (rc) return rc;
            const void *Wp;
            int K;
            rc = pack_conv_weight(h, {WSeg{&h->hp("conv_out.weight"), C0, 0, C0, 9}}, c.out_channels, 32, &Wp, &K);
            if (rc) return rc;
            const float *db = bias_of("conv_out");
            if (rc) return rc;
            ConvArgs a{};
            a.nseg = 1;
            a.seg[0].C = C0;
            a.seg[0].taps = 9;
            a.seg[0].up = 0;
            a.Wgt = Wp;
            a.bias = db;
            a.H = R;
            a.W = R;
            a.stride = 1;
            a.Cout = c.out_channels;
            a.Ktot = K;
            a.splitk = 1;
            a.zeros = h->zeros;
            const int sy = y.slot;
            cur_name = S("conv conv_out K=%d N=%d %dx%d", K, c.out_channels, R, R);
            push(OPC_CONV, 2.0 * 9 * C0 * c.out_channels * R * R, [=](RunCtx &r) {
                ConvArgs cc = a;
                cc.seg[0].src = hh->P(sy);
                cc.B = r.B;
                cc.out = r.out;
                return launch_conv(hh->dtype(), TILE_128x32, EPI_NCHW32, cc, r.st);
            });
        }

        // ---- all time_emb_proj layers as one [B x D] . [D x ntemb] GEMM -----------------------------
        {
            h->ntemb = temb_cursor;
            h->grow(h->s_tp, (size_t)MB * h->ntemb * 4);
            std::vector<WSeg> ws{WSeg{&tp_w, D, 0, D, 1}};
            const void *Wp;
            int K;
            if ((rc = pack_conv_weight(h, ws, h->ntemb, 128, &Wp, &K))) return rc;
            const float *db;
            if ((rc = upload_f32(h, tp_b, &db))) return rc;
            ConvArgs a{};
            a.nseg = 1;
            a.seg[0].C = D;
            a.seg[0].taps = 1;
            a.seg[0].up = 0;
            a.Wgt = Wp;
            a.bias = db;
            a.H = 1;
            a.W = 1;
            a.stride = 1;
            a.Cout = h->ntemb;
            a.Ktot = K;
            a.splitk = 1;
            a.zeros = h->zeros;
            a.wtiled = 1;
            auto mlp_fn = h->temb_table_fn;
            h->temb_table_fn = [=](int n, const float *t_dev, void *act, float *table, hipStream_t st) {
                int e = mlp_fn(n, t_dev, act, table, st);
                if (e) return e;
                ConvArgs cc = a;
                cc.seg[0].src = act;
                cc.B = n;
                cc.out = table;
                return launch_conv(hh->dtype(), TILE_128x128, EPI_F32_ROWS, cc, st);
            };
            h->ops[temb_proj_op] = Op{OPC_CONV, 2.0 * D * h->ntemb, [=](RunCtx &r) {
                                          if (r.tp_row) return 0;
                                          ConvArgs cc = a;
                                          cc.seg[0].src = hh->P(hh->s_actemb);
                                          cc.B = r.B;
                                          cc.out = hh->P(hh->s_tp);
                                          return launch_conv(hh->dtype(), TILE_128x128, EPI_F32_ROWS, cc, r.st);
                                      }, "conv time_emb_proj (all resnets)"};
            h->ops[temb_proj_op].kernel = "conv_igemm";
        }
        return 0;
    }
};

int run_forward(bndm_unet *h, RunCtx &r) {
    if (h->f32) return f32_model_forward(h->f32, r.sample, r.extra, r.timesteps, r.out, r.B, r.st);
    for (size_t i = 0; i < h->ops.size(); ++i) {
        if (r.prof) BNDM_CHECK_HIP(hipEventRecord((*r.ev)[2 * i], r.st));
        int e = h->ops[i].run(r);
        if (e) return e;
        if (r.prof) BNDM_CHECK_HIP(hipEventRecord((*r.ev)[2 * i + 1], r.st));
    }
    return 0;
}

// time-embedding projections of every step of a schedule, computed once per sampling call (K10)
int prepare_temb_table(bndm_unet *h, int n, const float *t_host, hipStream_t st) {
    if (!h->t_uploaded) BNDM_CHECK_HIP(hipEventCreateWithFlags(&h->t_uploaded, hipEventDisableTiming));
    else BNDM_CHECK_HIP(hipEventSynchronize(h->t_uploaded));      // the previous call's upload has left t_pinned
    if (n > h->tp_cap) {
        // no stream synchronisation: launches already queued may still read the old tables, so they are kept until
        // the handle is destroyed (a schedule table is n * ntemb floats, ~10 MB for 250 steps)
        for (void *p : {(void *)h->tp_table, (void *)h->t_steps, h->act_steps})
            if (p) h->retired.push_back(p);
        if (h->t_pinned) (void)hipHostFree(h->t_pinned);
        h->tp_table = nullptr; h->t_steps = nullptr; h->act_steps = nullptr; h->t_pinned = nullptr; h->tp_cap = 0;
        BNDM_CHECK_HIP(hipMalloc((void **)&h->tp_table, (size_t)n * h->ntemb * 4));
        BNDM_CHECK_HIP(hipMalloc((void **)&h->t_steps, (size_t)n * 4));
        BNDM_CHECK_HIP(hipMalloc(&h->act_steps, (size_t)n * h->temb_dim * 2));
        BNDM_CHECK_HIP(hipHostMalloc((void **)&h->t_pinned, (size_t)n * 4, hipHostMallocDefault));
        h->tp_cap = n;
    }
    memcpy(h->t_pinned, t_host, (size_t)n * 4);
    BNDM_CHECK_HIP(hipMemcpyAsync(h->t_steps, h->t_pinned, (size_t)n * 4, hipMemcpyHostToDevice, st));
    BNDM_CHECK_HIP(hipEventRecord(h->t_uploaded, st));
    return h->temb_table_fn(n, h->t_steps, h->act_steps, h->tp_table, st);
}

__global__ void fill_f32_kernel(float *p, float v, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

int check_ready(const bndm_unet *h, int B, const char *what) {
    if (!h) {
        set_error("%s: NULL handle", what);
        return BNDM_E_ARG;
    }
    if (!h->finalized) {
        set_error("%s: handle not finalised (load every parameter, then bndm_unet_finalize)", what);
        return BNDM_E_STATE;
    }
    if (B < 1 || B > h->cfg.max_batch) {
        set_error("%s: batch %d outside [1, max_batch=%d]", what, B, h->cfg.max_batch);
        return BNDM_E_ARG;
    }
    return 0;
}

}  // namespace

// ================================================================================================
// C ABI
// ================================================================================================
extern "C" int bndm_unet_create(bndm_unet **out, const bndm_unet_config *cfg) {
    BNDM_REQUIRE(out && cfg, "bndm_unet_create: NULL argument");
    BNDM_REQUIRE(cfg->num_levels >= 2 && cfg->num_levels <= BNDM_MAX_LEVELS, "bndm_unet_create: num_levels %d",
                 cfg->num_levels);
    BNDM_REQUIRE(cfg->dtype == BNDM_DTYPE_F16 || cfg->dtype == BNDM_DTYPE_BF16 || cfg->dtype == BNDM_DTYPE_F32,
                 "bndm_unet_create: dtype %d", cfg->dtype);
    BNDM_REQUIRE(cfg->dtype != BNDM_DTYPE_F32 || cfg->max_batch <= 8,
                 "bndm_unet_create: the fp32-compute mode is a verification mode, max_batch %d > 8", cfg->max_batch);
    BNDM_REQUIRE(cfg->resolution >= 16 && (cfg->resolution & (cfg->resolution - 1)) == 0,
                 "bndm_unet_create: resolution %d must be a power of two >= 16", cfg->resolution);
    BNDM_REQUIRE((cfg->resolution >> (cfg->num_levels - 1)) >= 1, "bndm_unet_create: too many levels for resolution");
    BNDM_REQUIRE(cfg->in_channels >= 1 && cfg->in_channels * 9 <= 64 && cfg->out_channels >= 1 &&
                     cfg->out_channels <= 32,
                 "bndm_unet_create: in/out channels %d/%d unsupported", cfg->in_channels, cfg->out_channels);
    BNDM_REQUIRE(cfg->max_batch >= 1 && cfg->layers_per_block >= 1, "bndm_unet_create: bad max_batch/layers");
    for (int i = 0; i < cfg->num_levels; ++i)
        BNDM_REQUIRE(cfg->block_out_channels[i] % 64 == 0 && cfg->block_out_channels[i] <= 512,
                     "bndm_unet_create: block_out_channels[%d]=%d must be a multiple of 64 and <= 512", i,
                     cfg->block_out_channels[i]);
    // conv_in_kernel / temb_mlp_kernel layouts: 256 threads cover C0/8 16-byte chunks per pixel row, D = 4*C0 <= 1024
    BNDM_REQUIRE(cfg->block_out_channels[0] == 64 || cfg->block_out_channels[0] == 128 ||
                     cfg->block_out_channels[0] == 256,
                 "bndm_unet_create: block_out_channels[0]=%d must be 64, 128 or 256", cfg->block_out_channels[0]);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
        (void)hipGetLastError();
        set_error("bndm_unet_create: no HIP device visible (this path has no CPU fallback)");
        return BNDM_E_NODEVICE;
    }
    bndm_unet *h = new (std::nothrow) bndm_unet();
    if (!h) return BNDM_E_NOMEM;
    h->cfg = *cfg;
    h->temb_dim = cfg->block_out_channels[0] * 4;
    build_specs(h);
    h->host.resize(h->params.size());
    h->loaded.assign(h->params.size(), 0);
    *out = h;
    return 0;
}

extern "C" int bndm_vae_decoder_create(bndm_unet **out, const bndm_vae_config *cfg) {
    BNDM_REQUIRE(out && cfg, "bndm_vae_decoder_create: NULL argument");
    BNDM_REQUIRE(cfg->num_levels >= 2 && cfg->num_levels <= BNDM_MAX_LEVELS, "bndm_vae_decoder_create: num_levels %d",
                 cfg->num_levels);
    BNDM_REQUIRE(cfg->dtype == BNDM_DTYPE_F16 || cfg->dtype == BNDM_DTYPE_BF16, "bndm_vae_decoder_create: dtype %d",
                 cfg->dtype);
    BNDM_REQUIRE(cfg->latent_resolution >= 16 && cfg->latent_resolution <= 64 &&
                     (cfg->latent_resolution & (cfg->latent_resolution - 1)) == 0,
                 "bndm_vae_decoder_create: latent resolution %d must be 16, 32 or 64", cfg->latent_resolution);
    BNDM_REQUIRE(cfg->latent_channels >= 1 && cfg->latent_channels * 9 <= 64 && cfg->out_channels >= 1 &&
                     cfg->out_channels <= 32,
                 "bndm_vae_decoder_create: latent/out channels %d/%d unsupported", cfg->latent_channels,
                 cfg->out_channels);
    BNDM_REQUIRE(cfg->max_batch >= 1 && cfg->max_batch <= 8 && cfg->layers_per_block >= 1,
                 "bndm_vae_decoder_create: max_batch %d outside [1, 8]", cfg->max_batch);
    for (int i = 0; i < cfg->num_levels; ++i)
        BNDM_REQUIRE(cfg->block_out_channels[i] % 128 == 0 && cfg->block_out_channels[i] <= 512,
                     "bndm_vae_decoder_create: block_out_channels[%d]=%d must be a multiple of 128 and <= 512", i,
                     cfg->block_out_channels[i]);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
        (void)hipGetLastError();
        set_error("bndm_vae_decoder_create: no HIP device visible (this path has no CPU fallback)");
        return BNDM_E_NODEVICE;
    }
    bndm_unet *h = new (std::nothrow) bndm_unet();
    if (!h) return BNDM_E_NOMEM;
    h->kind = 1;
    h->cfg.in_channels = cfg->latent_channels;
    h->cfg.out_channels = cfg->out_channels;
    h->cfg.resolution = cfg->latent_resolution;
    h->cfg.num_levels = cfg->num_levels;
    for (int i = 0; i < cfg->num_levels; ++i) h->cfg.block_out_channels[i] = cfg->block_out_channels[i];
    h->cfg.layers_per_block = cfg->layers_per_block;
    h->cfg.dtype = cfg->dtype;
    h->cfg.max_batch = cfg->max_batch;
    build_specs_vae(h);
    h->host.resize(h->params.size());
    h->loaded.assign(h->params.size(), 0);
    *out = h;
    return 0;
}

extern "C" int bndm_vae_decode(bndm_unet *h, const float *latents, float *out, int B, void *stream) {
    int rc = check_ready(h, B, "bndm_vae_decode");
    if (rc) return rc;
    BNDM_REQUIRE(h->kind == 1, "bndm_vae_decode: the handle is not a VAE decoder");
    BNDM_REQUIRE(latents && out, "bndm_vae_decode: NULL tensor");
    RunCtx r{B, (hipStream_t)stream, latents, nullptr, nullptr, out};
    return run_forward(h, r);
}

extern "C" void bndm_unet_destroy(bndm_unet *h) {
    if (!h) return;
    for (void *p : h->weights) (void)hipFree(p);
    if (h->tp_table) (void)hipFree(h->tp_table);
    if (h->t_steps) (void)hipFree(h->t_steps);
    if (h->act_steps) (void)hipFree(h->act_steps);
    if (h->t_pinned) (void)hipHostFree(h->t_pinned);
    if (h->t_uploaded) (void)hipEventDestroy(h->t_uploaded);
    for (void *p : h->retired) (void)hipFree(p);
    if (h->f32) f32_model_destroy(h->f32);
    for (Buf &b : h->bufs)
        if (b.ptr) (void)hipFree(b.ptr);
    delete h;
}

extern "C" int bndm_unet_num_params(const bndm_unet *h) { return h ? (int)h->params.size() : 0; }

extern "C" int bndm_unet_param_info(const bndm_unet *h, int index, char *name, size_t name_len, int64_t *numel) {
    BNDM_REQUIRE(h && index >= 0 && index < (int)h->params.size(), "bndm_unet_param_info: bad index %d", index);
    if (name && name_len) snprintf(name, name_len, "%s", h->params[index].name.c_str());
    if (numel) *numel = h->params[index].numel;
    return 0;
}

extern "C" int bndm_unet_load_param(bndm_unet *h, const char *name, const float *host_data, int64_t numel) {
    BNDM_REQUIRE(h && name && host_data, "bndm_unet_load_param: NULL argument");
    if (h->finalized) {
        set_error("bndm_unet_load_param: handle already finalised");
        return BNDM_E_STATE;
    }
    auto it = h->pindex.find(name);
    BNDM_REQUIRE(it != h->pindex.end(), "bndm_unet_load_param: unexpected key '%s'", name);
    const ParamSpec &ps = h->params[it->second];
    BNDM_REQUIRE(ps.numel == numel, "bndm_unet_load_param: size mismatch for '%s': expected %lld, got %lld", name,
                 (long long)ps.numel, (long long)numel);
    h->host[it->second].assign(host_data, host_data + numel);
    h->loaded[it->second] = 1;
    return 0;
}

extern "C" int bndm_unet_num_ops(const bndm_unet *h) { return h && h->finalized ? (int)h->ops.size() : 0; }

extern "C" int bndm_unet_op_info(const bndm_unet *h, int index, char *kernel, size_t kernel_len, char *label,
                                 size_t label_len, double *flops_per_sample) {
    BNDM_REQUIRE(h && h->finalized && index >= 0 && index < (int)h->ops.size(), "bndm_unet_op_info: bad index %d", index);
    const Op &op = h->ops[index];
    if (kernel && kernel_len) snprintf(kernel, kernel_len, "%s", op.kernel.c_str());
    if (label && label_len) snprintf(label, label_len, "%s", op.name.c_str());
    if (flops_per_sample) *flops_per_sample = op.flops_per_sample;
    return 0;
}

extern "C" int bndm_unet_finalize(bndm_unet *h) {
    BNDM_REQUIRE(h, "bndm_unet_finalize: NULL handle");
    if (h->finalized) return 0;
    for (size_t i = 0; i < h->params.size(); ++i)
        if (!h->loaded[i]) {
            set_error("bndm_unet_finalize: missing key '%s'", h->params[i].name.c_str());
            return BNDM_E_STATE;
        }
    if (h->cfg.dtype == BNDM_DTYPE_F32) {
        std::vector<std::string> names;
        for (const ParamSpec &ps : h->params) names.push_back(ps.name);
        int rc = f32_model_create(h->cfg, names, h->host, &h->f32);
        if (rc) return rc;
        h->s_t = h->new_slot((size_t)h->cfg.max_batch * 4);
        h->s_d = h->new_slot((size_t)h->cfg.max_batch * h->cfg.out_channels * h->cfg.resolution * h->cfg.resolution * 4);
        for (Buf &bf : h->bufs) BNDM_CHECK_HIP(hipMalloc(&bf.ptr, bf.bytes ? bf.bytes : 16));
        for (auto &v : h->host) std::vector<float>().swap(v);
        BNDM_CHECK_HIP(hipDeviceSynchronize());
        h->finalized = true;
        return 0;
    }
    Builder b{h};
    if (const char *e = getenv("BNDM_NO_FUSED")) b.use_fused = !(e[0] == '1');     // debugging: igemm + gn_apply everywhere
    if (const char *e = getenv("BNDM_NO_GN_SMALL")) b.use_gn_small = !(e[0] == '1');
    if (const char *e = getenv("BNDM_NO_DEFER")) b.use_defer = !(e[0] == '1');
    if (const char *e = getenv("BNDM_NO_TAIL")) b.use_tail = !(e[0] == '1');       // <= 8x8 levels on igemm + gn_small
    int rc = h->kind == 1 ? b.build_vae() : b.build();
    if (rc) return rc;
    {
        // the roofline figure follows the 256-pixel-tile launches; a handle too small for any of them (max_batch 8 at
        // 64 px) reports its 128-pixel-tile launches instead
        bool any = false;
        for (const Op &o : h->ops) any = any || o.dominant;
        if (!any)
            for (Op &o : h->ops) o.dominant = o.kernel == "conv_t32<TH=8>";
    }
    for (Buf &bf : h->bufs) {
        BNDM_CHECK_HIP(hipMalloc(&bf.ptr, bf.bytes ? bf.bytes : 16));
    }
    for (auto &fn : b.post_alloc)
        if ((rc = fn())) return rc;
    for (auto &v : h->host) std::vector<float>().swap(v);
    BNDM_CHECK_HIP(hipDeviceSynchronize());
    h->finalized = true;
    return 0;
}

extern "C" int bndm_unet_forward(bndm_unet *h, const float *sample, const float *timesteps, float *out, int B,
                                 void *stream) {
    int rc = check_ready(h, B, "bndm_unet_forward");
    if (rc) return rc;
    BNDM_REQUIRE(h->kind == 0, "bndm_unet_forward: the handle is a VAE decoder");
    BNDM_REQUIRE(sample && timesteps && out, "bndm_unet_forward: NULL tensor");
    RunCtx r{B, (hipStream_t)stream, sample, nullptr, timesteps, out};
    return run_forward(h, r);
}

extern "C" int bndm_unet_sample_iadb(bndm_unet *h, float *x, const float *extra_in, int B, int C, int nb_step,
                                     const float *t_in, const float *da, const float *dg,
                                     const uint8_t *snap_mask, float *snapshots, void *stream) {
    int rc = check_ready(h, B, "bndm_unet_sample_iadb");
    if (rc) return rc;
    BNDM_REQUIRE(h->kind == 0, "bndm_unet_sample_iadb: the handle is a VAE decoder");
    BNDM_REQUIRE(x && t_in && da && dg && nb_step >= 0, "bndm_unet_sample_iadb: NULL table");
    const int Cin = h->cfg.in_channels, Cout = h->cfg.out_channels, R = h->cfg.resolution;
    BNDM_REQUIRE((extra_in ? 2 * C : C) == Cin, "bndm_unet_sample_iadb: x has %d channels, model takes %d%s", C, Cin,
                 extra_in ? " (with conditioning)" : "");
    BNDM_REQUIRE(Cout == C || Cout == 2 * C, "bndm_unet_sample_iadb: out_channel %d for %d image channels", Cout, C);
    hipStream_t st = (hipStream_t)stream;
    float *tbuf = (float *)h->P(h->s_t), *dbuf = (float *)h->P(h->s_d);
    const size_t img = (size_t)B * C * R * R;
    int snap = 0;
    if (nb_step > 0 && !h->f32 && (rc = prepare_temb_table(h, nb_step, t_in, st))) return rc;
    for (int s = 0; s < nb_step; ++s) {
        RunCtx r{B, st, x, extra_in, tbuf, dbuf};
        if (h->f32) hipLaunchKernelGGL(fill_f32_kernel, dim3(1), dim3(64), 0, st, tbuf, t_in[s], B);
        else r.tp_row = h->tp_table + (size_t)s * h->ntemb;
        if ((rc = run_forward(h, r))) return rc;
        if ((rc = bndm_iadb_step(x, dbuf, da[s], dg[s], B, C, Cout, R * R, stream))) return rc;
        if (snap_mask && snapshots && snap_mask[s]) {
            BNDM_CHECK_HIP(hipMemcpyAsync(snapshots + (size_t)snap * img, x, img * 4, hipMemcpyDeviceToDevice, st));
            ++snap;
        }
    }
    return 0;
}

extern "C" int bndm_unet_sample_ddim(bndm_unet *h, float *x, int B, int nb_step, const float *coef, float clip,
                                     void *stream) {
    int rc = check_ready(h, B, "bndm_unet_sample_ddim");
    if (rc) return rc;
    BNDM_REQUIRE(x && coef, "bndm_unet_sample_ddim: NULL argument");
    BNDM_REQUIRE(h->cfg.in_channels == h->cfg.out_channels, "bndm_unet_sample_ddim: eps-prediction needs Cin == Cout");
    hipStream_t st = (hipStream_t)stream;
    const int R = h->cfg.resolution;
    float *tbuf = (float *)h->P(h->s_t), *dbuf = (float *)h->P(h->s_d);
    const size_t n = (size_t)B * h->cfg.in_channels * R * R;
    if (nb_step > 0) {
        std::vector<float> ts(nb_step);
        for (int s = 0; s < nb_step; ++s) ts[s] = coef[5 * s];
        if (!h->f32 && (rc = prepare_temb_table(h, nb_step, ts.data(), st))) return rc;   // copied to pinned staging there
    }
    for (int s = 0; s < nb_step; ++s) {
        const float *c = coef + 5 * s;
        RunCtx r{B, st, x, nullptr, tbuf, dbuf};
        if (h->f32) hipLaunchKernelGGL(fill_f32_kernel, dim3(1), dim3(64), 0, st, tbuf, c[0], B);
        else r.tp_row = h->tp_table + (size_t)s * h->ntemb;
        if ((rc = run_forward(h, r))) return rc;
        if ((rc = bndm_ddim_step(x, dbuf, c[1], c[2], c[3], c[4], clip, n, stream))) return rc;
    }
    return 0;
}

extern "C" int bndm_unet_profile(bndm_unet *h, const float *sample, const float *timesteps, float *out, int B,
                                 int iters, bndm_unet_profile_t *prof, void *stream) {
    int rc = check_ready(h, B, "bndm_unet_profile");
    if (rc) return rc;
    BNDM_REQUIRE(sample && timesteps && out && prof && iters >= 1, "bndm_unet_profile: bad argument");
    BNDM_REQUIRE(!h->f32, "bndm_unet_profile: the fp32-compute mode has no per-kernel profile");
    hipStream_t st = (hipStream_t)stream;
    const size_t nops = h->ops.size();
    std::vector<hipEvent_t> ev(2 * nops);
    for (auto &e : ev) BNDM_CHECK_HIP(hipEventCreate(&e));
    hipEvent_t t0, t1;
    BNDM_CHECK_HIP(hipEventCreate(&t0));
    BNDM_CHECK_HIP(hipEventCreate(&t1));
    // (1) whole forward, back-to-back launches
    RunCtx r{B, st, sample, nullptr, timesteps, out};
    if ((rc = run_forward(h, r))) return rc;   // warm-up
    BNDM_CHECK_HIP(hipEventRecord(t0, st));
    for (int i = 0; i < iters; ++i)
        if ((rc = run_forward(h, r))) return rc;
    BNDM_CHECK_HIP(hipEventRecord(t1, st));
    BNDM_CHECK_HIP(hipEventSynchronize(t1));
    float ms = 0;
    BNDM_CHECK_HIP(hipEventElapsedTime(&ms, t0, t1));
    prof->ms_total = ms / iters;
    // (2) per-op events, accumulated by class
    double conv_ms = 0, conv_flops = 0;
    int conv_launches = 0;
    std::vector<double> op_ms(nops, 0.0);
    for (int i = 0; i < iters; ++i) {
        RunCtx rp{B, st, sample, nullptr, timesteps, out};
        rp.prof = true;
        rp.ev = &ev;
        if ((rc = run_forward(h, rp))) return rc;
        BNDM_CHECK_HIP(hipStreamSynchronize(st));
        for (size_t k = 0; k < nops; ++k) {
            float m = 0;
            BNDM_CHECK_HIP(hipEventElapsedTime(&m, ev[2 * k], ev[2 * k + 1]));
            op_ms[k] += m;
            if (h->ops[k].cls == OPC_CONV) conv_ms += m;
        }
    }
    // profiling aid (BNDM_PROFILE_HOT=1): every op repeated 10x back to back right after a forward -- its operands and
    // weights are then cache-resident, which separates a kernel's own cost from cold-HBM / first-touch effects
    std::vector<double> hot_ms(nops, 0.0);
    if (getenv("BNDM_PROFILE_HOT")) {
        RunCtx rp{B, st, sample, nullptr, timesteps, out};
        if ((rc = run_forward(h, rp))) return rc;
        for (size_t k = 0; k < nops; ++k) {
            if ((rc = h->ops[k].run(rp))) return rc;
            BNDM_CHECK_HIP(hipEventRecord(t0, st));
            for (int i = 0; i < 10; ++i)
                if ((rc = h->ops[k].run(rp))) return rc;
            BNDM_CHECK_HIP(hipEventRecord(t1, st));
            BNDM_CHECK_HIP(hipEventSynchronize(t1));
            float m = 0;
            BNDM_CHECK_HIP(hipEventElapsedTime(&m, t0, t1));
            hot_ms[k] = m / 10;
        }
    }
    if (const char *dump = getenv("BNDM_PROFILE_DUMP")) {
        FILE *f = fopen(dump, "w");
        if (f) {
            for (size_t k = 0; k < nops; ++k) {
                const double ms_k = op_ms[k] / iters, fl = h->ops[k].flops_per_sample * B;
                fprintf(f, "%3zu %8.4f ms %8.1f TF/s  %s", k, ms_k, ms_k > 0 ? fl / (ms_k * 1e-3) / 1e12 : 0.0,
                        h->ops[k].name.c_str());
                if (hot_ms[k] > 0) fprintf(f, "   [hot x10: %.4f ms]", hot_ms[k]);
                fprintf(f, "\n");
            }
            fclose(f);
        }
    }
    for (size_t k = 0; k < nops; ++k)
        if (h->ops[k].cls == OPC_CONV) {
            conv_flops += h->ops[k].flops_per_sample * B;
            ++conv_launches;
        }
    double dom_ms = 0, dom_flops = 0, dom_bytes = 0;
    int dom_n = 0;
    for (size_t k = 0; k < nops; ++k)
        if (h->ops[k].dominant) {
            dom_ms += op_ms[k] / iters;
            dom_flops += h->ops[k].flops_per_sample * B;
            dom_bytes += h->ops[k].bytes_per_sample * B + h->ops[k].bytes_fixed;
            ++dom_n;
        }
    prof->ms_dom = (float)dom_ms;
    prof->dom_launches = dom_n;
    prof->dom_flops = dom_flops;
    prof->dom_bytes = dom_bytes;
    prof->ms_conv = (float)(conv_ms / iters);
    prof->conv_flops = conv_flops;
    prof->conv_launches = conv_launches;
    prof->launches = (int)nops;
    for (auto &e : ev) (void)hipEventDestroy(e);
    (void)hipEventDestroy(t0);
    (void)hipEventDestroy(t1);
    return 0;
}
