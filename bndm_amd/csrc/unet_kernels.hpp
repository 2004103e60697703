// Launch interface of the UNet kernels (csrc/unet_kernels.hip), used by the engine.
// Activations are NHWC 16-bit (f16 or bf16, chosen per handle); statistics, biases, time
// embeddings and accumulators are fp32.
#pragma once
#include "common.hpp"
#include <functional>
#include <vector>

namespace bndm {

constexpr int CONV_MAX_SEG = 4;

// One K-segment of an implicit-GEMM convolution: a source tensor read with 3x3 (pad 1) or 1x1 taps.
// Several segments express torch.cat([h, skip], 1) feeding conv1, and conv2 + the 1x1 conv_shortcut
// of a ResnetBlock2D accumulated into the same output tile.
struct ConvSeg {
    const void *src;  // NHWC 16-bit, [B, Hs, Ws, C]
    int C;            // channels (multiple of 64)
    int taps;         // 9 or 1
    int up;           // 1: src is at half the output resolution (nearest-2x Upsample2D fused)
};

struct ConvArgs {
    ConvSeg seg[CONV_MAX_SEG];
    int nseg;
    const void *Wgt;     // [Cout_pad][Ktot] 16-bit, k ordered segment -> tap -> channel
    const float *bias;   // [Cout] or nullptr
    const float *temb;   // fp32 [*, temb_stride]; adds temb[b*temb_bstride + temb_off + co]
    int temb_bstride, temb_off;
    const void *resid;   // NHWC 16-bit [M, Cout] or nullptr
    void *out;           // layout per epilogue
    int B, H, W;         // output spatial size (powers of two)
    int stride;          // 1, or 2 for Downsample2D (segments are then [B, 2H, 2W, C])
    int Cout;            // real output channels
    int Ktot;
    int splitk;          // >1: fp32 partial slabs [splitk][M][Cout] into `out`, no bias
    const void *zeros;   // >= 16 bytes of zeros (source of padding taps)
    const void *steps;   // device copy of build_conv_steps(...) or nullptr (generic addressing)
    unsigned *counters;  // profiling aid (BNDM_IGEMM_TRACE): s_memtime marks of workgroup (0, 0), else nullptr
    int wtiled;          // 1: Wgt is tile-contiguous and pre-swizzled, [Cout/128][K/64][128 rows][8 slots][8]: slot j of
                         // row r holds k-group j ^ ((r >> 1) & 7) of the step -- one linear 16 KiB read per K-step
    int wmajor;          // 1: consecutive tiles (same XCD, dispatched together) share the WEIGHT panel (same n-tile,
                         // neighbouring m-tiles) -- for layers whose weights outweigh their activations (<= 8x8)
};

enum ConvEpilogue { EPI_NHWC16 = 0, EPI_F32_ROWS = 1, EPI_NCHW32 = 2 };
enum ConvTile { TILE_128x128 = 0, TILE_128x32 = 1, TILE_256x128 = 2 };
int conv_tile_bm(int tile);

int launch_conv(int dtype, int tile, int epi, const ConvArgs &a, hipStream_t st);
std::vector<int> build_conv_steps(const ConvSeg *seg, int nseg, int W_out, int stride);

// sum split-K slabs + bias + temb + residual -> NHWC 16-bit
int launch_splitk_reduce(int dtype, const float *part, int splitk, const ConvArgs &a, hipStream_t st);

// conv_in: fp32 NCHW sample (+ optional extra fp32 NCHW tensor concatenated on channels) -> NHWC 16-bit
// W16 is [C0][KP] 16-bit with k = ci*9 + ky*3 + kx, zero-padded to KP (multiple of 16, <= 64)
// stats (optional): GroupNorm partial sums [B][H*W/128][C0][2] of the stored values
int launch_conv_in(int dtype, const float *x, int Cx, const float *extra, int Ce, const void *W16,
                   const float *bias, void *out, float *stats, int B, int H, int W, int C0, int KP, hipStream_t st);

// GroupNorm(32) statistics of cat(x1, x2) -> per-(sample, channel) scale/shift
//   y = x * scale[b][c] + shift[b][c]  ==  (x - mean_g) * rstd_g * gamma_c + beta_c
int launch_gn_stats(int dtype, const void *x1, int C1, const void *x2, int C2, int B, int HW, float *partial,
                    int nslab, hipStream_t st);
int launch_gn_finalize(const float *partial, int nslab, int B, int HW, int C, int groups, float eps,
                       const float *gamma, const float *beta, float *scale_shift /*[B][2][C]*/, hipStream_t st);
int launch_gn_apply(int dtype, const void *x1, int C1, const void *x2, int C2, const float *scale_shift, int B,
                    int HW, int silu, void *out, hipStream_t st);

// softmax(q k^T / sqrt(8)) v per (sample, head of 8 channels); qkv [B*T][3C] -> out [B*T][C]
int launch_attention(int dtype, const void *qkv, void *out, int B, int T, int C, hipStream_t st);
// in-place row softmax of a 16-bit [rows][n] matrix: p = softmax(scale * s) with fp32 maths (n % 8 == 0, n <= 4096)
int launch_softmax_rows(int dtype, void *s, int rows, int n, float scale, hipStream_t st);
// 1x1 conv on a tiny fp32 NCHW tensor (AutoencoderKL.post_quant_conv): out[b][m][p] = bias[m] + sum_c w[m][c]*z[b][c][p]
int launch_pointwise_f32(const float *z, const float *w, const float *bias, float *out, int B, int Cin, int Cout, int HW,
                         hipStream_t st);

// Timesteps(128, flip_sin_to_cos) -> Linear -> SiLU -> Linear -> SiLU, fp32; writes 16-bit [B][D]
int launch_temb_mlp(int dtype, const float *t, int B, int C0, int D, const float *W1t /*[C0][D]*/, const float *b1,
                    const float *W2t /*[D][D]*/, const float *b2, void *act_emb16, hipStream_t st);

int gn_num_slabs(int HW);

}  // namespace bndm

// ------------------------------------------------------------------------------------------------
// Fused 3x3 convolution (stride 1) for the high-resolution, FLOP-dominant layers: conv_t32 (unet_conv32.hip).
//   input side : GroupNorm scale/shift (+SiLU) applied in place to the (TH+2)x18 input halo patch of a 32-channel
//                chunk after it has been brought into LDS -- once per element, not once per tap;
//   main loop  : 9 taps read MFMA fragments from the same LDS patch at shifted pixel offsets, weight tiles streamed
//                by LDS-DMA through a 4-slot ring;
//   epilogue   : bias / time embedding + residual, tile staged through LDS for full-row stores, and
//                per-(sample, tile, channel) sum / sum-of-squares of the stored values for the next GroupNorm.
// ------------------------------------------------------------------------------------------------
namespace bndm {

struct FusedSeg {
    const void *src;  // NHWC 16-bit [B, Hs, Ws, C]
    int C;            // multiple of 64
    int taps;         // 9 (3x3, pad 1) or 1 (1x1 conv_shortcut on the raw tensor)
    int up;           // source at half resolution (nearest-2x upsample)
    int ss_off;       // channel offset into the scale/shift table, or -1: no normalisation
};

struct FusedArgs {
    FusedSeg seg[CONV_MAX_SEG];
    int nseg;
    const float *ss;     // [B][2][ssC] scale / shift, nullptr when no segment is normalised
    int ssC;
    int silu;
    const void *Wgt;     // pack_weights_t32: [Cout/128][K-step][128 rows][32 k], pre-swizzled
    int Ktot;
    const float *bias;
    const float *temb;
    int temb_bstride, temb_off;
    const void *resid;
    void *out;           // NHWC 16-bit; fp32 NCHW [B][Cout][H][W] when out_nchw32
    int out_nchw32;      // 1: network head (Cout <= 32, no residual / statistics / time embedding)
    int nco;             // output channels per workgroup tile (128)
    float *stats;        // [B][tiles_per_sample][Cout][2] or nullptr
    int B, H, W, Cout;
    const void *zeros;   // >= 16 bytes of zeros
    // In-kernel GroupNorm finalisation (gn_p1 != nullptr; `ss` is then ignored): every workgroup turns the per-tile
    // partial sums of its sample into the scale / shift table itself, in its prologue -- no gn_finalize2 launch.
    // gn_p1 / gn_p2: [B][gn_ns1 / gn_ns2][gn_C1 / ssC - gn_C1][2] (sum, sum of squares); at most 32 slabs each.
    const float *gn_p1, *gn_p2, *gn_gamma, *gn_beta;
    int gn_ns1, gn_ns2, gn_C1, gn_HW;
    float gn_eps;
};

// TH = 16 (256-pixel tiles) or 8 (128-pixel tiles); W must be a multiple of 16, H of TH
// conv_t32 (unet_conv32.hip): 256-thread workgroups, two per CU, 32-channel K-steps, tile-contiguous weights
bool conv_t32_supports(const FusedArgs &a);
int launch_conv_t32(int dtype, int TH, const FusedArgs &a, hipStream_t st);
int conv_t32_tiles_per_sample(int TH, int H, int W);
// w_of(segment, out channel, channel within the segment, tap) -> fp32 weight; result is [n-tile][K-step][128][32]
// rows: output channels per n-tile (128, or 32 for the head)
std::vector<float> pack_weights_t32(const FusedSeg *seg, int nseg, int Cout,
                                    const std::function<float(int, int, int, int)> &w_of, int rows = 128);
// one-launch GroupNorm(+SiLU) for small per-sample tensors (statistics + apply, one block per sample)
// x1 given as split-K partial sums: element = round16(sum_z part[z] + bias + temb + resid); the rounded value is also
// stored to raw_out (the tensor splitk_reduce would have produced, bit-identical)
struct GnSlabSrc {
    const float *part = nullptr;     // [splitk][B*HW][C1] fp32 slabs, nullptr: x1 is a plain 16-bit tensor
    int splitk = 0;
    const float *bias = nullptr;     // [C1]
    const float *temb = nullptr;     // + b * temb_bstride + temb_off + c
    int temb_bstride = 0, temb_off = 0;
    const void *resid = nullptr;     // 16-bit [B*HW][C1]
    void *raw_out = nullptr;         // 16-bit [B*HW][C1]
};
int launch_gn_small(int dtype, const void *x1, int C1, const void *x2, int C2, int B, int HW, int groups, float eps,
                    const float *gamma, const float *beta, int silu, void *out, hipStream_t st,
                    const GnSlabSrc *slab = nullptr);

// two-source variant of gn_finalize: statistics of cat(x1, x2) from per-tensor partial sums
int launch_gn_finalize2(const float *p1, int nslab1, int C1, const float *p2, int nslab2, int C2, int B, int HW,
                        int groups, float eps, const float *gamma, const float *beta, float *scale_shift,
                        hipStream_t st);

}  // namespace bndm

// ------------------------------------------------------------------------------------------------
// conv_s (unet_tail.hip): one launch per convolution of the <= 8x8 levels.  A workgroup owns whole samples x 32 (96)
// output channels over the full K range (K split across its 8 waves), writes the raw tensor and the GroupNorm(+SiLU)
// normalised tensor of every consumer, or -- for the attention blocks -- softmax(q k^T / sqrt(8)) v of its four heads.
// ------------------------------------------------------------------------------------------------
namespace bndm {

struct TailRound {       // device table entry (8 dwords), one per patch round
    const void *src;     // NHWC 16-bit source tensor
    int row_bytes;       // channels * 2 of the source
    int cbyte;           // first channel of the round * 2
    int mode;            // 0: same resolution, 1: source at half resolution (nearest-2x), 2: source at double resolution
    int phase;           // mode 2: 2 * py + px
    int nsub;            // 32-channel sub-chunks in the round (1..8)
    int pad;
};

struct TailNorm {        // one consuming GroupNorm of the produced tensor
    void *out;           // normalised (+SiLU) 16-bit tensor [M][Cout]
    const float *gamma, *beta;   // the consumer's affine parameters for THIS tensor's channels
    int gs;              // channels per group (8, 16 or 32)
    int silu;
};

enum TailEpilogue { TAIL_EPI_CONV = 0, TAIL_EPI_ATTN = 1 };

struct TailArgs {
    const void *wgt;            // weight stream (build_tail_plan)
    const uint32_t *desc;       // [8 waves][maxsteps]
    const TailRound *rounds;    // [nrounds]
    TailRound r0, r1;           // copies of rounds[0], rounds[1] (r1 unused when nrounds == 1)
    int nrounds, maxsteps;
    int nuse[8];                // entries of each wave's list up to its last real / boundary entry
    long long tile_bytes;       // weight stream bytes per n-tile
    int wave_bytes;             // ... per wave
    int B, hwlog, wlog;         // H * W = 1 << hwlog (4, 16, 64), W = 1 << wlog
    int Cout, ntn;              // channels per row of the output tensors; n-tiles
    const float *bias;
    const float *temb;
    int temb_bstride, temb_off;
    const void *resid;          // [M][Cout] 16-bit or nullptr
    void *raw_out;              // [M][Cout] 16-bit or nullptr
    int nreq;
    TailNorm req[3];
    float eps;
    int epi;
    void *attn_out;             // TAIL_EPI_ATTN: [M][Cout] 16-bit attention output (before to_out)
    unsigned long long *dbg;    // profiling aid: s_memtime marks of workgroup 0 (nullptr in the product path)
};

enum TailSegKind { TAIL_SEG_3x3 = 0, TAIL_SEG_3x3_UP = 1, TAIL_SEG_3x3_S2 = 2, TAIL_SEG_1x1 = 3 };
struct TailSeg {
    int kind;
    int C;               // channels of the source tensor (multiple of 32)
};
struct TailPlanRound {
    int seg, c0, nsub, mode, phase;
};
struct TailPlan {
    int nrounds = 0, maxsteps = 0, ntn = 0;
    int nuse[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    size_t wave_elems = 0, tile_elems = 0;
    std::vector<TailPlanRound> rounds;
    std::vector<uint32_t> desc;
    std::vector<float> wgt;          // to be converted to 16 bits
};
// w_of(segment, weight row, channel within the segment, tap 0..8 (0 for 1x1)) -> fp32 weight;
// row_of(n-tile, 32-row block, row in block) -> weight row (output channel), or -1 past the end
TailPlan build_tail_plan(const std::vector<TailSeg> &segs, int Cout_rows, int NB, int D,
                         const std::function<float(int, int, int, int)> &w_of,
                         const std::function<int(int, int, int)> &row_of);
int tail_ring_depth(int nb);
// TM x (32 NB): 128 x 32, 64 x 32, 64 x 96
int launch_conv_tail(int dtype, int TM, int NB, const TailArgs &a, hipStream_t st);

}  // namespace bndm
