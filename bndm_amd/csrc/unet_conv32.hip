// conv_t32: the fused GroupNorm+SiLU 3x3 convolution of the FLOP-dominant UNet layers (H >= 16), built so that TWO
// workgroups are resident per CU.
//
// One workgroup = 256 threads = 4 waves (one per SIMD) and at most 80 KiB of LDS; it computes a TH x 16 pixel tile of
// one sample x 128 output channels.  Wave w owns TH/4 pixel rows of the tile and ALL 128 channels (64 x 128 wave tile:
// 4 x 2 MFMA tiles of 32x32, six ds_read_b128 per eight MFMAs).  K runs over 32-channel chunks of up to 4 segments
// (concatenated inputs, nearest-2x upsampled inputs, a fused 1x1 conv_shortcut): per chunk the (TH+2) x 18 halo patch is
// brought in ONCE by LDS-DMA (64 B per pixel), normalised in place (GroupNorm scale/shift + SiLU) by the thread that
// issued each piece, and read by all nine taps; the [128][32] weight tile of every tap streams through a 4-slot ring
// three K-steps ahead.  Weights are packed tile-contiguous and pre-swizzled on the host (pack_weights_t32), so a K-step's
// tile is one linear 8 KiB read.
//
// Why two workgroups per CU: the launch-wide bursts of one-workgroup-per-CU kernels (every CU in its prologue, then
// every CU in its epilogue) leave the MFMA pipe idle for a third of each launch, and inside the loop all waves of a
// workgroup issue their LDS-DMA right behind the same barrier.  With two independent workgroups per CU -- one wave of
// each on every SIMD -- one's prologue / epilogue / DMA issue / barrier wait is the other's MFMA time.
//
// LDS patch layout: pixel p = py * 18 + px at byte 64 p, its four 16-byte channel groups XOR-ed with a 2-bit key
// g(py & 1, px) found by search (t32_patch_key) such that every 16-lane group of every ds_read_b128 fragment read hits
// 16 distinct bank quads for all nine tap shifts.  LDS-DMA destinations are lane-linear, so the permutation sits on the
// source address (which channel group a lane fetches) and on the read address.
#include "unet_kernels.hpp"
#include "unet_types.hpp"
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

namespace bndm {

// key of patch pixel (py, px): 2 bits per px, one 36-bit word per row parity
__host__ __device__ constexpr int t32_patch_key(int py, int px) {
    return (int)((((py & 1) ? 0x2479cc7f2ull : 0x712c992a7ull) >> (2 * px)) & 3ull);
}

namespace {

template <int N> struct IC {
    static constexpr int value = N;
};

// patch DMA rounds issued after the barrier of tap t: ceil(nround / 3) per tap over taps 0..2; t is taken modulo 9
constexpr int rounds_per_tap(int nround) { return (nround + 2) / 3; }
constexpr int rounds_at_tap(int nround, int t) {
    t = ((t % 9) + 9) % 9;
    const int rpt = rounds_per_tap(nround), left = nround - rpt * t;
    return t > 2 ? 0 : (left >= rpt ? rpt : (left > 0 ? left : 0));
}
// DMAs a thread has issued after patch round r by the end of group G_s; nwp = weight pieces per thread and tile
constexpr int dmas_after_round(int nround, int nwp, int r, int s) {
    const int rpt = rounds_per_tap(nround), g = r / rpt;
    int n = (rpt * g + rpt - 1 < nround - 1 ? rpt * g + rpt - 1 : nround - 1) - r;
    for (int j = g + 1; j <= s; ++j) n += nwp + rounds_at_tap(nround, j);
    return n;
}

typedef uint32_t u32x4t __attribute__((ext_vector_type(4)));

constexpr int T32_MAX_CHUNKS = 32;
constexpr int T32_SS_BYTES = 4096;           // scale/shift table [2][ssC] fp32, ssC <= 512
constexpr float T32_LOG2E = 1.4426950408889634f;

// wave-uniform description of one 32-channel chunk of a segment
struct Chunk {
    const void *src;             // the segment's tensor
    int bytes;                   // ... and its size
    int soff;                    // chunk * 64 bytes
    int C2;                      // bytes per source pixel
    int up;                      // source at half resolution
    int ssbase;                  // float offset of the chunk's scale row in the LDS table, or -1
};

// ABL: profiling switches (results are wrong when non-zero): 1 no MFMA, 2 no weight DMA, 4 no fragment reads,
// 8 no in-loop patch DMA / normalisation, 16 no in-loop normalisation (DMA kept), 128 / 256 prologue without its
// patch / weight DMAs, 64 record s_memtime marks of
// block 0 / chunk 1 into dbg[wave][tap][6]
// NW: waves per workgroup.  4: one wave per SIMD and workgroup, 64 x 128 wave tiles, two workgroups per CU -- for grids
// of at least two workgroups per CU.  8: 4(M) x 2(N) waves of 64 x 64, two waves per SIMD from ONE workgroup -- for
// the smaller grids (32x32 and 16x16 layers at batch 64), where a CU would otherwise host a single 4-wave workgroup.
// NCO: output channels per workgroup.  128: the ResnetBlock2D / Upsample2D convolutions (16-bit NHWC output + GroupNorm
// partial sums).  32: the network head -- conv_norm_out + SiLU + conv_out (iadb_bn.py:205-282: out_channels 3 / 6 / 8),
// written as the caller's fp32 NCHW tensor; the loop is then bound by the one read + normalisation of the input.
// SS: super-step schedule of the 8-wave variant (one workgroup per CU): one barrier per THREE taps, a weight ring of nine
// tiles (= one chunk: tap t lives in slot t) filled two super-steps ahead, three patch buffers (the chunk after next is
// DMA'd while the next one is normalised).  Same MFMA order as the per-tap schedule, i.e. bit-identical results.
template <typename T, int TH, int ABL, int NW, int NCO, int SS = 0>
__global__ __launch_bounds__(NW * 64, SS ? 1 : 2) void conv_t32(const FusedArgs a, const int tiles_x, const int tps, const int ntn,
                                                   const int nsteps_w, unsigned *__restrict__ dbg) {
    using v8 = typename TT<T>::v8;
    using v4 = typename TT<T>::v4;
    using v2 = typename TT<T>::v2;
    constexpr int TW = 16, PW = TW + 2, PH = TH + 2;
    constexpr int NT = NW * 64, WAVES_N = NW / 4;
    constexpr int NPIECE = PH * PW * 4;                // 16-byte pieces per patch chunk
    constexpr int NWP = (NCO * 64) / (NT * 16) > 0 ? (NCO * 64) / (NT * 16) : 1;   // weight-tile DMAs per thread (2 or 1)
    constexpr int NWW = NCO * 64 / 1024 < NW ? NCO * 64 / 1024 : NW;              // waves that carry weight pieces
    constexpr int NROUND = (NPIECE + NT - 1) / NT;     // patch DMA rounds per chunk (the last one may be partial)
    constexpr int NREMW = (NPIECE - (NROUND - 1) * NT + 63) / 64;    // waves with pieces in the last round
    constexpr int PATCH_BYTES = (NROUND - 1) * NT * 16 + NREMW * 1024;
    constexpr int BM = TH * TW;
    constexpr int TM = TH / 8;                         // 32-pixel MFMA tiles per wave along M (2 rows x 16 columns each)
    constexpr int TN = NCO / 32 / WAVES_N;             // 32-channel MFMA tiles per wave along N
    static_assert(TN >= 1, "wave tiling");
    static_assert(!SS || (NW == 8 && NCO == 128), "super-steps: 8-wave, 128-channel variant only");
    constexpr int WSTAGES = SS ? 9 : 4, W_BYTES = NCO * 64, NPB = SS ? 3 : 2;
    constexpr int OFF_W = NPB * PATCH_BYTES;
    constexpr int OFF_SS = OFF_W + WSTAGES * W_BYTES;
    constexpr int OFF_TAB = OFF_SS + T32_SS_BYTES;     // chunk descriptors, 16 B each
    constexpr int OFF_BIAS = OFF_TAB + T32_MAX_CHUNKS * 16;   // 128 fp32: bias (+ time embedding) row of this tile
    constexpr int OFF_DUMP = OFF_BIAS + 512;                  // dead slot: DMAs of waves / lanes without a piece
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l = tid & 63;
    auto life = [&](int k) {                           // profiling aid: lifetime marks of two workgroups
        if constexpr ((ABL & 64) != 0) {
            if ((blockIdx.x == 0 || blockIdx.x == 700) && a.Ktot == 2304) {
                const unsigned tm = (unsigned)__builtin_amdgcn_s_memtime();
                if (tid == 0) dbg[216 + (blockIdx.x ? 16 : 0) + k] = tm;
            }
        }
    };
    life(0);

    // ---- tile id (XCD-aware: neighbouring tiles of a sample share halos and weights) ---------------
    int tix;
    {
        const int nblk = gridDim.x, bid = blockIdx.x;
        const int xcd = bid & 7, q = nblk >> 3, r = nblk & 7;
        tix = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    tix = __builtin_amdgcn_readfirstlane(tix);
    const int mt = tix / ntn, nt = tix - mt * ntn;
    const int b = __builtin_amdgcn_readfirstlane(mt / tps), tin = __builtin_amdgcn_readfirstlane(mt - b * tps);
    const int ty = tin / tiles_x, tx = tin - ty * tiles_x;
    const int y0 = __builtin_amdgcn_readfirstlane(ty * TH), x0 = __builtin_amdgcn_readfirstlane(tx * TW);
    const int n0 = __builtin_amdgcn_readfirstlane(nt * NCO);
    const int H = a.H, Wd = a.W;

    // ---- chunk descriptors of the whole K loop (3x3 chunks first, then the 1x1 ones), built once into LDS ----------
    int nchunk9 = 0, nchunk1 = 0;           // 3x3 segments come first (checked by the launcher)
#pragma unroll
    for (int i = 0; i < CONV_MAX_SEG; ++i)
        if (i < a.nseg) {
            if (a.seg[i].taps == 9) nchunk9 += a.seg[i].C >> 5;
            else nchunk1 += a.seg[i].C >> 5;
        }
    if (tid < nchunk9 + nchunk1) {
        // every segment's fields as SCALARS first, then per-lane selects of values: a select between the kernel-argument
        // structs themselves becomes a vector load through a selected pointer, i.e. a cold global round trip at the head
        // of every launch
        int rem = tid;
        bool found = false;
        u32x4t e = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int i = 0; i < CONV_MAX_SEG; ++i) {
            const bool on = i < a.nseg;
            const uint64_t u = (uint64_t)a.seg[i].src;
            const uint32_t ulo = __builtin_amdgcn_readfirstlane((uint32_t)u), uhi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
            const int Ci = __builtin_amdgcn_readfirstlane(on ? a.seg[i].C : 0);
            const uint32_t upi = __builtin_amdgcn_readfirstlane((uint32_t)a.seg[i].up);
            const int ssi = __builtin_amdgcn_readfirstlane(a.seg[i].ss_off);
            const int nci = Ci >> 5;
            const bool hit = !found && rem < nci;
            e[0] = hit ? ulo : e[0];
            e[1] = hit ? uhi : e[1];
            e[2] = hit ? (uint32_t)(Ci * 2) | ((uint32_t)(rem * 64) << 12) | (upi << 31) : e[2];
            e[3] = hit ? (uint32_t)(ssi >= 0 ? ssi + rem * 32 : -1) : e[3];
            found = found || hit;
            rem -= nci;
        }
        *reinterpret_cast<u32x4t *>(smem + OFF_TAB + tid * 16) = e;
    }
    auto load_chunk = [&](int n) {
        const u32x4t e = *reinterpret_cast<const u32x4t *>(smem + OFF_TAB + n * 16);
        Chunk c;
        const uint32_t plo = __builtin_amdgcn_readfirstlane(e[0]), phi = __builtin_amdgcn_readfirstlane(e[1]);
        c.src = (const void *)(((uint64_t)phi << 32) | plo);
        const uint32_t m = __builtin_amdgcn_readfirstlane(e[2]);
        c.C2 = m & 0xfff;
        c.soff = (m >> 12) & 0xfff;
        c.up = m >> 31;
        c.ssbase = __builtin_amdgcn_readfirstlane(e[3]);
        c.bytes = a.B * (c.up ? (H >> 1) * (Wd >> 1) : H * Wd) * c.C2;
        return c;
    };

    // ---- patch piece descriptors (independent of the chunk) ----------------------------------------
    // piece = round * 256 + tid = (patch pixel, physical 16-byte slot); slot j of pixel (py, px) holds channel group
    // j ^ key(py, px)
    const bool src_up = a.seg[0].up != 0;              // nearest-2x sources: all segments of a launch or none (conv_t32_supports)
    int p_full[NROUND];                                // source pixel index of the piece, -1: padding
    int p_pack = 0;                                    // per round: bits 3r, 3r+1 source channel group, bit 3r+2 valid
#pragma unroll
    for (int r = 0; r < NROUND; ++r) {
        const int piece = r * NT + tid;
        const int pc = piece < NPIECE ? piece : NPIECE - 1;
        const int pp = pc >> 2, pch = pc & 3;
        const int pyy = pp / PW, pxx = pp - pyy * PW;
        const int iy = y0 - 1 + pyy, ix = x0 - 1 + pxx;
        const bool ok = piece < NPIECE && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)Wd;
        // (a half-resolution source is indexed at its own resolution here, once, instead of per chunk in the K loop)
        const int pix = src_up ? (b * (H >> 1) + (iy >> 1)) * (Wd >> 1) + (ix >> 1) : (b * H + iy) * Wd + ix;
        p_full[r] = ok ? pix : -1;                                         // -1: out-of-range offset -> zeros
        p_pack |= ((pch ^ t32_patch_key(pyy, pxx)) | (ok ? 4 : 0)) << (3 * r);
    }
    struct Piece {
        int pix, lc;
        bool valid;
    };
    auto piece_of = [&](int r) { return Piece{p_full[r], (p_pack >> (3 * r)) & 3, ((p_pack >> (3 * r + 2)) & 1) != 0}; };
    auto patch_dma = [&](auto rc, const Chunk &c, int buf) {
        constexpr int r = decltype(rc)::value;
        const Piece pc = piece_of(r);
        const int pix = pc.pix;
        const unsigned voff = (unsigned)(pix * c.C2 + (pc.lc << 4));
        char *dst = smem + ((r < NROUND - 1 || w < NREMW) ? buf * PATCH_BYTES + r * (NT * 16) + w * 1024 : OFF_DUMP);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(uniform_rsrc(c.src, c.bytes), (lds_ptr_t)dst, 16, voff,
                                                 __builtin_amdgcn_readfirstlane(c.soff), 0, 0);
    };
    // GroupNorm scale/shift + SiLU applied in place to this thread's own piece of a round (the reference pads
    // AFTER the activation: padding / tail pieces are rewritten unchanged, i.e. stay zero)
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const float *ssL = reinterpret_cast<const float *>(smem + OFF_SS);
    auto xf_begin = [&](auto rc, int buf, u32x4 &x) {
        constexpr int r = decltype(rc)::value;
        x = *reinterpret_cast<const u32x4 *>(smem + buf * PATCH_BYTES + r * (NT * 16) + tid * 16);
    };
    // (padding / tail pieces are not written back: they stay the zeros the DMA left.  The select is on the address -- one
    // v_cndmask per piece instead of one per dword -- and the losers land in the dead slot)
    auto xf_end = [&](auto rc, int buf, const u32x4 &x) {
        constexpr int r = decltype(rc)::value;
        const int real = buf * PATCH_BYTES + r * (NT * 16) + tid * 16, dead = OFF_DUMP + l * 16;
        *reinterpret_cast<u32x4 *>(smem + (piece_of(r).valid ? real : dead)) = x;
    };
    // Two elements (one dword) of GroupNorm scale/shift + SiLU.  The table holds log2(e) * (scale, shift), so
    //     y' = s' x + h' = log2(e) y,   o' = y' / (1 + 2^-y') = log2(e) silu(y),
    // and the weights of a normalised segment are packed multiplied by ln 2 (conv_fused): the multiplication by -log2(e)
    // in front of the exponential is gone (the negation is a source modifier).  This file is compiled with
    // -fno-slp-vectorize: the SLP vectoriser pairs the two elements into v_pk_fma_f32 / v_pk_mul_f32 (half-rate beside
    // MFMAs, + wait states) and thereby blocks v_fma_mix_f32, which reads the f16 halves directly -- 7 plain + 4
    // transcendental issues per dword instead of 13 + 4.
    auto norm2 = [&](unsigned xq, float s0, float s1, float h0, float h1) -> unsigned {
        const v2 in = __builtin_bit_cast(v2, xq);
        v2 o;
        const float f0 = fmaf((float)in[0], s0, h0), f1 = fmaf((float)in[1], s1, h1);
        o[0] = (T)(f0 * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-f0)));
        o[1] = (T)(f1 * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-f1)));
        return __builtin_bit_cast(unsigned, o);
    };
    // half h (0 / 1) of round r of chunk c: dwords 2h, 2h+1 of the piece = channels 4h .. 4h+3 of its group
    auto xf_half = [&](auto rc, auto hc, const Chunk &c, u32x4 &x) {
        constexpr int r = decltype(rc)::value, h = decltype(hc)::value;
        const Piece pc = piece_of(r);
        const float *sc = ssL + c.ssbase + pc.lc * 8 + 4 * h;
        const f32x4 s4 = *reinterpret_cast<const f32x4 *>(sc), h4 = *reinterpret_cast<const f32x4 *>(sc + a.ssC);
        x[2 * h] = norm2(x[2 * h], s4[0], s4[1], h4[0], h4[1]);
        x[2 * h + 1] = norm2(x[2 * h + 1], s4[2], s4[3], h4[2], h4[3]);
    };
    auto xf_owner = [&](int r) { return r < NROUND - 1 || w < NREMW; };    // wave-uniform

    // ---- weight tiles: [128 rows][64 B] per K-step, packed per (n-tile, step) as one linear 8 KiB block ----------
    const __amdgpu_buffer_rsrc_t wrs =
        uniform_rsrc((const char *)a.Wgt + (size_t)nt * nsteps_w * W_BYTES, nsteps_w * W_BYTES);
    auto w_issue1 = [&](int slot, int step, int i) {   // piece i (< NWP) of a tile
        char *base = smem + (w < NWW ? OFF_W + slot * W_BYTES + w * 1024 : OFF_DUMP);
        const int so = __builtin_amdgcn_readfirstlane(step * W_BYTES);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lds_ptr_t)(base + i * (NT * 16)), 16,
                                                 w < NWW ? (unsigned)(tid * 16 + i * (NT * 16)) : 0x80000000u, so, 0, 0);
    };
    auto w_issue = [&](int slot, int step) {
        char *base = smem + (w < NWW ? OFF_W + slot * W_BYTES + w * 1024 : OFF_DUMP);   // (narrow tiles: the other waves' DMA
        const int so = __builtin_amdgcn_readfirstlane(step * W_BYTES);                  //  reads past the tile = zeros, dumped)
#pragma unroll
        for (int i = 0; i < NWP; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lds_ptr_t)(base + i * (NT * 16)), 16,
                                                     w < NWW ? (unsigned)(tid * 16 + i * (NT * 16)) : 0x80000000u, so, 0, 0);
    };

    f32x16 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int q = l & 31, kh = l >> 5;
    const int wm = w & 3, wn = w >> 2;                 // wave's pixel-row group / channel half
    const int row_base = wm * (TH / 4);
    const int lr = q >> 4, lcx = q & 15;

    // ---- fragment addresses ---------------------------------------------------------------------------
    // weights: row q of the tile (+ 32 i), slot (2 ks + kh) ^ ((q >> 2) & 3).  patch: pixel (row_base + 2j + lr + ky,
    // lcx + kx), slot (2 ks + kh) ^ key(row parity, lcx + kx); ky = 0 and 2 share the parity, so they differ by an
    // immediate.
    // The second k16 slice (ks = 1) flips bit 1 of the slot index, i.e. bit 5 of the byte address: one v_xor per
    // fragment base instead of a second set of address registers.
    int wa, pa[2][3];
    wa = OFF_W + (wn * TN * 32 + q) * 64 + ((kh ^ ((q >> 2) & 3)) << 4);
#pragma unroll
    for (int kyp = 0; kyp < 2; ++kyp)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
            pa[kyp][kx] = ((row_base + lr + kyp) * PW + lcx + kx) * 64 +
                          ((kh ^ t32_patch_key(row_base + lr + kyp, lcx + kx)) << 4);
    // Weight fragments live in ONE register set that is refilled in place: fa[i] is read again (for the next phase) right
    // behind the MFMAs that consumed it, six or more MFMAs before its next use.  Patch fragments are used by every
    // MFMA of a phase and are double-buffered.
    v8 fa[TN], fb[2][TM];
    auto read_b = [&](auto tc, auto kc) {
        constexpr int t = decltype(tc)::value, ks = decltype(kc)::value;
        constexpr int ky = t / 3, kx = t % 3;
        if (ABL & 4) return;
        const int pb = pa[ky & 1][kx] ^ (ks << 5);
#pragma unroll
        for (int j = 0; j < TM; ++j)
            fb[ks][j] = *reinterpret_cast<const v8 *>(smem + pb + ((ky & 2) * PW + 2 * j * PW) * 64);
    };
    auto read_a = [&](auto tc, auto kc, int i) {     // (SS: tap t of a chunk lives in ring slot t)
        constexpr int ks = decltype(kc)::value, t = decltype(tc)::value;
        if (ABL & 4) return;
        fa[i] = *reinterpret_cast<const v8 *>(smem + (wa ^ (ks << 5)) + i * 2048 + (SS ? t * W_BYTES : 0));
    };
    // one phase: the MFMAs of k16 slice `kcur` (operands in fa, fb[kcur]) and the reads of slice (tnext, knext)
    auto mma_refill = [&](auto kcur, auto tnext, auto knext) {
        constexpr int ks = decltype(kcur)::value;
        read_b(tnext, knext);
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            if (ABL & 1) {
                asm volatile("" ::"v"(fa[i]));
#pragma unroll
                for (int j = 0; j < TM; ++j) asm volatile("" ::"v"(fb[ks][j]));
            } else {
#pragma unroll
                for (int j = 0; j < TM; ++j) acc[i][j] = TT<T>::mfma(fa[i], fb[ks][j], acc[i][j]);
            }
            read_a(tnext, knext, i);
        }
    };

    // One phase in PINNED order (scheduling fences instead of sched_group_barrier patterns, whose greedy solver bunches
    // the normalisation arithmetic at the head of a phase as soon as its instruction mix changes):
    //     [patch fragment reads]  { [MFMAs of weight tile i] | [its refill read, extra(i)] } x TN
    // extra(i): the i-th LDS-DMA of the phase and stage i of the normalisation, supplied by the caller.
    auto phase_pinned = [&](auto kcur, auto tnext, auto knext, auto &&extra) {
        constexpr int ks = decltype(kcur)::value;
        read_b(tnext, knext);
        auto grp = [&](auto self, auto ic) {
            constexpr int i = decltype(ic)::value;
            if constexpr (i < TN) {
                if (ABL & 1) {
                    asm volatile("" ::"v"(fa[i]));
#pragma unroll
                    for (int j = 0; j < TM; ++j) asm volatile("" ::"v"(fb[ks][j]));
                } else {
#pragma unroll
                    for (int j = 0; j < TM; ++j) acc[i][j] = TT<T>::mfma(fa[i], fb[ks][j], acc[i][j]);
                }
                __builtin_amdgcn_sched_barrier(0);
                read_a(tnext, knext, i);
                extra(ic);
                __builtin_amdgcn_sched_barrier(0);
                self(self, IC<i + 1>{});
            }
        };
        grp(grp, IC<0>{});
    };
    // normalisation of two dwords (four elements) in three stages that ride behind successive MFMA groups: every stage
    // is four independent chains, and a transcendental's consumer sits a whole MFMA group behind it
    constexpr int STG_B = TN >= 2 ? 1 : 0, STG_C = TN >= 3 ? 2 : TN - 1;
    auto norm_stage = [&](auto ic, auto pc, u32x4 &xv, const f32x4 &sv, const f32x4 &hv, float (&ny)[4], float (&ne)[4]) {
        constexpr int i = decltype(ic)::value, p = decltype(pc)::value;
        if constexpr (i == 0) {
            const unsigned x0 = xv[2 * p], x1 = xv[2 * p + 1];
            const v2 i0 = __builtin_bit_cast(v2, x0), i1 = __builtin_bit_cast(v2, x1);
            ny[0] = fmaf((float)i0[0], sv[0], hv[0]);
            ny[1] = fmaf((float)i0[1], sv[1], hv[1]);
            ny[2] = fmaf((float)i1[0], sv[2], hv[2]);
            ny[3] = fmaf((float)i1[1], sv[3], hv[3]);
#pragma unroll
            for (int k = 0; k < 4; ++k) ne[k] = __builtin_amdgcn_exp2f(-ny[k]);
        }
        if constexpr (i == STG_B) {
#pragma unroll
            for (int k = 0; k < 4; ++k) ne[k] = __builtin_amdgcn_rcpf(1.0f + ne[k]);
        }
        if constexpr (i == STG_C) {
            v2 o0, o1;
            o0[0] = (T)(ny[0] * ne[0]);
            o0[1] = (T)(ny[1] * ne[1]);
            o1[0] = (T)(ny[2] * ne[2]);
            o1[1] = (T)(ny[3] * ne[3]);
            xv[2 * p] = __builtin_bit_cast(unsigned, o0);
            xv[2 * p + 1] = __builtin_bit_cast(unsigned, o1);
        }
    };

    // ---- prologue -------------------------------------------------------------------------------------
    // scale/shift table (one 16-byte piece per thread, zeros past its end), patch of chunk 0, weight tiles of
    // steps 0..3 -- all by LDS-DMA, so they retire in issue order and one counted wait separates them
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                      // chunk table visible
    asm volatile("" ::: "memory");
    life(6);
    Chunk cur = load_chunk(0);
    Chunk nxt = cur;
    {
        if (!a.gn_p1) {
            const __amdgpu_buffer_rsrc_t srs =
                uniform_rsrc(a.ss ? a.ss + (size_t)b * 2 * a.ssC : (const float *)a.zeros, a.ss ? 2 * a.ssC * 4 : 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srs, (lds_ptr_t)(smem + (w < 4 ? OFF_SS + w * 1024 : OFF_DUMP)), 16,
                                                     (unsigned)(tid * 16), 0, 0, 0);
        }
        // the tile's additive row (the launcher passes bias OR the time-embedding row, which has the bias folded in):
        // 32 pieces by the first lanes of wave 0 (its other lanes and the other waves read zeros into the dead slot,
        // which starts right behind the row)
        const float *row = a.temb ? a.temb + (size_t)b * a.temb_bstride + a.temb_off + n0 : a.bias ? a.bias + n0 : nullptr;
        const __amdgpu_buffer_rsrc_t brs =
            uniform_rsrc(row ? row : (const float *)a.zeros, row ? (a.Cout - n0 < NCO ? a.Cout - n0 : NCO) * 4 : 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(brs, (lds_ptr_t)(smem + (w == 0 ? OFF_BIAS : OFF_DUMP)), 16,
                                                 (unsigned)(tid * 16), 0, 0, 0);
    }
    const int nstep9 = nchunk9 * 9;
    if (nchunk9 > 0) {
        auto issue_all = [&](auto self, auto rc) {
            constexpr int r = decltype(rc)::value;
            if constexpr (r < NROUND) {
                patch_dma(rc, cur, 0);
                self(self, IC<r + 1>{});
            }
        };
        if constexpr (!(ABL & 128)) issue_all(issue_all, IC<0>{});      // (ABL 128 / 256: prologue without its patch / weight DMAs)
        if constexpr (SS) {
            // group 1: patch of chunk 0 (above), tiles 0..2, round 0 of chunk 1; groups 2 and 3 have the shape of the
            // main loop's DMA batches (one patch round, three tiles), so that its counted waits hold from the first barrier
            nxt = load_chunk(nchunk9 > 1 ? 1 : 0);
            auto tile = [&](int t) { w_issue(t, t < nstep9 ? t : nstep9 - 1); };
            auto round = [&](auto rc) {
                if constexpr (decltype(rc)::value < NROUND) patch_dma(rc, nxt, 1);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lds_ptr_t)(smem + OFF_DUMP), 16, 0x80000000u, 0, 0, 0);
            };
            tile(0), tile(1), tile(2);
            round(IC<0>{});
            round(IC<1>{});
            tile(3), tile(4), tile(5);
            round(IC<2>{});
            tile(6), tile(7), tile(8);
        } else if constexpr (!(ABL & 256)) {
            w_issue(0, 0);
            w_issue(1, 1);
            w_issue(2, 2);
            w_issue(3, 3);
        }
        life(7);
        if (a.gn_p1) {
            // GroupNorm(32) statistics of cat(x1, x2) for this sample from the producers' per-tile partial sums (fixed slab
            // order, fp64 group combine: the arithmetic of gn_finalize2), while the DMAs above are in flight.  Scratch: the
            // second patch buffer, idle until the main loop.
            float *cs = reinterpret_cast<float *>(smem + (NPB - 1) * PATCH_BYTES), *css = cs + 512;
            const int C = a.ssC, C1 = a.gn_C1;
            float gam[2], bet[2];                        // C <= 512: at most two channels per thread; loaded up front
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int c = tid + k * NT;
                gam[k] = c < C ? a.gn_gamma[c] : 0.f;
                bet[k] = c < C ? a.gn_beta[c] : 0.f;
            }
            for (int c = tid; c < C; c += NT) {
                const bool first = c < C1;
                const float *p = first ? a.gn_p1 : a.gn_p2;
                const int ns = first ? a.gn_ns1 : a.gn_ns2, Cs = first ? C1 : C - C1, cc = first ? c : c - C1;
                float s = 0.f, q2 = 0.f;
                for (int k0 = 0; k0 < ns; k0 += 16) {        // 16 slabs in flight; the sums keep the slab order
                    float2 v[16];
#pragma unroll
                    for (int k = 0; k < 16; ++k)
                        v[k] = k0 + k < ns ? *reinterpret_cast<const float2 *>(p + ((size_t)(b * ns + k0 + k) * Cs + cc) * 2)
                                           : float2{0.f, 0.f};
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        s += v[k].x;
                        q2 += v[k].y;
                    }
                }
                cs[c] = s;
                css[c] = q2;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            life(8);
            const int Cg = C >> 5;
            float *ssW = reinterpret_cast<float *>(smem + OFF_SS);
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                const int c = tid + k2 * NT;
                if (c >= C) break;
                const int g0 = (c / Cg) * Cg;
                double s = 0, q2 = 0;
                for (int k = 0; k < Cg; ++k) {
                    s += cs[g0 + k];
                    q2 += css[g0 + k];
                }
                const double n = (double)Cg * a.gn_HW;
                const double mean = s / n;
                double var = q2 / n - mean * mean;
                var = var > 0 ? var : 0;
                const float rstd = (float)(1.0 / sqrt(var + (double)a.gn_eps));
                const float sc = rstd * gam[k2];
                ssW[c] = T32_LOG2E * sc;                                   // (norm2: the table carries log2(e))
                ssW[C + c] = T32_LOG2E * (bet[k2] - (float)mean * sc);
            }
        }
        life(9);
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(SS ? 8 : 4 * NWP) : "memory");   // table + own patch pieces landed
        if (!a.gn_p1 && w < 4) {                             // a table from memory (gn_finalize2): own piece times log2(e)
            f32x4 *tp = reinterpret_cast<f32x4 *>(smem + OFF_SS + tid * 16);
            *tp = *tp * T32_LOG2E;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();                         // ... in every wave (the table is shared)
        asm volatile("" ::: "memory");
        life(10);
        if (!(ABL & 8) && cur.ssbase >= 0) {
            auto xf_all = [&](auto self, auto rc) {
                constexpr int r = decltype(rc)::value;
                if constexpr (r < NROUND) {
                    if (xf_owner(r)) {
                        u32x4 x;
                        xf_begin(rc, 0, x);
                        xf_half(rc, IC<0>{}, cur, x);
                        xf_half(rc, IC<1>{}, cur, x);
                        xf_end(rc, 0, x);
                    }
                    self(self, IC<r + 1>{});
                }
            };
            xf_all(xf_all, IC<0>{});
        }
    }
    life(11);
    if constexpr (SS) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");    // (tiles 0..2 of every thread)
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (nchunk9 > 1) nxt = load_chunk(1);
    life(1);

    // ---- 3x3 chunks -----------------------------------------------------------------------------------
    // Step t of a chunk (tap t) runs two k16 phases of eight MFMAs; the fragments of a phase are read while the MFMAs
    // of the previous one run (two register sets).  Barrier B_t sits between the phases.  Every wave reaches it with
    // lgkmcnt(0), i.e. with all its reads of weight tile t (and, at t = 8, its normalisation writes) complete, and with
    // tile t+1 landed (vmcnt), so after B_t tile t+1 (at t = 8 also the next chunk's patch) may be read and the DMA group
    //     G_t = [ weight tile t+4 -> the ring slot of tile t (2 pieces), patch rounds of the next chunk into the other
    //             patch buffer (t = 0, 1, 2) ]
    // is issued: every tile has three full steps to land.  vmcnt before B_t certifies the tile issued in G_(t-3);
    // younger are the patch pieces of G_(t-3) and all of G_(t-2), G_(t-1):
    //     N_t = np(t-3) + 2 + np(t-2) + 2 + np(t-1).
    // Patch round r is normalised in place by its issuing thread between B_s and B_(s+1), s = 3 + r, in two halves that
    // ride in the shadow of the 16 MFMAs of that window; the window opens with its own counted wait
    // (dmas_after_round) for that piece, normally long satisfied; B_8 publishes the patch.
    int slot = 0;                                        // ring slot of the current step's weight tile
    int pbuf = 0;                                        // patch buffer of the current chunk
    int wstep = 0;                                       // global K-step index of the current step
    if (nchunk9 > 0) {
        read_b(IC<0>{}, IC<0>{});
#pragma unroll
        for (int i = 0; i < TN; ++i) read_a(IC<0>{}, IC<0>{}, i);
    }
    constexpr int NFULL = NROUND - (NREMW < NW ? 1 : 0);   // rounds in which every wave owns pieces
    static_assert(NFULL <= 5, "normalisation schedule: one full round per window of steps 3..7 (+ a partial one)");
    auto chunk_body = [&](auto doxc, const int c) __attribute__((always_inline)) {
        constexpr bool DOX = decltype(doxc)::value != 0 && !(ABL & 8) && !(ABL & 16);
        u32x4 xa = {0u, 0u, 0u, 0u};                           // round in flight
        f32x4 sa = {0.f, 0.f, 0.f, 0.f}, ha = {0.f, 0.f, 0.f, 0.f};
        // window of step s (3..7): position 0 = second phase of step s, position 1 = first phase of step s+1; it
        // normalises full round s-3, four channels per position
        // xf_pre: LDS reads of a position, issued at the end of the phase before it (they return during the barrier wait)
        auto xf_pre = [&](auto sc, auto pc) {
            constexpr int s = decltype(sc)::value, p = decltype(pc)::value;
            constexpr int r = s >= 3 && s <= 7 && s - 3 < NFULL ? s - 3 : -1;
            if constexpr (DOX && r >= 0) {
                if constexpr (p == 0) {
                    // own piece landed?  (G_s is issued later in this phase: count up to G_(s-1))
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(dmas_after_round(NROUND, NWP, r, s - 1)) : "memory");
                    xf_begin(IC<r>{}, pbuf ^ 1, xa);
                }
                const float *sc4 = ssL + nxt.ssbase + piece_of(r).lc * 8 + 4 * p;
                sa = *reinterpret_cast<const f32x4 *>(sc4);
                ha = *reinterpret_cast<const f32x4 *>(sc4 + a.ssC);
            }
        };
        float ny[4] = {0.f, 0.f, 0.f, 0.f}, ne[4] = {0.f, 0.f, 0.f, 0.f};
        auto xf_math = [&](auto sc, auto pc, auto ic) {          // stage ic of the window position (s, p)
            constexpr int s = decltype(sc)::value, p = decltype(pc)::value, i = decltype(ic)::value;
            constexpr int r = s >= 3 && s <= 7 && s - 3 < NFULL ? s - 3 : -1;
            if constexpr (DOX && r >= 0) {
                norm_stage(ic, pc, xa, sa, ha, ny, ne);
                if constexpr (p == 1 && i == STG_C) xf_end(IC<r>{}, pbuf ^ 1, xa);
            }
        };
        // the partial last round (pieces of waves < NREMW only): in one go at the end of the last window (the
        // registers of the finished round are free again)
        auto xf_tail = [&](auto sc, auto pc) {
            constexpr int s = decltype(sc)::value, p = decltype(pc)::value;
            if constexpr (DOX && NREMW < NW && s == 7 && p == 1) {
                constexpr int r = NROUND - 1;
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(dmas_after_round(NROUND, NWP, r, s)) : "memory");
                if (w < NREMW) {
                    xf_begin(IC<r>{}, pbuf ^ 1, xa);
                    xf_half(IC<r>{}, IC<0>{}, nxt, xa);
                    xf_half(IC<r>{}, IC<1>{}, nxt, xa);
                    xf_end(IC<r>{}, pbuf ^ 1, xa);
                }
            }
        };
        // MFMAs of a phase with everything else of the phase issued in their shadow, in pinned order (phase_pinned): per
        // MFMA group one fragment read (the next phase's operands), one LDS-DMA while there are any (dma(i)), and one stage
        // of the normalisation arithmetic.
        auto phase = [&](auto kcur, auto tnext, auto knext, auto sc, auto pc, auto &&dma) {
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (!(ABL & 512)) __builtin_amdgcn_s_setprio(1);      // two independent workgroups per SIMD: the MFMA
                                                                           // phase of one outranks the other's DMA issue / VALU
            phase_pinned(kcur, tnext, knext, [&](auto ic) {
                dma(ic);
                xf_math(sc, pc, ic);
            });
            if constexpr (!(ABL & 512)) __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
        };
        auto no_dma = [](auto) {};
        auto mark = [&](int t, int k) {
            if constexpr ((ABL & 64) != 0) {
                if (blockIdx.x == 0 && c == 1) {
                    const unsigned tm = (unsigned)__builtin_amdgcn_s_memtime();
                    if (l == 0) dbg[(w * 9 + t) * 6 + k] = tm;
                }
            }
        };
        auto step = [&](auto tc) {
            constexpr int t = decltype(tc)::value;
            mark(t, 0);
            // first phase: MFMAs of (tap t, k 0..15); reads of (tap t, k 16..31); then the LDS reads of the normalisation
            // slice that rides under the second phase, so that they return during the wait / barrier
            phase(IC<0>{}, tc, IC<1>{}, IC<t - 1>{}, IC<1>{}, no_dma);
            xf_tail(IC<t - 1>{}, IC<1>{});
            xf_pre(tc, IC<0>{});
            // advance the weight ring (and, at the last tap, the chunk) before the reads of the next step
            {
                const int d = slot == WSTAGES - 1 ? -(WSTAGES - 1) * W_BYTES : W_BYTES;
                wa += d;
                slot = (slot + 1) & (WSTAGES - 1);
                ++wstep;
            }
            if constexpr (t == 8) {
                const int d = pbuf ? -PATCH_BYTES : PATCH_BYTES;
#pragma unroll
                for (int kyp = 0; kyp < 2; ++kyp)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) pa[kyp][kx] += d;
                pbuf ^= 1;
                cur = nxt;
                nxt = load_chunk(c + 2 < nchunk9 ? c + 2 : nchunk9 - 1);
            }
            constexpr int N = rounds_at_tap(NROUND, t - 3) + NWP + rounds_at_tap(NROUND, t - 2) + NWP +
                              rounds_at_tap(NROUND, t - 1);
            mark(t, 1);
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
            mark(t, 2);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            mark(t, 3);
            // second phase: DMA group G_t, one DMA behind each MFMA group; MFMAs of (tap t, k 16..31); reads of (tap t+1, k 0..15)
            {
                // tile of step t+4 (wstep was advanced: it is the index of step t+1) -> the slot of tile t
                const int ws = wstep + 3 < nstep9 ? wstep + 3 : nstep9 - 1;
                const int wslot = (slot + 3) & (WSTAGES - 1);
                constexpr int R0 = rounds_per_tap(NROUND) * t, NR = rounds_at_tap(NROUND, t), ND = NWP + NR;
                // DMA k of the group (weight pieces, then patch rounds) goes behind MFMA group k, the overflow behind the last
                phase(IC<1>{}, IC<(t + 1) % 9>{}, IC<0>{}, tc, IC<0>{}, [&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    auto one = [&](auto self, auto kc) {
                        constexpr int k = decltype(kc)::value;
                        if constexpr (k < ND) {
                            if constexpr (k == i || (i == TN - 1 && k >= TN)) {
                                if constexpr (k < NWP) {
                                    if (!(ABL & 2)) w_issue1(wslot, ws, k);
                                } else {
                                    if (!(ABL & 8)) patch_dma(IC<(k < ND ? R0 + k - NWP : 0)>{}, nxt, pbuf ^ 1);
                                }
                            }
                            self(self, IC<k + 1>{});
                        }
                    };
                    one(one, IC<0>{});
                });
            }
            xf_pre(tc, IC<1>{});                                 // for the first phase of the next step
            mark(t, 4);
        };
        step(IC<0>{});
        step(IC<1>{});
        step(IC<2>{});
        step(IC<3>{});
        step(IC<4>{});
        step(IC<5>{});
        step(IC<6>{});
        step(IC<7>{});
        step(IC<8>{});
    };
    // ---- super-step schedule (SS) ---------------------------------------------------------------------------
    // Steps as above, but the barrier (between the two phases of a step) only at taps 2, 5, 8: B(c, j).  All reads of the
    // tiles of taps <= 3j+2 are issued before it, so behind it the DMA batch
    //     [ patch round j of chunk c+2 -> buffer (c+2) % 3,  weight tiles 9c + 3j + 9 .. +2 -> slots 3j .. 3j+2 ]
    // may go out: four DMAs per thread, always (a dummy stands in for a missing round).  The wait before B(k) lets the
    // four DMAs of batch k-1 stay in flight, i.e. certifies batch k-2: the tiles of the next super-step (two super-steps
    // to land) and a patch round of the next chunk, which its issuing thread normalises in place behind B(k) (round j
    // in the second phase of step 3j and the first phase of step 3j+1, before B(c, j)); B(c, 2) publishes the chunk.
    int b1 = 1, b2 = 2;                                  // patch buffers of chunks c+1, c+2
    Chunk nx2 = cur;
    if constexpr (SS) nx2 = load_chunk(nchunk9 > 2 ? 2 : nchunk9 - 1 > 0 ? nchunk9 - 1 : 0);
    auto chunk_body_ss = [&](auto doxc, const int c) __attribute__((always_inline)) {
        constexpr bool DOX = decltype(doxc)::value != 0 && !(ABL & 8);
        u32x4 xa = {0u, 0u, 0u, 0u};
        f32x4 sa = {0.f, 0.f, 0.f, 0.f}, ha = {0.f, 0.f, 0.f, 0.f};
        auto mark = [&](int t, int k) {
            if constexpr ((ABL & 64) != 0) {
                if (blockIdx.x == 0 && c == 1 && w < 4) {
                    const unsigned tm = (unsigned)__builtin_amdgcn_s_memtime();
                    if (l == 0) dbg[(w * 9 + t) * 6 + k] = tm;
                }
            }
        };
        auto xs_pre = [&](auto sc, auto pc) {
            constexpr int s = decltype(sc)::value, p = decltype(pc)::value;
            constexpr int r = s >= 0 && s % 3 == 0 && s / 3 < NROUND ? s / 3 : -1;
            if constexpr (DOX && r >= 0) {
                if (xf_owner(r)) {
                    if constexpr (p == 0) xf_begin(IC<r>{}, b1, xa);
                    const float *sc4 = ssL + nxt.ssbase + piece_of(r).lc * 8 + 4 * p;
                    sa = *reinterpret_cast<const f32x4 *>(sc4);
                    ha = *reinterpret_cast<const f32x4 *>(sc4 + a.ssC);
                }
            }
        };
        float ny[4] = {0.f, 0.f, 0.f, 0.f}, ne[4] = {0.f, 0.f, 0.f, 0.f};
        auto xs_math = [&](auto sc, auto pc, auto ic) {
            constexpr int s = decltype(sc)::value, p = decltype(pc)::value, i = decltype(ic)::value;
            constexpr int r = s >= 0 && s % 3 == 0 && s / 3 < NROUND ? s / 3 : -1;
            if constexpr (DOX && r >= 0) {
                if (xf_owner(r)) {
                    norm_stage(ic, pc, xa, sa, ha, ny, ne);
                    if constexpr (p == 1 && i == STG_C) xf_end(IC<r>{}, b1, xa);
                }
            }
        };
        auto phase = [&](auto kcur, auto tnext, auto knext, auto sc, auto pc, auto &&dma) {
            __builtin_amdgcn_sched_barrier(0);
            phase_pinned(kcur, tnext, knext, [&](auto ic) {
                dma(ic);
                xs_math(sc, pc, ic);
            });
            __builtin_amdgcn_sched_barrier(0);
        };
        auto no_dma = [](auto) {};
        auto step = [&](auto tc) {
            constexpr int t = decltype(tc)::value;
            mark(t, 0);
            phase(IC<0>{}, tc, IC<1>{}, IC<t - 1>{}, IC<1>{}, no_dma);
            xs_pre(tc, IC<0>{});
            mark(t, 1);
            if constexpr (t % 3 == 2) {
                constexpr int j = t / 3;
                if constexpr (t == 8) {                       // the next chunk's patch, for the reads behind the barrier
                    const int d = c % 3 == 2 ? -2 * PATCH_BYTES : PATCH_BYTES;
#pragma unroll
                    for (int kyp = 0; kyp < 2; ++kyp)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) pa[kyp][kx] += d;
                }
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(1 + 3 * NWP) : "memory");
                mark(t, 2);
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                mark(t, 3);
                // the DMA batch: patch round j (or its stand-in) and tile 0 behind the first MFMA group, tiles 1 and 2 behind
                // the second.  (b1 / b2 / nxt / nx2 advance at t == 8 BEFORE the phase: the batch then belongs to the roles of
                // the next chunk -- so the round's chunk and buffer are taken now.)
                const Chunk dch = nx2;
                const int dbuf = b2;
                if constexpr (t == 8) {
                    const int b0 = b1;
                    b1 = b2;
                    b2 = b0 == 0 ? 2 : b0 - 1;               // (c+3) % 3 = c % 3: the buffer the finished chunk leaves
                    nxt = nx2;
                    nx2 = load_chunk(c + 3 < nchunk9 ? c + 3 : nchunk9 - 1);
                }
                static_assert(!SS || (TN == 2 && NWP == 1), "super-step DMA batch: two DMAs behind each of two MFMA groups");
                phase(IC<1>{}, IC<(t + 1) % 9>{}, IC<0>{}, tc, IC<0>{}, [&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    auto tile = [&](int k) {
                        const int ws = 9 * c + 3 * j + 9 + k;
                        if (!(ABL & 2)) w_issue(3 * j + k, ws < nstep9 ? ws : nstep9 - 1);
                    };
                    if constexpr (i == 0) {
                        if constexpr (j < NROUND && !(ABL & 8)) patch_dma(IC<(j < NROUND ? j : 0)>{}, dch, dbuf);
                        else __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lds_ptr_t)(smem + OFF_DUMP), 16, 0x80000000u, 0, 0, 0);
                        tile(0);
                    } else {
                        tile(1);
                        tile(2);
                    }
                });
            } else {
                phase(IC<1>{}, IC<(t + 1) % 9>{}, IC<0>{}, tc, IC<0>{}, no_dma);
            }
            xs_pre(tc, IC<1>{});
            mark(t, 4);
        };
        step(IC<0>{});
        step(IC<1>{});
        step(IC<2>{});
        step(IC<3>{});
        step(IC<4>{});
        step(IC<5>{});
        step(IC<6>{});
        step(IC<7>{});
        step(IC<8>{});
    };
    if (nchunk9 > 0) {
        if constexpr (SS) {
            if (cur.ssbase >= 0) {
                for (int c = 0; c + 1 < nchunk9; ++c) chunk_body_ss(IC<1>{}, c);
            } else {
                for (int c = 0; c + 1 < nchunk9; ++c) chunk_body_ss(IC<0>{}, c);
            }
            chunk_body_ss(IC<0>{}, nchunk9 - 1);
        } else {
            if (cur.ssbase >= 0) {
                for (int c = 0; c + 1 < nchunk9; ++c) chunk_body(IC<1>{}, c);
            } else {
                for (int c = 0; c + 1 < nchunk9; ++c) chunk_body(IC<0>{}, c);
            }
            chunk_body(IC<0>{}, nchunk9 - 1);
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    life(2);
    // ---- 1x1 chunks (raw centre pixels; patch + weight tile of chunk n+1 fly while chunk n multiplies) ----
    if (nchunk1 > 0) {
        auto issue1 = [&](const Chunk &c, int buf, int step) {
            auto issue_all = [&](auto self, auto rc) {
                constexpr int r = decltype(rc)::value;
                if constexpr (r < NROUND) {
                    patch_dma(rc, c, buf);
                    self(self, IC<r + 1>{});
                }
            };
            issue_all(issue_all, IC<0>{});
            w_issue(buf, step);
        };
        issue1(load_chunk(nchunk9), 0, nstep9);
        for (int n = 0; n < nchunk1; ++n) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (n + 1 < nchunk1) issue1(load_chunk(nchunk9 + n + 1), (n + 1) & 1, nstep9 + n + 1);
            const int buf = n & 1;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int wb = OFF_W + buf * W_BYTES + (wn * TN * 32 + q) * 64 + (((2 * ks + kh) ^ ((q >> 2) & 3)) << 4);
                const int pb = buf * PATCH_BYTES + ((row_base + lr + 1) * PW + lcx + 1) * 64 +
                               (((2 * ks + kh) ^ t32_patch_key(row_base + lr + 1, lcx + 1)) << 4);
                v8 ga[TN], gb[TM];
#pragma unroll
                for (int i = 0; i < TN; ++i) ga[i] = *reinterpret_cast<const v8 *>(smem + wb + i * 2048);
#pragma unroll
                for (int j = 0; j < TM; ++j) gb[j] = *reinterpret_cast<const v8 *>(smem + pb + 2 * j * PW * 64);
#pragma unroll
                for (int i = 0; i < TN; ++i)
#pragma unroll
                    for (int j = 0; j < TM; ++j) acc[i][j] = TT<T>::mfma(ga[i], gb[j], acc[i][j]);
            }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }

    life(3);
    if constexpr (NCO == 32) {
        // ---- head epilogue: bias, fp32 NCHW (lane = one pixel, 16 of the 32 padded channels) ------------------------
        float *o = (float *)a.out;
        const size_t HWs = (size_t)H * Wd;
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const int prow = row_base + 2 * j + lr;
            const size_t pix = (size_t)(y0 + prow) * Wd + x0 + lcx;
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int co = 8 * g + 4 * kh + e;
                    if (co < a.Cout)
                        store_wt(o + ((size_t)b * a.Cout + co) * HWs + pix,
                                 acc[0][j][4 * g + e] + *reinterpret_cast<const float *>(smem + OFF_BIAS + co * 4));
                }
        }
        return;
    }
    // ---- epilogue 1: bias + time embedding -> 16-bit tile in LDS ([BM][NCO ch], 16-byte chunks swizzled) ---------------
    constexpr int CH = NCO / 8;                          // 16-byte chunks per staged pixel row (16 or 8)
    constexpr int RB = NCO * 2;                          // bytes per staged pixel row
    auto skey = [](int pl) { return CH == 16 ? (pl & 15) : ((pl >> 1) & 7); };   // conflict-free for the writes and the reads
    char *stg = smem;
    f32x4 addv[TN][4];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            addv[i][g] = *reinterpret_cast<const f32x4 *>(smem + OFF_BIAS + (wn * TN * 32 + i * 32 + 8 * g + 4 * kh) * 4);
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int prow = row_base + 2 * j + lr;
        const int pl = prow * TW + lcx;                                   // pixel inside the tile
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int cl = wn * TN * 32 + i * 32 + 8 * g + 4 * kh;    // channel inside the block
                v4 ov;
#pragma unroll
                for (int e = 0; e < 4; ++e) ov[e] = (T)(acc[i][j][4 * g + e] + addv[i][g][e]);
                *reinterpret_cast<v4 *>(stg + pl * RB + ((((cl >> 3) ^ skey(pl)) << 4) | ((cl & 7) * 2))) = ov;
            }
    }
    __syncthreads();
    life(4);

    // ---- epilogue 2: residual (row-coalesced 16-B reads) + full-row stores + per-channel statistics ------
    constexpr int RPE = NT / CH;                         // pixel rows handled per pass
    constexpr int NPASS = BM / RPE;
    const int c16 = tid % CH, prw = tid / CH;            // 16-byte chunk (8 channels), pixel row slot
    // pass i handles tile pixel prw + RPE i, i.e. RPE / 16 image rows further down per pass: one element offset + a stride
    static_assert(RPE % 16 == 0, "a pass advances by whole tile rows");
    const size_t e0 = ((size_t)(b * H + y0 + (prw >> 4)) * Wd + x0 + (prw & 15)) * a.Cout + n0 + c16 * 8;
    const size_t estep = (size_t)(RPE / 16) * Wd * a.Cout;
    v8 rres[NPASS];
    if (a.resid) {
#pragma unroll
        for (int i = 0; i < NPASS; ++i) rres[i] = *reinterpret_cast<const v8 *>((const T *)a.resid + e0 + i * estep);
    }
    float s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
        const int pl = prw + RPE * i;
        v8 v = *reinterpret_cast<const v8 *>(stg + pl * RB + ((c16 ^ skey(pl)) << 4));
        if (a.resid) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (T)((float)v[e] + (float)rres[i][e]);
        }
        store_wt(reinterpret_cast<v8 *>((T *)a.out + e0 + i * estep), v);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float f = (float)v[e];
            s1[e] += f;
            s2[e] = fmaf(f, f, s2[e]);
        }
    }
    if (a.stats) {
        // the row slots of a wave (lanes with equal l % CH) in a fixed order, then the waves through LDS
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if constexpr (CH == 8) {
                s1[e] += __shfl_xor(s1[e], 8);
                s2[e] += __shfl_xor(s2[e], 8);
            }
            s1[e] += __shfl_xor(s1[e], 16);
            s2[e] += __shfl_xor(s2[e], 16);
            s1[e] += __shfl_xor(s1[e], 32);
            s2[e] += __shfl_xor(s2[e], 32);
        }
        float *red = reinterpret_cast<float *>(smem + BM * RB);            // [NW waves][NCO][2]
        if (l < CH) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                red[((w * NCO) + c16 * 8 + e) * 2 + 0] = s1[e];
                red[((w * NCO) + c16 * 8 + e) * 2 + 1] = s2[e];
            }
        }
        __syncthreads();
        if (tid < 2 * NCO) {
            float t = 0.f;
#pragma unroll
            for (int wv = 0; wv < NW; ++wv) t += red[wv * 2 * NCO + tid];
            store_wt(a.stats + ((size_t)(b * tps + tin) * a.Cout + n0) * 2 + tid, t);
        }
    }
    life(5);
}

template <int TH, int NW, int NCO = 128, int SS = 0> constexpr int t32_smem_bytes() {
    constexpr int NT = NW * 64;
    constexpr int NPIECE = (TH + 2) * 18 * 4, NROUND = (NPIECE + NT - 1) / NT;
    constexpr int NREMW = (NPIECE - (NROUND - 1) * NT + 63) / 64;
    constexpr int PATCH_BYTES = (NROUND - 1) * NT * 16 + NREMW * 1024;
    constexpr int main_bytes =
        (SS ? 3 : 2) * PATCH_BYTES + (SS ? 9 : 4) * NCO * 64 + T32_SS_BYTES + T32_MAX_CHUNKS * 16 + 512 + 1024;
    constexpr int epi_bytes = NCO >= 64 ? TH * 16 * NCO * 2 + NW * NCO * 2 * 4 : 0;
    return main_bytes > epi_bytes ? main_bytes : epi_bytes;
}

template <typename T, int TH, int ABL, int NW = 4, int NCO = 128, int SS = 0>
int launch_t32_t(const FusedArgs &a, hipStream_t st) {
    constexpr int smem = t32_smem_bytes<TH, NW, NCO, SS>();
    static_assert(smem <= (SS ? 160 : 80) * 1024, "LDS budget: two workgroups per CU (one with super-steps)");
    static bool attr = false;
    if (!attr) {
        BNDM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_t32<T, TH, ABL, NW, NCO, SS>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr = true;
    }
    const int tiles_x = a.W / 16, tiles_y = a.H / TH, tps = tiles_x * tiles_y, ntn = (a.Cout + NCO - 1) / NCO;
    int nsteps = 0;
    for (int i = 0; i < a.nseg; ++i) nsteps += a.seg[i].taps * (a.seg[i].C / 32);
    dim3 grid(a.B * tps * ntn);
    unsigned *dbg = nullptr;
    if constexpr ((ABL & 64) != 0) {
        static unsigned *buf = nullptr;
        if (!buf) BNDM_CHECK_HIP(hipMalloc(&buf, (4 * 9 * 6 + 32) * sizeof(unsigned)));
        dbg = buf;
    }
    hipLaunchKernelGGL((conv_t32<T, TH, ABL, NW, NCO, SS>), grid, dim3(NW * 64), smem, st, a, tiles_x, tps, ntn, nsteps, dbg);
#ifdef BNDM_ABLATION      // profiling builds only (tools/ablate.sh)
#include "ablation_t32_trace.inc"
#endif
    return launch_status("conv_t32");
}

}  // namespace

// true when conv_t32 can run this segment list: 3x3 segments first, 1x1 segments after them
bool conv_t32_supports(const FusedArgs &a) {
    bool seen1 = false;
    int nchunks = 0;
    for (int i = 0; i < a.nseg; ++i) {
        if (a.seg[i].C % 32 || a.seg[i].C > 2047) return false;
        if (a.seg[i].up != a.seg[0].up) return false;      // the piece descriptors carry the source resolution
        if (a.seg[i].taps == 1 && a.seg[i].ss_off >= 0) return false;   // 1x1 chunks are read raw: no normalised 1x1 segment
        if (a.seg[i].taps == 1) seen1 = true;
        else if (seen1) return false;
        else if ((a.seg[i].ss_off >= 0) != (a.seg[0].ss_off >= 0)) return false;   // all 3x3 segments normalised, or none
        nchunks += a.seg[i].C / 32;
    }
    // 32-bit buffer offsets: every source tensor must stay below 2 GiB
    for (int i = 0; i < a.nseg; ++i) {
        const long long px = a.seg[i].up ? (long long)(a.H / 2) * (a.W / 2) : (long long)a.H * a.W;
        if ((long long)a.B * px * a.seg[i].C * 2 >= (1LL << 31)) return false;
    }
    if (a.ss && 2 * a.ssC * 4 > T32_SS_BYTES) return false;
    return a.nseg >= 1 && a.seg[0].taps == 9 && nchunks <= T32_MAX_CHUNKS && a.W % 16 == 0 &&
           (a.out_nchw32 ? a.Cout <= 32 && !a.resid && !a.stats && !a.temb
                         : a.Cout % 128 == 0);
}

int conv_t32_tiles_per_sample(int TH, int H, int W) { return (H / TH) * (W / 16); }

// Host-side weight packing for conv_t32: out[n-tile][K-step][row r of 128][physical slot j of 4][8 elements], where
// K-steps run over the 3x3 segments' 32-channel chunks x 9 taps, then over the 1x1 segments' chunks, and slot j of row
// r holds channels 8 (j ^ ((r >> 2) & 3)) .. +7 of the step.  `w_of(seg, co, c, tap)` returns the fp32 weight.
std::vector<float> pack_weights_t32(const FusedSeg *seg, int nseg, int Cout,
                                    const std::function<float(int, int, int, int)> &w_of, int rows) {
    int nsteps = 0;
    for (int i = 0; i < nseg; ++i) nsteps += seg[i].taps * (seg[i].C / 32);
    const int ntn = (Cout + rows - 1) / rows;
    std::vector<float> out((size_t)ntn * nsteps * rows * 32, 0.f);
    int step = 0;
    for (int pass = 0; pass < 2; ++pass)                 // 3x3 segments first
        for (int i = 0; i < nseg; ++i) {
            if ((seg[i].taps == 9) != (pass == 0)) continue;
            for (int ci = 0; ci < seg[i].C / 32; ++ci)
                for (int t = 0; t < seg[i].taps; ++t, ++step)
                    for (int nt = 0; nt < ntn; ++nt)
                        for (int r = 0; r < rows; ++r) {
                            const int co = nt * rows + r;
                            if (co >= Cout) continue;
                            for (int j = 0; j < 4; ++j) {
                                const int s = j ^ ((r >> 2) & 3);
                                float *dst = &out[(((size_t)nt * nsteps + step) * rows + r) * 32 + j * 8];
                                for (int e = 0; e < 8; ++e) dst[e] = w_of(i, co, ci * 32 + s * 8 + e, t);
                            }
                        }
        }
    return out;
}

int launch_conv_t32(int dtype, int TH, const FusedArgs &a, hipStream_t st) {
    // two 4-wave workgroups per CU need a grid of at least ~two per CU; smaller grids get 8-wave workgroups
    if (a.out_nchw32) {                              // network head: 32-channel tiles, fp32 NCHW output
        if (dtype == BNDM_DTYPE_F16)
            return TH == 16 ? launch_t32_t<_Float16, 16, 0, 4, 32>(a, st) : launch_t32_t<_Float16, 8, 0, 4, 32>(a, st);
        return TH == 16 ? launch_t32_t<__bf16, 16, 0, 4, 32>(a, st) : launch_t32_t<__bf16, 8, 0, 4, 32>(a, st);
    }
    const long long nblk = (long long)a.B * (a.H / TH) * (a.W / 16) * (a.Cout / 128);
    const int nw = nblk >= 448 ? 4 : 8;
    if (dtype == BNDM_DTYPE_F16) {
#ifdef BNDM_ABLATION      // profiling builds only (tools/ablate.sh)
#include "ablation_t32_dispatch.inc"
#endif
        if (nw == 8)
            return TH == 16 ? launch_t32_t<_Float16, 16, 0, 8, 128, 1>(a, st) : launch_t32_t<_Float16, 8, 0, 8, 128, 1>(a, st);
        return TH == 16 ? launch_t32_t<_Float16, 16, 0>(a, st) : launch_t32_t<_Float16, 8, 0>(a, st);
    }
    if (nw == 8) return TH == 16 ? launch_t32_t<__bf16, 16, 0, 8, 128, 1>(a, st) : launch_t32_t<__bf16, 8, 0, 8, 128, 1>(a, st);
    return TH == 16 ? launch_t32_t<__bf16, 16, 0>(a, st) : launch_t32_t<__bf16, 8, 0>(a, st);
}

}  // namespace bndm
