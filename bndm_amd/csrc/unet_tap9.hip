// conv_tap9: the fused GroupNorm+SiLU 3x3 convolution of the FLOP-dominant UNet layers, written so that
// everything that varies per K-step is a compile-time constant.
//
// Same interface and tiling as conv_fused (unet_fused.hip): 512 threads = 8 waves as 4(M) x 2(N), a
// TH x 16 pixel tile of one sample x 128 output channels, K over 64-channel chunks of up to 4 segments,
// (TH+2) x 18 halo patch per chunk in LDS (double-buffered, normalised in place), weight tiles [128][64]
// per tap in a 4-slot LDS ring.  What differs is the loop: the nine taps of a chunk are unrolled, so
//   * every ds_read of a step is `base VGPR + immediate` (12 patch bases per (kx, ks) and 4 weight bases
//     per ks live in registers; nothing is recomputed per step),
//   * all global->LDS traffic is `buffer_load_dwordx4 ... lds` with the per-step part in the scalar offset
//     (no per-lane address arithmetic for weights, one multiply-add per patch piece, padding = the buffer's
//     out-of-range zeros),
//   * the `s_waitcnt vmcnt(N)` before each step's barrier is a per-tap constant,
//   * fragments are software-pipelined one k16 slice ahead inside each wave (no ping-pong groups), with one
//     barrier per K-step.
// 1x1 segments (conv_shortcut on the raw block input) run after the 3x3 chunks in a short double-buffered
// loop.  See DESIGN.md section 4 for the per-step timeline and the vmcnt bookkeeping.
#include "unet_kernels.hpp"
#include "unet_types.hpp"
#include <cstdlib>

namespace bndm {
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
// DMAs a thread has issued after patch round r by the end of group G_s (see the loop comment); nwp = weight pieces
// per thread and tile
constexpr int dmas_after_round(int nround, int nwp, int r, int s) {
    const int rpt = rounds_per_tap(nround), g = r / rpt;
    int n = (rpt * g + rpt - 1 < nround - 1 ? rpt * g + rpt - 1 : nround - 1) - r;
    for (int j = g + 1; j <= s; ++j) n += nwp + rounds_at_tap(nround, j);
    return n;
}

typedef uint32_t u32x4t __attribute__((ext_vector_type(4)));

constexpr int TAP9_MAX_CHUNKS = 64;

// wave-uniform description of one 64-channel chunk of a segment
struct Chunk {
    const void *src;             // the segment's tensor
    int bytes;                   // ... and its size
    int soff;                    // chunk * 128 bytes
    int C2;                      // bytes per source pixel
    int up;                      // source at half resolution
    int ssbase;                  // float offset of the chunk's scale row in the LDS table, or -1
    int kbase;                   // weight k-offset (elements) of tap 0
    int kstride;                 // k-offset between taps (= C of the segment)
};

// ABL: profiling switches (results are wrong when non-zero): 1 no MFMA, 2 no weight DMA, 4 no fragment reads,
// 8 no in-loop patch DMA / normalisation, 16 no in-loop normalisation (DMA kept), 64 record s_memtime marks of
// block 0 / chunk 1 into dbg[wave][tap][6]
// NW: waves per block.  8: 4(M) x 2(N) waves of 64x64, two per SIMD.  4: 2(M) x 2(N) waves of 128x64, one per SIMD
// with the whole 512-register file.
template <typename T, int TH, int ABL, int NW>
__global__ __launch_bounds__(NW * 64) void conv_tap9(const FusedArgs a, const int tiles_x, const int tps, const int ntn,
                                                 unsigned *__restrict__ dbg) {
    using v8 = typename TT<T>::v8;
    using v4 = typename TT<T>::v4;
    constexpr int TW = 16, PW = TW + 2, PH = TH + 2;
    constexpr int NPIECE = PH * PW * 8;                // 16-byte pieces per patch chunk
    constexpr int NT = NW * 64;
    constexpr int WAVES_M = NW / 2;
    constexpr int NWP = 1024 / NT;                     // weight-tile pieces per thread
    constexpr int NROUND = (NPIECE + NT - 1) / NT;     // patch DMA rounds per chunk (the last one is partial)
    constexpr int NREMW = (NPIECE - (NROUND - 1) * NT + 63) / 64;    // waves with pieces in the last round
    constexpr int DUMP_OFF = (NROUND - 1) * NT * 16 + NREMW * 1024;  // dead slot for the other waves' last round
    constexpr int PATCH_BYTES = DUMP_OFF + (NREMW < NW ? 1024 : 0);
    constexpr int BM = TH * TW;
    constexpr int TM = BM / (WAVES_M * 32);            // 32-pixel MFMA tiles per wave along M
    constexpr int TN = 2;
    static_assert(TM >= 1 && (TH / WAVES_M) % 2 == 0, "wave tiling");
    constexpr int WSTAGES = 4, W_BYTES = 128 * 128;
    constexpr int OFF_W = 2 * PATCH_BYTES;
    constexpr int OFF_SS = OFF_W + WSTAGES * W_BYTES;
    constexpr int OFF_TAB = OFF_SS + 8192;             // chunk descriptors, 32 B each (TAP9_MAX_CHUNKS of them)
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l = tid & 63;

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
    const int n0 = __builtin_amdgcn_readfirstlane(nt * 128);
    const int H = a.H, Wd = a.W;
    const int lgH = 31 - __builtin_clz(H), lgW = 31 - __builtin_clz(Wd);

    // ---- segment bookkeeping (all scalar) -------------------------------------------------------------
    // The segment fields are pulled into opaque scalars once: selecting among plain SSA values keeps the
    // descriptor in SGPRs, whereas a select chain over `a.seg[i]` is folded into a dynamically indexed stack
    // copy of the kernel argument (scratch loads inside the loop, and with them vmcnt(0) waits).
    uint32_t sg_lo[CONV_MAX_SEG], sg_hi[CONV_MAX_SEG];
    int sg_C[CONV_MAX_SEG], sg_up[CONV_MAX_SEG], sg_ss[CONV_MAX_SEG], sg_k0[CONV_MAX_SEG];
    {
        int k0 = 0;
#pragma unroll
        for (int i = 0; i < CONV_MAX_SEG; ++i) {
            const uint64_t u = (uint64_t)a.seg[i].src;
            sg_lo[i] = (uint32_t)u;
            sg_hi[i] = (uint32_t)(u >> 32);
            sg_C[i] = a.seg[i].C;
            sg_up[i] = a.seg[i].up;
            sg_ss[i] = a.seg[i].ss_off;
            sg_k0[i] = k0;
            k0 += a.seg[i].taps * a.seg[i].C;
            asm volatile("" : "+s"(sg_lo[i]), "+s"(sg_hi[i]), "+s"(sg_C[i]), "+s"(sg_up[i]), "+s"(sg_ss[i]), "+s"(sg_k0[i]));
        }
    }
    static_assert(CONV_MAX_SEG == 4, "select chains below cover four segments");
    auto pick = [&](int si, auto &arr) { return si == 0 ? arr[0] : si == 1 ? arr[1] : si == 2 ? arr[2] : arr[3]; };
    auto make_chunk = [&](int si, int ci) {
        Chunk c;
        const int C = pick(si, sg_C), up = pick(si, sg_up), ss = pick(si, sg_ss);
        const int px = up ? (H >> 1) * (Wd >> 1) : H * Wd;
        c.src = (const void *)(((uint64_t)pick(si, sg_hi) << 32) | pick(si, sg_lo));
        c.bytes = a.B * px * C * 2;
        c.soff = ci * 128;
        c.C2 = C * 2;
        c.up = up;
        c.ssbase = ss >= 0 ? ss + ci * 64 : -1;
        c.kbase = pick(si, sg_k0) + ci * 64;
        c.kstride = C;
        return c;
    };
    int nchunk9 = 0, nchunk1 = 0;           // 3x3 segments come first (checked by the launcher)
#pragma unroll
    for (int i = 0; i < CONV_MAX_SEG; ++i)
        if (i < a.nseg) {
            if (a.seg[i].taps == 9) {
                nchunk9 += a.seg[i].C >> 6;
            } else {
                nchunk1 += a.seg[i].C >> 6;
            }
        }
    // Chunk descriptors of the whole K loop (3x3 chunks first, then the 1x1 ones) are built once by the first
    // threads and kept in LDS: advancing to the next chunk is two broadcast ds_reads + readfirstlanes instead
    // of ~170 scalar instructions of select chains per chunk and wave.
    if (tid < nchunk9 + nchunk1) {
        int rem = tid, si = 0, ci = 0;
        bool found = false;
#pragma unroll
        for (int i = 0; i < CONV_MAX_SEG; ++i) {
            const int nci = sg_C[i] >> 6;
            if (!found && rem < nci) {
                si = i;
                ci = rem;
                found = true;
            }
            rem -= nci;
        }
        const Chunk c = make_chunk(si, ci);
        u32x4t lo, hi;
        lo[0] = (uint32_t)(uint64_t)c.src;
        lo[1] = (uint32_t)((uint64_t)c.src >> 32);
        lo[2] = (uint32_t)c.bytes;
        lo[3] = (uint32_t)c.soff;
        hi[0] = (uint32_t)c.C2 | ((uint32_t)c.up << 31);
        hi[1] = (uint32_t)c.ssbase;
        hi[2] = (uint32_t)c.kbase;
        hi[3] = 0;
        *reinterpret_cast<u32x4t *>(smem + OFF_TAB + tid * 32) = lo;
        *reinterpret_cast<u32x4t *>(smem + OFF_TAB + tid * 32 + 16) = hi;
    }
    auto load_chunk = [&](int n) {
        const u32x4t lo = *reinterpret_cast<const u32x4t *>(smem + OFF_TAB + n * 32);
        const u32x4t hi = *reinterpret_cast<const u32x4t *>(smem + OFF_TAB + n * 32 + 16);
        Chunk c;
        const uint32_t plo = __builtin_amdgcn_readfirstlane(lo[0]), phi = __builtin_amdgcn_readfirstlane(lo[1]);
        c.src = (const void *)(((uint64_t)phi << 32) | plo);
        c.bytes = __builtin_amdgcn_readfirstlane(lo[2]);
        c.soff = __builtin_amdgcn_readfirstlane(lo[3]);
        const uint32_t cu = __builtin_amdgcn_readfirstlane(hi[0]);
        c.C2 = cu & 0x7fffffff;
        c.up = cu >> 31;
        c.ssbase = __builtin_amdgcn_readfirstlane(hi[1]);
        c.kbase = __builtin_amdgcn_readfirstlane(hi[2]);
        c.kstride = c.C2 >> 1;
        return c;
    };

    // ---- patch piece descriptors (independent of the chunk) ----------------------------------------
    // piece = round * 512 + tid = (patch pixel, 16-byte slot); the XOR swizzle sits on the source side:
    // slot s of pixel (py, px) holds channel group s ^ ((px >> 1) & 7), which makes every ds_read_b128 lane
    // group of the fragment reads hit 16 distinct bank classes for all nine tap shifts.
    int p_full[NROUND];                                // source pixel index of the piece, -1: padding
    int p_valid = 0;
    uint64_t p_lcpack = 0;                             // 3 bits per round: source channel group of the piece
#pragma unroll
    for (int r = 0; r < NROUND; ++r) {
        const int piece = r * NT + tid;
        const int pc = piece < NPIECE ? piece : NPIECE - 1;
        const int pp = pc >> 3, pch = pc & 7;
        const int pyy = pp / PW, pxx = pp - pyy * PW;
        const int iy = y0 - 1 + pyy, ix = x0 - 1 + pxx;
        const bool ok = piece < NPIECE && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)Wd;
        p_full[r] = ok ? (b * H + iy) * Wd + ix : -1;                      // -1: out-of-range offset -> zeros
        p_valid |= ok ? (1 << r) : 0;
        p_lcpack |= (uint64_t)(pch ^ ((pxx >> 1) & 7)) << (3 * r);
    }
    auto patch_dma = [&](auto rc, const Chunk &c, int buf) {
        constexpr int r = decltype(rc)::value;
        int pix = p_full[r];
        if (c.up) {                                    // nearest-2x source: H and W are powers of two
            const int ix = pix & (Wd - 1), iy = (pix >> lgW) & (H - 1), bb = pix >> (lgW + lgH);
            pix = pix < 0 ? -1 : ((((bb << (lgH - 1)) + (iy >> 1)) << (lgW - 1)) + (ix >> 1));
        }
        const int lc16 = (int)((p_lcpack >> (3 * r)) & 7) << 4;
        const unsigned voff = (unsigned)(pix * c.C2 + lc16);
        char *dst = smem + buf * PATCH_BYTES + ((r < NROUND - 1 || w < NREMW) ? r * (NT * 16) + w * 1024 : DUMP_OFF);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(uniform_rsrc(c.src, c.bytes), (lds_ptr_t)dst, 16, voff,
                                                 __builtin_amdgcn_readfirstlane(c.soff), 0, 0);
    };
    // GroupNorm scale/shift + SiLU applied in place to this thread's own piece of a round (the reference pads
    // AFTER the activation: padding / tail pieces are rewritten unchanged, i.e. stay zero)
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    using v2 = typename TT<T>::v2;
    const float *ssL = reinterpret_cast<const float *>(smem + OFF_SS);
    // a round is processed as: xf_begin (read the piece), four xf_slice calls (two channels each; the result
    // replaces dword q of the piece, so a round in flight costs four registers), xf_end (write it back)
    auto xf_begin = [&](auto rc, int buf, u32x4 &x) {
        constexpr int r = decltype(rc)::value;
        x = *reinterpret_cast<const u32x4 *>(smem + buf * PATCH_BYTES + r * (NT * 16) + tid * 16);
    };
    auto xf_slice = [&](auto rc, auto qc, const Chunk &c, u32x4 &x) {
        constexpr int r = decltype(rc)::value, q = decltype(qc)::value;
        const int lc = (int)((p_lcpack >> (3 * r)) & 7);
        const bool valid = (p_valid >> r) & 1;
        const float *sc = ssL + c.ssbase + lc * 8 + 2 * q;
        const f32x2 s2 = *reinterpret_cast<const f32x2 *>(sc), h2 = *reinterpret_cast<const f32x2 *>(sc + a.ssC);
        const unsigned xq = x[q];                      // (bit_cast straight from the element lvalue reads element 0)
        const v2 in = __builtin_bit_cast(v2, xq);
        v2 o;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const float f = fmaf((float)in[e], s2[e], h2[e]);
            o[e] = (T)(f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * f)));
        }
        x[q] = valid ? __builtin_bit_cast(unsigned, o) : xq;
    };
    auto xf_end = [&](auto rc, int buf, const u32x4 &x) {
        constexpr int r = decltype(rc)::value;
        *reinterpret_cast<u32x4 *>(smem + buf * PATCH_BYTES + r * (NT * 16) + tid * 16) = x;
    };
    auto xf_owner = [&](int r) { return r < NROUND - 1 || w < NREMW; };    // wave-uniform

    // ---- weight tiles: [128 rows][128 B] per tap, 1024 pieces, 2 per thread ----------------------------
    const __amdgpu_buffer_rsrc_t wrs = uniform_rsrc((const char *)a.Wgt + (size_t)n0 * a.Ktot * 2, 128 * a.Ktot * 2);
    const unsigned wvoff0 = (unsigned)(((tid >> 3) * a.Ktot + (((tid & 7) ^ ((tid >> 4) & 7)) << 3)) * 2);
    const unsigned wvstep = (unsigned)((NT / 8) * a.Ktot * 2);          // NT/8 tile rows per block-wide instruction
    auto w_issue = [&](int slot, int kofs) {
        char *base = smem + OFF_W + slot * W_BYTES + w * 1024;
        const int so = __builtin_amdgcn_readfirstlane(kofs * 2);
#pragma unroll
        for (int i = 0; i < NWP; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lds_ptr_t)(base + i * (NT * 16)), 16, wvoff0 + i * wvstep, so, 0, 0);
    };

    f32x16 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int wm = w % WAVES_M, wn = w / WAVES_M;
    const int q = l & 31, kh = l >> 5;
    const int row_base = wm * (TH / WAVES_M);
    const int lr = q >> 4, lcx = q & 15;

    // ---- fragment addresses ---------------------------------------------------------------------------
    // weights: row (wn*64 + q) of the tile, 16-byte slot (2*ks + kh) ^ ((q >> 1) & 7); + i*4096 for the second
    // 32-row tile.  patch: pixel (row_base + 2j + lr + ky, lcx + kx), slot (2*ks + kh) ^ (((lcx + kx) >> 1) & 7).
    int wa[4], pa[3][4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        wa[ks] = OFF_W + (wn * 64 + q) * 128 + ((((2 * ks + kh) ^ ((q >> 1) & 7))) << 4);
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
            pa[kx][ks] = ((row_base + lr) * PW + lcx) * 128 + (((2 * ks + kh) ^ (((lcx + kx) >> 1) & 7)) << 4);
    }
    v8 fa[3][TN], fb[3][TM];
    auto read_frags = [&](auto tc, auto kc, auto sc) {
        constexpr int t = decltype(tc)::value, ks = decltype(kc)::value, set = decltype(sc)::value;
        constexpr int ky = t / 3, kx = t % 3;
        if (ABL & 4) return;
#pragma unroll
        for (int i = 0; i < TN; ++i) fa[set][i] = *reinterpret_cast<const v8 *>(smem + wa[ks] + i * 4096);
#pragma unroll
        for (int j = 0; j < TM; ++j)
            fb[set][j] = *reinterpret_cast<const v8 *>(smem + pa[kx][ks] + (ky * PW + kx + 2 * j * PW) * 128);
    };
    auto multiply = [&](auto sc) {
        constexpr int set = decltype(sc)::value;
        if (ABL & 1) {
#pragma unroll
            for (int i = 0; i < TN; ++i) asm volatile("" ::"v"(fa[set][i]));
#pragma unroll
            for (int j = 0; j < TM; ++j) asm volatile("" ::"v"(fb[set][j]));
            return;
        }
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j) acc[i][j] = TT<T>::mfma(fa[set][i], fb[set][j], acc[i][j]);
    };

    // ---- prologue -------------------------------------------------------------------------------------
    // scale/shift table (one 16-byte piece per thread, zeros past its end), patch of chunk 0, weight tiles of
    // taps 0..3 -- all by LDS-DMA, so they retire in issue order and one counted wait separates them
    Chunk cur = make_chunk(0, 0);
    Chunk nxt = cur;
    {
        const __amdgpu_buffer_rsrc_t srs =
            uniform_rsrc(a.ss ? a.ss + (size_t)b * 2 * a.ssC : (const float *)a.zeros, a.ss ? 2 * a.ssC * 4 : 0);
#pragma unroll
        for (int i = 0; i < 512 / NT; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srs, (lds_ptr_t)(smem + OFF_SS + i * (NT * 16) + w * 1024), 16,
                                                     (unsigned)((i * NT + tid) * 16), 0, 0, 0);
    }
    if (nchunk9 > 0) {
        auto issue_all = [&](auto self, auto rc) {
            constexpr int r = decltype(rc)::value;
            if constexpr (r < NROUND) {
                patch_dma(rc, cur, 0);
                self(self, IC<r + 1>{});
            }
        };
        issue_all(issue_all, IC<0>{});
        w_issue(0, cur.kbase);
        w_issue(1, cur.kbase + cur.kstride);
        w_issue(2, cur.kbase + 2 * cur.kstride);
        w_issue(3, cur.kbase + 3 * cur.kstride);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * NWP) : "memory");      // table + own patch pieces landed
        __builtin_amdgcn_s_barrier();                         // ... in every wave (the table is shared)
        asm volatile("" ::: "memory");
        if (!(ABL & 8) && cur.ssbase >= 0) {
            auto xf_all = [&](auto self, auto rc) {
                constexpr int r = decltype(rc)::value;
                if constexpr (r < NROUND) {
                    if (xf_owner(r)) {
                        u32x4 x;
                        xf_begin(rc, 0, x);
                        xf_slice(rc, IC<0>{}, cur, x);
                        xf_slice(rc, IC<1>{}, cur, x);
                        xf_slice(rc, IC<2>{}, cur, x);
                        xf_slice(rc, IC<3>{}, cur, x);
                        xf_end(rc, 0, x);
                    }
                    self(self, IC<r + 1>{});
                }
            };
            xf_all(xf_all, IC<0>{});
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (nchunk9 > 1) nxt = load_chunk(1);

    // ---- 3x3 chunks -----------------------------------------------------------------------------------
    // Step t of a chunk (tap t) runs four k16 phases; the fragments of phase p+2 are read while the MFMAs of
    // phase p run (three register sets, set = p % 3; a chunk has 36 phases, so the assignment is static).
    // Barrier B_t sits between phases 1 and 2.  Every wave reaches it with lgkmcnt(0), i.e. with all its reads of
    // weight tile t (and, at t = 8, its normalisation writes) complete, and with tile t+1 landed (vmcnt), so
    // after B_t tile t+1 (at t = 8 also the next chunk's patch) may be read and the DMA group
    //     G_t = [ weight tile t+4 -> the ring slot of tile t (2 pieces), patch rounds 3t..3t+2 of the next chunk
    //             into the other patch buffer (t = 0, 1) ]
    // is issued: every tile has three full steps to land.  vmcnt before B_t certifies the tile issued in G_(t-3);
    // younger are the patch pieces of G_(t-3) and all of G_(t-2), G_(t-1):
    //     N_t = np(t-3) + 2 + np(t-2) + 2 + np(t-1).
    // Patch round r is normalised in place by its issuing thread between B_s and B_(s+1), s = 3 + r (rounds >= 4:
    // s = 7), in four slices that ride in the shadow of the 16 MFMAs of that window; the window opens with its own
    // counted wait (dmas_after_round) for that piece, normally long satisfied; B_8 publishes the patch.
    int slot = 0;                                        // ring slot of the current step's weight tile
    int pbuf = 0;                                        // patch buffer of the current chunk
    if (nchunk9 > 0) {
        read_frags(IC<0>{}, IC<0>{}, IC<0>{});
        read_frags(IC<0>{}, IC<1>{}, IC<1>{});
    }
    constexpr int NFULL = NROUND - (NREMW < NW ? 1 : 0);   // rounds in which every wave owns pieces
    constexpr int RPW = (NFULL + 4) / 5;                   // full rounds normalised per window (windows of steps 3..7)
    static_assert(RPW >= 1 && RPW <= 2, "normalisation schedule: five windows of one or two full rounds + a partial one");
    // The chunk body is instantiated with (DOX) and without the normalisation of the next chunk's patch, so that
    // the slices sit in the same basic block as the MFMAs and are interleaved with them.  A conv's 3x3 segments
    // are either all normalised or none (checked by the launcher): the loop over chunks 0..n-2 uses one variant,
    // the last chunk (no successor to prepare) always the plain one.
    auto chunk_body = [&](auto doxc, const int c) __attribute__((always_inline)) {
        constexpr bool DOX = decltype(doxc)::value != 0 && !(ABL & 8) && !(ABL & 16);
        u32x4 xa[RPW], xb = {0u, 0u, 0u, 0u};                  // rounds in flight (xb: the partial last round)
        f32x2 sa[RPW], ha[RPW];                                // scale / shift of the slices in flight
#pragma unroll
        for (int k = 0; k < RPW; ++k) {
            xa[k] = u32x4{0u, 0u, 0u, 0u};
            sa[k] = ha[k] = f32x2{0.f, 0.f};
        }
        // window of step s (3..7): positions 0..3 = phase 2, 3 of step s and phase 0, 1 of step s+1; it normalises
        // the full rounds RPW*(s-3) .. RPW*(s-3)+RPW-1, two channels of each per position
        // xf_pre: LDS reads of the position, issued before the fragment reads of the phase
        auto xf_pre1 = [&](auto sc, auto wc, auto kc) {
            constexpr int s = decltype(sc)::value, q = decltype(wc)::value, k = decltype(kc)::value;
            constexpr int r = s >= 3 && s <= 7 && RPW * (s - 3) + k < NFULL ? RPW * (s - 3) + k : -1;
            if constexpr (DOX && r >= 0) {
                if constexpr (q == 0) {
                    // own piece landed?  (G_s is issued later in this phase: count up to G_(s-1))
                    if constexpr (!(ABL & 128))
                        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(dmas_after_round(NROUND, NWP, r, s - 1)) : "memory");
                    xf_begin(IC<r>{}, pbuf ^ 1, xa[k]);
                }
                const int lc = (int)((p_lcpack >> (3 * r)) & 7);
                const float *sc = ssL + nxt.ssbase + lc * 8 + 2 * q;
                sa[k] = *reinterpret_cast<const f32x2 *>(sc);
                ha[k] = *reinterpret_cast<const f32x2 *>(sc + a.ssC);
            }
        };
        auto xf_pre = [&](auto sc, auto wc) {
            xf_pre1(sc, wc, IC<0>{});
            if constexpr (RPW > 1) xf_pre1(sc, wc, IC<1>{});
        };
        auto xf_math1 = [&](auto sc, auto wc, auto kc) {
            constexpr int s = decltype(sc)::value, q = decltype(wc)::value, k = decltype(kc)::value;
            constexpr int r = s >= 3 && s <= 7 && RPW * (s - 3) + k < NFULL ? RPW * (s - 3) + k : -1;
            if constexpr (DOX && r >= 0) {
                const bool valid = (p_valid >> r) & 1;
                const unsigned xq = xa[k][q];
                const v2 in = __builtin_bit_cast(v2, xq);
                v2 o;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const float f = fmaf((float)in[e], sa[k][e], ha[k][e]);
                    o[e] = (T)(f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * f)));
                }
                xa[k][q] = valid ? __builtin_bit_cast(unsigned, o) : xq;
                if constexpr (q == 3) xf_end(IC<r>{}, pbuf ^ 1, xa[k]);
            }
        };
        auto xf_math = [&](auto sc, auto wc) {
            xf_math1(sc, wc, IC<0>{});
            if constexpr (RPW > 1) xf_math1(sc, wc, IC<1>{});
        };
        // the partial last round (pieces of waves < NREMW only) slice by slice in the window of step 7
        auto xf_tail = [&](auto sc, auto wc) {
            constexpr int s = decltype(sc)::value, q = decltype(wc)::value;
            if constexpr (DOX && NREMW < NW && s == 7) {
                constexpr int r = NROUND - 1;
                if constexpr (q == 0 && !(ABL & 128))
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(dmas_after_round(NROUND, NWP, r, s)) : "memory");
                if (w < NREMW) {
                    if constexpr (q == 0) xf_begin(IC<r>{}, pbuf ^ 1, xb);
                    xf_slice(IC<r>{}, wc, nxt, xb);
                    if constexpr (q == 3) xf_end(IC<r>{}, pbuf ^ 1, xb);
                }
            }
        };
        // MFMAs of one phase with the slice of the window interleaved (1 MFMA : a few VALU)
        auto phase_math = [&](auto setc, auto sc, auto wc) {
            constexpr int s = decltype(sc)::value;
            constexpr int r = s >= 3 && s <= 7 && RPW * (s - 3) < NFULL ? RPW * (s - 3) : -1;
            multiply(setc);
            xf_math(sc, wc);
            if constexpr (DOX && r >= 0 && !(ABL & 1) && !(ABL & 256)) {
#pragma unroll
                for (int g = 0; g < TN * TM; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                          // 1 MFMA
                    __builtin_amdgcn_sched_group_barrier(0x002, 20 * RPW / (TN * TM), 0);       // VALU in its shadow
                }
            }
        };
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
            constexpr int p0 = 4 * t;
            mark(t, 0);
            // phase 0
            xf_pre(IC<t - 1>{}, IC<2>{});
            read_frags(tc, IC<2>{}, IC<(p0 + 2) % 3>{});
            phase_math(IC<p0 % 3>{}, IC<t - 1>{}, IC<2>{});
            xf_tail(IC<t - 1>{}, IC<2>{});
            // phase 1
            xf_pre(IC<t - 1>{}, IC<3>{});
            read_frags(tc, IC<3>{}, IC<(p0 + 3) % 3>{});
            phase_math(IC<(p0 + 1) % 3>{}, IC<t - 1>{}, IC<3>{});
            xf_tail(IC<t - 1>{}, IC<3>{});
            // advance the weight ring (and, at the last tap, the chunk) before the reads of the next step
            {
                const int d = slot == WSTAGES - 1 ? -(WSTAGES - 1) * W_BYTES : W_BYTES;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) wa[ks] += d;
                slot = (slot + 1) & (WSTAGES - 1);
            }
            if constexpr (t == 8) {
                const int d = pbuf ? -PATCH_BYTES : PATCH_BYTES;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) pa[kx][ks] += d;
                pbuf ^= 1;
                cur = nxt;
                nxt = load_chunk(c + 2 < nchunk9 ? c + 2 : nchunk9 - 1);
            }
            constexpr int N = rounds_at_tap(NROUND, t - 3) + NWP + rounds_at_tap(NROUND, t - 2) + NWP +
                              rounds_at_tap(NROUND, t - 1);
            mark(t, 1);
            if constexpr ((ABL & 128) != 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
            mark(t, 2);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            mark(t, 3);
            // phase 2
            xf_pre(tc, IC<0>{});
            read_frags(IC<(t + 1) % 9>{}, IC<0>{}, IC<(p0 + 4) % 3>{});
            if (!(ABL & 2)) {
                // tile of step t+4: after the advance above `cur` is already the next chunk at t = 8
                const int kofs = t + 4 < 9 ? cur.kbase + (t + 4) * cur.kstride
                                 : t == 8  ? cur.kbase + 3 * cur.kstride
                                           : nxt.kbase + (t + 4 - 9) * nxt.kstride;
                w_issue((slot + 3) & (WSTAGES - 1), kofs);       // slot was advanced: the slot of tile t
            }
            if constexpr (rounds_at_tap(NROUND, t) > 0) {
                if (!(ABL & 8)) {
                    constexpr int R0 = rounds_per_tap(NROUND) * t, NR = rounds_at_tap(NROUND, t);
                    static_assert(NR <= 4, "at most four patch rounds per tap");
                    patch_dma(IC<R0>{}, nxt, pbuf ^ 1);
                    if constexpr (NR > 1) patch_dma(IC<(NR > 1 ? R0 + 1 : 0)>{}, nxt, pbuf ^ 1);
                    if constexpr (NR > 2) patch_dma(IC<(NR > 2 ? R0 + 2 : 0)>{}, nxt, pbuf ^ 1);
                    if constexpr (NR > 3) patch_dma(IC<(NR > 3 ? R0 + 3 : 0)>{}, nxt, pbuf ^ 1);
                }
            }
            phase_math(IC<(p0 + 2) % 3>{}, tc, IC<0>{});
            xf_tail(tc, IC<0>{});
            mark(t, 4);
            // phase 3
            xf_pre(tc, IC<1>{});
            read_frags(IC<(t + 1) % 9>{}, IC<1>{}, IC<(p0 + 5) % 3>{});
            phase_math(IC<(p0 + 3) % 3>{}, tc, IC<1>{});
            xf_tail(tc, IC<1>{});
            mark(t, 5);
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
        if (cur.ssbase >= 0) {
            for (int c = 0; c + 1 < nchunk9; ++c) chunk_body(IC<1>{}, c);
        } else {
            for (int c = 0; c + 1 < nchunk9; ++c) chunk_body(IC<0>{}, c);
        }
        chunk_body(IC<0>{}, nchunk9 - 1);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    // ---- 1x1 chunks (raw centre pixels; patch + weight tile of chunk n+1 fly while chunk n multiplies) ----
    if (nchunk1 > 0) {
        Chunk c1 = load_chunk(nchunk9);
        auto issue1 = [&](const Chunk &c, int buf) {
            auto issue_all = [&](auto self, auto rc) {
                constexpr int r = decltype(rc)::value;
                if constexpr (r < NROUND) {
                    patch_dma(rc, c, buf);
                    self(self, IC<r + 1>{});
                }
            };
            issue_all(issue_all, IC<0>{});
            w_issue(buf, c.kbase);
        };
        issue1(c1, 0);
        for (int n = 0; n < nchunk1; ++n) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            const Chunk c2 = load_chunk(nchunk9 + (n + 1 < nchunk1 ? n + 1 : nchunk1 - 1));
            if (n + 1 < nchunk1) issue1(c2, (n + 1) & 1);
            const int buf = n & 1;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int wb = OFF_W + buf * W_BYTES + (wn * 64 + q) * 128 + ((((2 * ks + kh) ^ ((q >> 1) & 7))) << 4);
                const int pb = buf * PATCH_BYTES + ((row_base + lr + 1) * PW + lcx + 1) * 128 +
                               (((2 * ks + kh) ^ (((lcx + 1) >> 1) & 7)) << 4);
                v8 ga[TN], gb[TM];
#pragma unroll
                for (int i = 0; i < TN; ++i) ga[i] = *reinterpret_cast<const v8 *>(smem + wb + i * 4096);
#pragma unroll
                for (int j = 0; j < TM; ++j) gb[j] = *reinterpret_cast<const v8 *>(smem + pb + 2 * j * PW * 128);
#pragma unroll
                for (int i = 0; i < TN; ++i)
#pragma unroll
                    for (int j = 0; j < TM; ++j) acc[i][j] = TT<T>::mfma(ga[i], gb[j], acc[i][j]);
            }
            c1 = c2;
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }

    // ---- epilogue 1: bias + time embedding -> 16-bit tile in LDS ([BM][128 ch], swizzled) ---------------
    char *stg = smem;
    f32x4 addv[TN][4];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) addv[i][g] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (a.bias) {
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                addv[i][g] = *reinterpret_cast<const f32x4 *>(a.bias + n0 + wn * 64 + i * 32 + 8 * g + 4 * kh);
    }
    if (a.temb) {
        const float *tembp = a.temb + (size_t)b * a.temb_bstride + a.temb_off;
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                addv[i][g] += *reinterpret_cast<const f32x4 *>(tembp + n0 + wn * 64 + i * 32 + 8 * g + 4 * kh);
    }
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int prow = row_base + 2 * j + lr;
        const int pl = prow * TW + lcx;                                   // pixel inside the tile
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int cl = wn * 64 + i * 32 + 8 * g + 4 * kh;         // channel inside the block
                v4 ov;
#pragma unroll
                for (int e = 0; e < 4; ++e) ov[e] = (T)(acc[i][j][4 * g + e] + addv[i][g][e]);
                *reinterpret_cast<v4 *>(stg + pl * 256 + ((((cl >> 3) ^ (pl & 15)) << 4) | ((cl & 7) * 2))) = ov;
            }
    }
    __syncthreads();

    // ---- epilogue 2: residual (row-coalesced 16-B reads) + full-row stores + per-channel statistics ------
    constexpr int RPE = NT / 16;                         // pixel rows handled per pass
    constexpr int NPASS = BM / RPE;
    const int c16 = tid & 15, prw = tid >> 4;            // 16-byte chunk (8 channels), pixel row slot
    v8 rres[NPASS];
    if (a.resid) {
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            const int pl = prw + RPE * i;
            const size_t m = (size_t)(b * H + y0 + (pl >> 4)) * Wd + x0 + (pl & 15);
            rres[i] = *reinterpret_cast<const v8 *>((const T *)a.resid + m * a.Cout + n0 + c16 * 8);
        }
    }
    float s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
        const int pl = prw + RPE * i;
        v8 v = *reinterpret_cast<const v8 *>(stg + pl * 256 + ((c16 ^ (pl & 15)) << 4));
        if (a.resid) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (T)((float)v[e] + (float)rres[i][e]);
        }
        const size_t m = (size_t)(b * H + y0 + (pl >> 4)) * Wd + x0 + (pl & 15);
        *reinterpret_cast<v8 *>((T *)a.out + m * a.Cout + n0 + c16 * 8) = v;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float f = (float)v[e];
            s1[e] += f;
            s2[e] = fmaf(f, f, s2[e]);
        }
    }
    if (a.stats) {
        float *red = reinterpret_cast<float *>(smem + BM * 256);           // [RPE][128][2]
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            red[((prw * 128) + c16 * 8 + e) * 2 + 0] = s1[e];
            red[((prw * 128) + c16 * 8 + e) * 2 + 1] = s2[e];
        }
        __syncthreads();
        if (tid < 256) {
            float part[RPE];
#pragma unroll
            for (int r = 0; r < RPE; ++r) part[r] = red[r * 256 + tid];     // independent loads, then a fixed-order sum
            float t = 0.f;
#pragma unroll
            for (int r = 0; r < RPE; ++r) t += part[r];
            a.stats[((size_t)(b * tps + tin) * a.Cout + n0) * 2 + tid] = t;
        }
    }
}

template <typename T, int TH, int ABL, int NW = 8>
int launch_tap9_t(const FusedArgs &a, hipStream_t st) {
    constexpr int NT = NW * 64;
    constexpr int NPIECE = (TH + 2) * 18 * 8, NROUND = (NPIECE + NT - 1) / NT;
    constexpr int NREMW = (NPIECE - (NROUND - 1) * NT + 63) / 64;
    constexpr int PATCH_BYTES = (NROUND - 1) * NT * 16 + NREMW * 1024 + (NREMW < NW ? 1024 : 0);
    constexpr int main_bytes = 2 * PATCH_BYTES + 4 * 16384 + 8192 + TAP9_MAX_CHUNKS * 32;
    constexpr int epi_bytes = TH * 16 * 256 + (NT / 16) * 128 * 2 * 4;
    constexpr int smem = main_bytes > epi_bytes ? main_bytes : epi_bytes;
    static_assert(smem <= 160 * 1024, "LDS budget");
    static bool attr = false;
    if (!attr) {
        BNDM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_tap9<T, TH, ABL, NW>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr = true;
    }
    const int tiles_x = a.W / 16, tiles_y = a.H / TH, tps = tiles_x * tiles_y, ntn = a.Cout / 128;
    dim3 grid(a.B * tps * ntn);
    unsigned *dbg = nullptr;
    if constexpr ((ABL & 64) != 0) {
        static unsigned *buf = nullptr;
        if (!buf) BNDM_CHECK_HIP(hipMalloc(&buf, 8 * 9 * 6 * sizeof(unsigned)));
        dbg = buf;
    }
    hipLaunchKernelGGL((conv_tap9<T, TH, ABL, NW>), grid, dim3(NT), smem, st, a, tiles_x, tps, ntn, dbg);
    if constexpr ((ABL & 64) != 0) {
        // profiling aid: dump the marks of 4-chunk launches (the K = 2304 layers) as text
        int n9 = 0;
        for (int i = 0; i < a.nseg; ++i) n9 += a.seg[i].taps == 9 ? a.seg[i].C / 64 : 0;
        if (n9 == 4 && getenv("BNDM_TAP9_TRACE")) {
            unsigned h[8 * 9 * 6];
            BNDM_CHECK_HIP(hipStreamSynchronize(st));
            BNDM_CHECK_HIP(hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost));
            FILE *f = fopen(getenv("BNDM_TAP9_TRACE"), "w");
            if (f) {
                for (int w = 0; w < NW; ++w)
                    for (int t = 0; t < 9; ++t) {
                        fprintf(f, "w%d t%d", w, t);
                        for (int k = 0; k < 6; ++k) fprintf(f, " %u", h[(w * 9 + t) * 6 + k] - h[0]);
                        fprintf(f, "\n");
                    }
                fclose(f);
            }
        }
    }
    return launch_status("conv_tap9");
}

}  // namespace

// true when conv_tap9 can run this segment list: 3x3 segments first, 1x1 segments after them
bool conv_tap9_supports(const FusedArgs &a) {
    bool seen1 = false;
    int nchunks = 0;
    for (int i = 0; i < a.nseg; ++i) {
        if (a.seg[i].taps == 1) seen1 = true;
        else if (seen1) return false;
        else if ((a.seg[i].ss_off >= 0) != (a.seg[0].ss_off >= 0)) return false;   // all 3x3 segments normalised, or none
        nchunks += a.seg[i].C / 64;
    }
    // 32-bit buffer offsets: every source tensor (and the weight panel) must stay below 2 GiB
    for (int i = 0; i < a.nseg; ++i) {
        const long long px = a.seg[i].up ? (long long)(a.H / 2) * (a.W / 2) : (long long)a.H * a.W;
        if ((long long)a.B * px * a.seg[i].C * 2 >= (1LL << 31)) return false;
    }
    if ((long long)128 * a.Ktot * 2 >= (1LL << 31)) return false;
    return a.nseg >= 1 && a.seg[0].taps == 9 && nchunks <= TAP9_MAX_CHUNKS;
}

int launch_conv_tap9(int dtype, int TH, const FusedArgs &a, hipStream_t st) {
    static const int abl = getenv("BNDM_ABLATE") ? atoi(getenv("BNDM_ABLATE")) : 0;
    static const int nw = getenv("BNDM_TAP9_NW") ? atoi(getenv("BNDM_TAP9_NW")) : 8;      // waves per block: 8 or 4
    if (dtype == BNDM_DTYPE_F16) {
        if (abl && TH == 16 && !(abl == 64 && nw == 4)) {
            switch (abl) {
                case 1: return launch_tap9_t<_Float16, 16, 1>(a, st);
                case 2: return launch_tap9_t<_Float16, 16, 2>(a, st);
                case 4: return launch_tap9_t<_Float16, 16, 4>(a, st);
                case 8: return launch_tap9_t<_Float16, 16, 8>(a, st);
                case 10: return launch_tap9_t<_Float16, 16, 10>(a, st);
                case 14: return launch_tap9_t<_Float16, 16, 14>(a, st);
                case 15: return launch_tap9_t<_Float16, 16, 15>(a, st);
                case 16: return launch_tap9_t<_Float16, 16, 16>(a, st);
                case 64: return launch_tap9_t<_Float16, 16, 64>(a, st);
                case 128: return launch_tap9_t<_Float16, 16, 128>(a, st);
                case 80: return launch_tap9_t<_Float16, 16, 80>(a, st);
                case 17: return launch_tap9_t<_Float16, 16, 17>(a, st);
                case 19: return launch_tap9_t<_Float16, 16, 19>(a, st);
                case 21: return launch_tap9_t<_Float16, 16, 21>(a, st);
                case 23: return launch_tap9_t<_Float16, 16, 23>(a, st);
                case 9: return launch_tap9_t<_Float16, 16, 9>(a, st);
                case 11: return launch_tap9_t<_Float16, 16, 11>(a, st);
                case 13: return launch_tap9_t<_Float16, 16, 13>(a, st);
                default: break;
            }
        }
        if (abl == 128 && TH == 8) return launch_tap9_t<_Float16, 8, 128>(a, st);
        if (abl == 64 && nw == 4 && TH == 16) return launch_tap9_t<_Float16, 16, 64, 4>(a, st);
        if (nw == 4) return TH == 16 ? launch_tap9_t<_Float16, 16, 0, 4>(a, st) : launch_tap9_t<_Float16, 8, 0, 4>(a, st);
        return TH == 16 ? launch_tap9_t<_Float16, 16, 0>(a, st) : launch_tap9_t<_Float16, 8, 0>(a, st);
    }
    if (nw == 4) return TH == 16 ? launch_tap9_t<__bf16, 16, 0, 4>(a, st) : launch_tap9_t<__bf16, 8, 0, 4>(a, st);
    return TH == 16 ? launch_tap9_t<__bf16, 16, 0>(a, st) : launch_tap9_t<__bf16, 8, 0>(a, st);
}

}  // namespace bndm
