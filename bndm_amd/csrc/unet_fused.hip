// Fused stride-1 3x3 convolution for the FLOP-dominant UNet layers (ResnetBlock2D conv1 / conv2 +
// conv_shortcut, Upsample2D conv) -- see the interface comment in unet_kernels.hpp.
//
// Block = 512 threads (8 waves as 4(M) x 2(N)) computing a TH x 16 pixel tile of one sample x 128
// output channels.  K runs over 64-channel chunks of up to 4 segments; for every chunk the
// (TH+2) x 18 input halo patch is loaded ONCE into registers, GroupNorm scale/shift (+SiLU) is
// applied there, and the result is written to an XOR-swizzled LDS patch (double-buffered).  The 9 taps
// of the chunk then read their MFMA fragments from that patch at shifted pixel offsets, so activation
// traffic and the normalisation arithmetic are paid 1.27x instead of 9x.  Weight tiles ([128][64] per
// tap) stream through a 3-deep global_load_lds ring with counted vmcnt waits and one raw s_barrier per
// K-step.  The epilogue stages the tile through LDS: full 256-B rows are stored, and per-(sample,
// channel) sums / sums of squares of the stored 16-bit values are reduced deterministically for the
// GroupNorm that consumes this tensor next.
#include "unet_kernels.hpp"
#include "unet_types.hpp"
#include <algorithm>
#include <cstdlib>
#include <vector>

namespace bndm {
namespace {

// One int4 per K-step, built on the host (build_fused_steps) so the device loop carries no iterator
// state, no segment-table loads and almost no scalar bookkeeping:
//   x  weight k-offset (elements) of step s+3 (clamped)        -> weight DMA issued in the even phase of s
//   y  read descriptor of step s: tap offset (ky*18+kx) | patch buffer << 16
//   z  patch DMA: bit31 issue after the weight DMA (chunk with 8 steps of slack), bit30 issue before it
//      (raw 1x1 chunk needed next step) | segment << 24 | buffer << 23 | chunk
//   w  in-place normalisation: bit31 valid | bit30 first round (drain DMA) | segment << 24 | buffer << 23 |
//      round << 16 | chunk
struct StepDesc {
    int x, y, z, w;
};

// ABL: ablation switches for profiling only (results are wrong when non-zero):
//   1 skip the MFMAs, 2 skip in-loop weight DMA, 4 skip in-loop LDS fragment reads, 8 skip in-loop patch work
// NW: waves per block (8: 64x64 wave tiles, 2 waves per SIMD; 16: 32x64 wave tiles, 4 waves per SIMD)
template <typename T, int TH, int ABL, int NW>
__global__ __launch_bounds__(NW * 64) void conv_fused(const FusedArgs a, const StepDesc *__restrict__ steps,
                                                  const int tiles_x, const int tps, const int ntn,
                                                  const int nsteps) {
    using v8 = typename TT<T>::v8;
    using v4 = typename TT<T>::v4;
    constexpr int TW = 16, PW = TW + 2, PH = TH + 2;
    constexpr int NPP = PH * PW;                       // patch pixels
    constexpr int NPIECE = NPP * 8;                    // 16-byte pieces per patch chunk
    constexpr int NT = NW * 64;                        // threads per block
    constexpr int NROUND = (NPIECE + NT - 1) / NT;     // patch DMA rounds per chunk (the last one is partial)
    constexpr int NREMW = (NPIECE - (NROUND - 1) * NT + 63) / 64;    // waves that take part in the last round
    constexpr int PATCH_BYTES = (NROUND - 1) * NT * 16 + NREMW * 1024;
    constexpr int WAVES_M = NW / 2;                    // waves along M (2 along N)
    constexpr int NWP = 1024 / NT;                     // weight-tile DMAs per thread per K-step
    constexpr int BM = TH * TW;
    constexpr int TM = (TH * TW) / (WAVES_M * 32);     // 32-pixel MFMA tiles per wave along M
    static_assert(TM >= 1 && (TH % WAVES_M) == 0 && (TH / WAVES_M) % 2 == 0, "wave tiling");
    constexpr int TN = 2;
    constexpr bool WREG = (ABL & 64) != 0;             // weight tiles through registers instead of LDS-DMA
    constexpr int WSTAGES = 4;
    constexpr int W_BYTES = 128 * 128;
    constexpr int OFF_W = 2 * PATCH_BYTES;
    constexpr int OFF_SS = OFF_W + WSTAGES * W_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l = tid & 63;

    // ---- tile id (XCD-aware: neighbouring tiles of a sample share halos and weights) ---------------
    const int nblk = gridDim.x;
    int tix;
    {
        const int bid = blockIdx.x;
        const int xcd = bid & 7, q = nblk >> 3, r = nblk & 7;
        tix = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int mt = tix / ntn, nt = tix - mt * ntn;
    const int b = mt / tps, tin = mt - b * tps;
    const int ty = tin / tiles_x, tx = tin - ty * tiles_x;
    const int y0 = ty * TH, x0 = tx * TW, n0 = nt * 128;
    const int H = a.H, Wd = a.W;

    // ---- scale/shift table of this sample -> LDS ([0..ssC) scale, [ssC..2ssC) shift) ---------------
    float *ssL = reinterpret_cast<float *>(smem + OFF_SS);
    if (a.ss) {
        const float *g = a.ss + (size_t)b * 2 * a.ssC;
        for (int i = tid; i < 2 * a.ssC; i += NT) ssL[i] = g[i];
    }

    // ---- patch piece descriptors (independent of the chunk) ----------------------------------------
    int p_lds[NROUND], p_full[NROUND], p_half[NROUND], p_lc[NROUND];
    int p_valid = 0;                                   // bit r: piece of round r is inside the image
    int p_lcpack = 0;                                  // 3 bits per round: swizzled source chunk
#pragma unroll
    for (int r = 0; r < NROUND; ++r) {
        const int piece = r * NT + tid;
        const int pc = piece < NPIECE ? piece : NPIECE - 1;
        const int pp = pc >> 3, pch = pc & 7;
        const int pyy = pp / PW, pxx = pp - pyy * PW;
        const int iy = y0 - 1 + pyy, ix = x0 - 1 + pxx;
        const bool ok = piece < NPIECE && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)Wd;
        p_lds[r] = piece < NPIECE ? pp * 128 + pch * 16 : -1;
        // swizzle key = patch COLUMN pair: with it the 16 lanes of every ds_read_b128 lane group hit 16
        // distinct (pixel parity, chunk) bank classes for all nine tap shifts (a key on the linear pixel
        // index collides between the two image rows of an MFMA tile because the row stride is 18)
        p_lc[r] = pch ^ ((pxx >> 1) & 7);
        p_lcpack |= p_lc[r] << (3 * r);
        p_full[r] = ok ? (b * H + iy) * Wd + ix : -1;
        p_half[r] = ok ? (b * (H >> 1) + (iy >> 1)) * (Wd >> 1) + (ix >> 1) : -1;
        p_valid |= ok ? (1 << r) : 0;
    }
    // The patch of a chunk goes by LDS-DMA straight into its LDS buffer (piece index == LDS order, the
    // XOR swizzle sits on the source side).  Every thread issues exactly NROUND DMAs per chunk (tail
    // lanes re-read the zero page into a dead slot past the patch), so vmcnt bookkeeping is identical
    // in all waves and -- with no ordinary loads left in the loop -- stays a pure LDS-DMA count.
    // segment parameters by index without dynamic indexing of the kernel argument (that would be a
    // scalar memory load + lgkmcnt(0) in the middle of the loop)
    auto seg_of = [&](int si) {
        FusedSeg sg = a.seg[0];
        if (si == 1) sg = a.seg[1];
        if (si == 2) sg = a.seg[2];
        if (si == 3) sg = a.seg[3];
        return sg;
    };
    auto patch_dma = [&](int sidx, int chunk, int buf) {
        const FusedSeg sg = seg_of(sidx);
        const char *sbase = (const char *)sg.src;
        char *P = smem + buf * PATCH_BYTES;
#pragma unroll
        for (int r = 0; r < NROUND; ++r) {
            if (r == NROUND - 1 && w >= NREMW) break;     // wave-uniform: this wave owns no piece of the last round
            const int pix = sg.up ? p_half[r] : p_full[r];
            const char *src = pix >= 0 ? sbase + ((size_t)pix * sg.C + chunk * 64 + p_lc[r] * 8) * 2
                                       : (const char *)a.zeros;
            glds16(src, P + r * (NT * 16) + w * 1024);
        }
    };
    const bool last_round_wave = w < NREMW;            // wave-uniform
    // GroupNorm scale/shift + SiLU applied in place to this thread's own piece of a round.  Branch-free
    // (everything is derived from the wave-uniform round number plus the validity bitmask; padding and
    // tail pieces are rewritten unchanged -- the reference pads AFTER the activation) so that the
    // arithmetic can sit in the same basic block as the MFMAs and be interleaved with them.
    struct XfRegs {
        v8 v;
        f32x4 s0, s1, h0, h1;
        char *addr;
        bool valid;
    };
    auto xf_load = [&](int xd, XfRegs &x) {
        const int round = (xd >> 16) & 7, chunk = xd & 0xffff, buf = (xd >> 23) & 1;
        const int ss_off = seg_of((xd >> 24) & 3).ss_off;
        const int piece = round * NT + tid;
        const int lc = (p_lcpack >> (3 * round)) & 7;
        x.addr = smem + buf * PATCH_BYTES + piece * 16;
        x.valid = (p_valid >> round) & 1;
        x.v = *reinterpret_cast<const v8 *>(x.addr);
        const float *sc = ssL + ss_off + chunk * 64 + lc * 8;
        const float *sh = sc + a.ssC;
        x.s0 = *reinterpret_cast<const f32x4 *>(sc);
        x.s1 = *reinterpret_cast<const f32x4 *>(sc + 4);
        x.h0 = *reinterpret_cast<const f32x4 *>(sh);
        x.h1 = *reinterpret_cast<const f32x4 *>(sh + 4);
    };
    auto xf_math_store = [&](const XfRegs &x) {
        v8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float f = fmaf((float)x.v[e], e < 4 ? x.s0[e & 3] : x.s1[e & 3], e < 4 ? x.h0[e & 3] : x.h1[e & 3]);
            // SiLU with hardware exp2 / rcp (every normalised segment of this kernel is followed by SiLU)
            const float g = f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * f));
            o[e] = x.valid ? (T)g : x.v[e];
        }
        *reinterpret_cast<v8 *>(x.addr) = o;
    };

    // ---- weight tile staging: 128 rows x 128 B = 1024 pieces, 2 per thread ---------------------------
    const int wrow = tid >> 3;
    const int wlchunk = (tid & 7) ^ ((tid >> 4) & 7);
    const char *wsrc0 = (const char *)a.Wgt + ((size_t)(n0 + wrow) * a.Ktot + wlchunk * 8) * 2;
    const char *wsrc1 = wsrc0 + (size_t)64 * a.Ktot * 2;
    auto w_issue = [&](int buf, int kofs) {
        char *base = smem + OFF_W + buf * W_BYTES;
        glds16(wsrc0 + (size_t)kofs * 2, base + w * 1024);
        if (NWP == 2) glds16(wsrc1 + (size_t)kofs * 2, base + 8192 + w * 1024);
    };

    // register-staged alternative: tile s+2 is loaded to VGPRs in the even phase of step s and written to
    // LDS buffer (s+2)&1 ... one step later, in the even phase of step s+1 (as tile (s+1)+1)
    v8 wreg[NWP];
    auto w_load = [&](int kofs) {
        wreg[0] = *reinterpret_cast<const v8 *>(wsrc0 + (size_t)kofs * 2);
        if (NWP == 2) wreg[NWP - 1] = *reinterpret_cast<const v8 *>(wsrc1 + (size_t)kofs * 2);
    };
    auto w_store = [&](int buf) {
        char *base = smem + OFF_W + buf * W_BYTES;
        *reinterpret_cast<v8 *>(base + tid * 16) = wreg[0];
        if (NWP == 2) *reinterpret_cast<v8 *>(base + 8192 + tid * 16) = wreg[NWP - 1];
    };

    f32x16 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int wm = w % WAVES_M, wn = w / WAVES_M;
    const int q = l & 31, kh = l >> 5, wkey = (l >> 1) & 7;
    const int row_base = wm * (TH / WAVES_M);
    const int lr = q >> 4, lcx = q & 15;

    // ---- ping-pong schedule ------------------------------------------------------------------------
    // Waves 0-3 (group A) and 4-7 (group B) share the four SIMDs pairwise.  Every K-step has two phases,
    // each opened by a barrier:   even: A multiplies step s from registers | B reads step s from LDS
    //                             odd : A reads step s+1 from LDS          | B multiplies step s
    // so on every SIMD one wave feeds the matrix pipe while its partner does LDS / DMA work.
    // Weight tile s+2 is issued in the even phase of step s (3-deep ring; a counted vmcnt before the odd
    // barrier certifies tile s+1).  The patch of the next chunk is DMA'd at tap 0 of the current one and
    // normalised in place one round per step (taps 1..NROUND), all by the thread that issued the piece.
    // Everything that varies per step comes from the host-built StepDesc table (one s_load per step).
    v8 fa[4][TN], fb[4][TM];
    // per-lane address constants
    const int wconst = (wn * 64 + q) * 128 + ((kh ^ wkey) << 4);       // ks folded in by xor (ks << 5)
    int pr0[TM];
#pragma unroll
    for (int j = 0; j < TM; ++j) pr0[j] = (row_base + 2 * j + lr) * PW + lcx;
    const int c0x = kh << 4;
    auto read_frags = [&](int rd, int wbuf) {
        const int tapoff = rd & 0xff, kx = (rd >> 8) & 3, pb = rd >> 16;
        const char *P = smem + pb * PATCH_BYTES;
        const char *Wt = smem + OFF_W + wbuf * W_BYTES;
        const int kc = ((((lcx + kx) >> 1) & 7) << 4) ^ c0x;
        int xb[TM];
#pragma unroll
        for (int j = 0; j < TM; ++j) xb[j] = ((pr0[j] + tapoff) << 7) | kc;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
            for (int i = 0; i < TN; ++i)
                fa[ks][i] = *reinterpret_cast<const v8 *>(Wt + ((wconst ^ (ks << 5)) + i * 4096));
#pragma unroll
            for (int j = 0; j < TM; ++j) fb[ks][j] = *reinterpret_cast<const v8 *>(P + (xb[j] ^ (ks << 5)));
        }
    };
    auto multiply = [&]() {
        if (ABL & 1) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
                for (int i = 0; i < TN; ++i) asm volatile("" ::"v"(fa[ks][i]));
#pragma unroll
                for (int j = 0; j < TM; ++j) asm volatile("" ::"v"(fb[ks][j]));
            }
            return;
        }
        if (!(ABL & 32)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j) acc[i][j] = TT<T>::mfma(fa[ks][i], fb[ks][j], acc[i][j]);
        if (!(ABL & 32)) __builtin_amdgcn_s_setprio(0);
    };
    // DMA work of the even phase of a step (returns true if a slack patch DMA followed the weight DMA)
    // returns 0: only the weight tile, 1: a slack patch DMA followed it, 2: a late 1x1 patch DMA preceded it
    auto even_dma = [&](const StepDesc &d, int wnext) -> int {
        int kind = 0;
        if (!(ABL & 8) && (d.z & (1 << 30))) {
            patch_dma((d.z >> 24) & 3, d.z & 0xffff, (d.z >> 23) & 1);
            kind = 2;
        }
        asm volatile("" ::: "memory");
        if (!WREG && !(ABL & 2)) w_issue(wnext, d.x);
        asm volatile("" ::: "memory");
        if (!(ABL & 8) && d.z < 0) {
            patch_dma((d.z >> 24) & 3, d.z & 0xffff, (d.z >> 23) & 1);
            asm volatile("" ::: "memory");
            kind = 1;
        }
        return kind;
    };

    const bool grpA = wn == 0;                         // wave-uniform (w is an SGPR)
    StepDesc dcur = steps[0];
    StepDesc dnext = steps[nsteps > 1 ? 1 : 0];

    // ---- prologue: chunk 0 and the first WSTAGES-1 weight tiles in flight together; normalise chunk 0 ----
    patch_dma(0, 0, 0);
    asm volatile("" ::: "memory");
    if (WREG) {
        w_load(steps[nsteps].x);
        w_store(0);
        w_load(steps[nsteps].y);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        w_issue(0, steps[nsteps].x);                    // the table's extra entry carries the k-offsets of steps 0..2
        w_issue(1, steps[nsteps].y);
        w_issue(2, steps[nsteps].z);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * NWP) : "memory");    // own patch pieces landed (weights may still fly)
    }
    __syncthreads();                                   // ss table visible (written above by plain stores)
    if (!(ABL & 256) && a.seg[0].ss_off >= 0) {
        // rounds are independent: load them all, then transform (latencies overlap)
        XfRegs xr[NROUND];
#pragma unroll
        for (int r = 0; r < NROUND; ++r)
            if (r < NROUND - 1 || last_round_wave) xf_load(r << 16, xr[r]);
#pragma unroll
        for (int r = 0; r < NROUND; ++r)
            if (r < NROUND - 1 || last_round_wave) xf_math_store(xr[r]);
    }
    if (WREG) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * NWP) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    int wcur = 0, wnext = WSTAGES - 1;
    int kx_prev = steps[nsteps].z;                      // k-offset of weight tile s+2 (register-staged path)
    if (grpA) {
        read_frags(dcur.y, 0);
        for (int s = 0; s < nsteps; ++s) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            // descriptor of step s+2: issued now so that its latency hides under this phase
            const StepDesc dn2 = steps[s + 2 < nsteps ? s + 2 : nsteps - 1];
            // (rounds start two taps after the patch DMA: only one younger weight tile is in flight then)
            const bool dox = !(ABL & 8) && dcur.w < 0 && (last_round_wave || ((dcur.w >> 16) & 7) != NROUND - 1);
            if (WREG) {
                // tile s+1 (loaded one step ago) -> LDS, then start loading tile s+2
                if (s + 1 < nsteps) w_store((s + 1) & 1);
                if (dox && (dcur.w & (1 << 30))) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // patch DMA landed
                if (s + 2 < nsteps) w_load(kx_prev);
                kx_prev = dcur.x;
            } else if (!(ABL & 16) && dox && (dcur.w & (1 << 30))) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NWP) : "memory");   // patch DMA landed
            }
            const int issued = even_dma(dcur, wnext);
            if (dox) {
                // normalisation arithmetic rides in the shadow of the matrix pipe
                XfRegs x;
                xf_load(dcur.w, x);
                if (ABL & 1) {
                    multiply();
                    xf_math_store(x);
                } else {
                    // no s_setprio here: it would fence the scheduler and keep the VALU work behind the MFMAs
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                        for (int i = 0; i < TN; ++i)
#pragma unroll
                            for (int j = 0; j < TM; ++j) acc[i][j] = TT<T>::mfma(fa[ks][i], fb[ks][j], acc[i][j]);
                    xf_math_store(x);
#pragma unroll
                    for (int g = 0; g < 4 * TN * TM; ++g) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);    // 1 MFMA
                        __builtin_amdgcn_sched_group_barrier(0x002, TM == 2 ? 4 : 8, 0);    // VALU in its shadow
                    }
                }
            } else {
                multiply();
            }
            // weight tile s+1 landed: WSTAGES-2 younger tiles (+ a patch DMA issued after them) may still fly
            if (WREG) {
                if (issued == 2) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // late 1x1 patch needed now
                else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            } else if (ABL & 16) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // profiling: never wait for DMA
            else if (issued == 0) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NWP * (WSTAGES - 2)) : "memory");
            else if (issued == 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NWP) : "memory");   // late patch: only the newest tile may fly
            else if (last_round_wave) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NWP * (WSTAGES - 2) + NROUND) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NWP * (WSTAGES - 2) + NROUND - 1) : "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            const int wn1 = wcur + 1 == WSTAGES ? 0 : wcur + 1;
            if (!(ABL & 4) && s + 1 < nsteps) read_frags(dnext.y, WREG ? ((s + 1) & 1) : wn1);
            dcur = dnext;
            dnext = dn2;
            wcur = wn1;
            wnext = wnext + 1 == WSTAGES ? 0 : wnext + 1;
        }
    } else {
        for (int s = 0; s < nsteps; ++s) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            const StepDesc dn2 = steps[s + 2 < nsteps ? s + 2 : nsteps - 1];
            const bool dox = !(ABL & 8) && dcur.w < 0 && (last_round_wave || ((dcur.w >> 16) & 7) != NROUND - 1);
            if (WREG) {
                // tile s+1 (loaded one step ago) -> LDS, then start loading tile s+2
                if (s + 1 < nsteps) w_store((s + 1) & 1);
                if (dox && (dcur.w & (1 << 30))) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // patch DMA landed
                if (s + 2 < nsteps) w_load(kx_prev);
                kx_prev = dcur.x;
            } else if (!(ABL & 16) && dox && (dcur.w & (1 << 30))) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NWP) : "memory");   // patch DMA landed
            }
            const int issued = even_dma(dcur, wnext);
            if (dox) {
                // fragment reads first, normalisation arithmetic while they are in flight
                XfRegs x;
                xf_load(dcur.w, x);
                if (!(ABL & 4)) read_frags(dcur.y, WREG ? (s & 1) : wcur);
                xf_math_store(x);
            } else {
                if (!(ABL & 4)) read_frags(dcur.y, WREG ? (s & 1) : wcur);
            }
            // weight tile s+1 landed: WSTAGES-2 younger tiles (+ a patch DMA issued after them) may still fly
            if (WREG) {
                if (issued == 2) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // late 1x1 patch needed now
                else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            } else if (ABL & 16) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // profiling: never wait for DMA
            else if (issued == 0) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NWP * (WSTAGES - 2)) : "memory");
            else if (issued == 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NWP) : "memory");   // late patch: only the newest tile may fly
            else if (last_round_wave) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NWP * (WSTAGES - 2) + NROUND) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NWP * (WSTAGES - 2) + NROUND - 1) : "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            multiply();
            dcur = dnext;
            dnext = dn2;
            wcur = wcur + 1 == WSTAGES ? 0 : wcur + 1;
            wnext = wnext + 1 == WSTAGES ? 0 : wnext + 1;
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();                                   // all LDS reads and tail DMAs done: reuse LDS

    if (ABL & 128) return;                             // profiling: no epilogue
    // ---- epilogue 1: bias + time embedding -> 16-bit tile in LDS ([BM][128 ch], swizzled) ---------------
    // all per-channel addends are fetched up front (16 independent loads) instead of inside the store loop
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
int launch_fused_t(const FusedArgs &a, hipStream_t st) {
    constexpr int NT = NW * 64;
    constexpr int NPIECE = (TH + 2) * 18 * 8, NROUND = (NPIECE + NT - 1) / NT;
    constexpr int PATCH_BYTES = (NROUND - 1) * NT * 16 + ((NPIECE - (NROUND - 1) * NT + 63) / 64) * 1024;
    constexpr int main_bytes = 2 * PATCH_BYTES + 4 * 16384 + 8192;
    constexpr int epi_bytes = TH * 16 * 256 + (NT / 16) * 128 * 2 * 4;
    constexpr int smem = main_bytes > epi_bytes ? main_bytes : epi_bytes;
    static bool attr = false;
    if (!attr) {
        BNDM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_fused<T, TH, ABL, NW>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr = true;
    }
    const int tiles_x = a.W / 16, tiles_y = a.H / TH, tps = tiles_x * tiles_y, ntn = a.Cout / 128;
    int nsteps = 0;
    for (int i = 0; i < a.nseg; ++i) nsteps += a.seg[i].taps * (a.seg[i].C / 64);
    dim3 grid(a.B * tps * ntn);
    hipLaunchKernelGGL((conv_fused<T, TH, ABL, NW>), grid, dim3(NT), smem, st, a, (const StepDesc *)a.steps, tiles_x, tps,
                       ntn, nsteps);
    return launch_status("conv_fused");
}

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
    if (prow < RP) {
        f32x4 add0 = {0.f, 0.f, 0.f, 0.f}, add1 = {0.f, 0.f, 0.f, 0.f};
        if (from_slabs) {
            if (sl.bias) {
                add0 = *reinterpret_cast<const f32x4 *>(sl.bias + c0);
                add1 = *reinterpret_cast<const f32x4 *>(sl.bias + c0 + 4);
            }
        }
        for (int p = prow; p < HW; p += RP) {
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
        }
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
    if (prow < RP)
        for (int p = prow; p < HW; p += RP) {
            const v8 v = *reinterpret_cast<const v8 *>(src + (size_t)p * Cs);
            v8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float f = fmaf((float)v[e], ss[cl + e], ss[Cb + cl + e]);
                if (silu) f = f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * f));
                o[e] = (T)f;
            }
            *reinterpret_cast<v8 *>(out + ((size_t)b * HW + p) * C + c0) = o;
        }
}

}  // namespace

// Host: the per-step schedule of conv_fused for a segment list (see StepDesc).  Returns nsteps + 1
// entries; the extra last entry carries the weight k-offsets of steps 0 and 1 for the prologue.
std::vector<int> build_fused_steps(const FusedSeg *seg, int nseg, int TH, int nthreads) {
    const int PW = 18;
    const int NROUND = ((TH + 2) * 18 * 8 + nthreads - 1) / nthreads;
    struct St { int seg, chunk, tap, taps, cidx, kofs; };
    std::vector<St> st;
    int koff = 0, cidx = 0;
    for (int i = 0; i < nseg; ++i) {
        for (int c = 0; c < seg[i].C / 64; ++c, ++cidx)
            for (int t = 0; t < seg[i].taps; ++t)
                st.push_back(St{i, c, t, seg[i].taps, cidx, koff + (seg[i].taps == 9 ? t * seg[i].C : 0) + c * 64});
        koff += seg[i].taps * seg[i].C;
    }
    const int n = (int)st.size(), nchunks = cidx;
    std::vector<int> out((size_t)(n + 1) * 4, 0);
    std::vector<int> first(nchunks + 1, n);             // first step of every chunk
    for (int s = n - 1; s >= 0; --s) first[st[s].cidx] = s;
    for (int s = 0; s < n; ++s) {
        const St &c = st[s];
        int *d = &out[(size_t)s * 4];
        d[0] = st[std::min(s + 3, n - 1)].kofs;          // weight tile issued at step s: ring depth 4
        const int ky = c.taps == 9 ? c.tap / 3 : 1, kx = c.taps == 9 ? c.tap % 3 : 1;
        d[1] = (ky * PW + kx) | (kx << 8) | ((c.cidx & 1) << 16);
        // chunk DMA'd at tap 0 of its 9-tap predecessor
        if (c.taps == 9 && c.tap == 0 && c.cidx + 1 < nchunks) {
            const St &nx = st[first[c.cidx + 1]];
            d[2] = (int)(0x80000000u | (nx.seg << 24) | ((nx.cidx & 1) << 23) | nx.chunk);
        }
        // raw 1x1 chunk that follows a 1x1 chunk: issued one step ahead, before the weight DMA
        if (c.taps == 1 && s + 1 < n && st[s + 1].cidx != c.cidx) {
            const St &nx = st[s + 1];
            d[2] |= (1 << 30) | (nx.seg << 24) | ((nx.cidx & 1) << 23) | nx.chunk;
        }
        // in-place normalisation rounds of the next chunk at taps 2..NROUND+1 of a 9-tap chunk
        if (c.taps == 9 && c.tap >= 2 && c.tap <= NROUND + 1 && c.cidx + 1 < nchunks) {
            const St &nx = st[first[c.cidx + 1]];
            if (seg[nx.seg].ss_off >= 0)
                d[3] = (int)(0x80000000u | (c.tap == 2 ? (1 << 30) : 0) | (nx.seg << 24) | ((nx.cidx & 1) << 23) |
                             ((c.tap - 2) << 16) | nx.chunk);
        }
    }
    out[(size_t)n * 4 + 0] = st[0].kofs;
    out[(size_t)n * 4 + 1] = st[std::min(1, n - 1)].kofs;
    out[(size_t)n * 4 + 2] = st[std::min(2, n - 1)].kofs;
    return out;
}

int conv_fused_tiles_per_sample(int TH, int H, int W) { return (H / TH) * (W / 16); }

// threads per block of the variant used for tile height TH (BNDM_FUSED_NW=8|16 overrides the default)
int conv_fused_threads(int TH) {
    static const int nw = getenv("BNDM_FUSED_NW") ? atoi(getenv("BNDM_FUSED_NW")) : 8;
    return (TH == 16 && nw == 16) ? 1024 : 512;
}

int launch_conv_fused(int dtype, int TH, const FusedArgs &a, hipStream_t st) {
    if ((TH != 8 && TH != 16) || a.H % TH || a.W % 16 || a.Cout % 128 || a.nseg < 1 || a.nseg > CONV_MAX_SEG) {
        set_error("launch_conv_fused: unsupported shape TH=%d H=%d W=%d Cout=%d nseg=%d", TH, a.H, a.W, a.Cout,
                  a.nseg);
        return BNDM_E_ARG;
    }
    for (int i = 0; i < a.nseg; ++i)
        if (a.seg[i].C % 64 || (a.seg[i].taps != 9 && a.seg[i].taps != 1) || (a.seg[i].ss_off >= 0 && !a.ss)) {
            set_error("launch_conv_fused: bad segment %d", i);
            return BNDM_E_ARG;
        }
    for (int i = 1; i < a.nseg; ++i)
        if (a.seg[i].ss_off >= 0 && a.seg[i - 1].taps == 1) {
            set_error("launch_conv_fused: a normalised segment may not follow a 1x1 segment");
            return BNDM_E_ARG;
        }
    if (a.ss && a.ssC > 1024) {
        set_error("launch_conv_fused: scale/shift table of %d channels exceeds 1024", a.ssC);
        return BNDM_E_ARG;
    }
    static const int ver = getenv("BNDM_FUSED_V") ? atoi(getenv("BNDM_FUSED_V")) : 9;
    if (ver == 9 && conv_tap9_supports(a) && conv_fused_threads(TH) == 512) {
        static const int spec = getenv("BNDM_TAP9_SPEC") ? atoi(getenv("BNDM_TAP9_SPEC")) : 0;
        return spec ? launch_conv_tap9s(dtype, TH, a, st) : launch_conv_tap9(dtype, TH, a, st);
    }
    static const int abl = getenv("BNDM_ABLATE") ? atoi(getenv("BNDM_ABLATE")) : 0;
    if (dtype == BNDM_DTYPE_F16) {
        if (abl && TH == 16) {       // profiling-only variants (f16, 256-pixel tiles)
            switch (abl) {
                case 1: return launch_fused_t<_Float16, 16, 1>(a, st);
                case 2: return launch_fused_t<_Float16, 16, 2>(a, st);
                case 4: return launch_fused_t<_Float16, 16, 4>(a, st);
                case 8: return launch_fused_t<_Float16, 16, 8>(a, st);
                case 7: return launch_fused_t<_Float16, 16, 7>(a, st);
                case 14: return launch_fused_t<_Float16, 16, 14>(a, st);
                case 11: return launch_fused_t<_Float16, 16, 11>(a, st);
                case 13: return launch_fused_t<_Float16, 16, 13>(a, st);
                case 10: return launch_fused_t<_Float16, 16, 10>(a, st);
                case 16: return launch_fused_t<_Float16, 16, 16>(a, st);
                case 32: return launch_fused_t<_Float16, 16, 32>(a, st);
                case 64: return launch_fused_t<_Float16, 16, 64>(a, st);
                case 143: return launch_fused_t<_Float16, 16, 143>(a, st);
                case 271: return launch_fused_t<_Float16, 16, 271>(a, st);
                case 399: return launch_fused_t<_Float16, 16, 399>(a, st);
                case 24: return launch_fused_t<_Float16, 16, 24>(a, st);
                case 15: return launch_fused_t<_Float16, 16, 15>(a, st);
                default: break;
            }
        }
        if (TH == 16 && conv_fused_threads(16) == 1024) return launch_fused_t<_Float16, 16, 0, 16>(a, st);
        return TH == 16 ? launch_fused_t<_Float16, 16, 0>(a, st) : launch_fused_t<_Float16, 8, 0>(a, st);
    }
    if (TH == 16 && conv_fused_threads(16) == 1024) return launch_fused_t<__bf16, 16, 0, 16>(a, st);
    return TH == 16 ? launch_fused_t<__bf16, 16, 0>(a, st) : launch_fused_t<__bf16, 8, 0>(a, st);
}

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
