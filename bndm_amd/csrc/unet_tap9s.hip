// conv_tap9s: conv_tap9 with specialised waves.  Same tiling, LDS layout, K order and epilogue as conv_tap9
// (unet_tap9.hip), but the block has 12 waves:
//   waves 0..7  (consumers, 4(M) x 2(N) of 64x64): fragment reads, MFMAs, and the weight-tile DMA of the ring --
//               nothing else, so their instruction streams stay MFMA-dense;
//   waves 8..11 (producers, one per SIMD): the patch DMA of the next chunk and its in-place GroupNorm + SiLU.
// The three waves of a SIMD are {w, w+4, w+8}: two consumers and one producer, and the hardware interleaves the
// producer's VALU / transcendental work with the consumers' MFMAs.  One s_barrier per K-step for all 12 waves, at
// the same place as in conv_tap9 (B_t between the consumers' phases 1 and 2):
//   consumers reach B_t with vmcnt(4) lgkmcnt(0) (tile t+1 landed; the tiles of G_(t-2), G_(t-1) may fly) and
//             issue tile t+4 after it;
//   producers issue the next chunk's patch rounds after B_0 / B_1, normalise two rounds per step after
//             B_3 .. B_7 (vmcnt(NROUND-1-r) certifies round r: their only DMAs are patch rounds, in order) and
//             reach B_8 with lgkmcnt(0), which publishes the patch.
#include "unet_kernels.hpp"
#include "unet_types.hpp"
#include <cstdlib>

namespace bndm {
namespace {

template <int N> struct IC {
    static constexpr int value = N;
};
typedef uint32_t u32x4t __attribute__((ext_vector_type(4)));
constexpr int TAP9S_MAX_CHUNKS = 64;

struct Chunk {
    const void *src;
    int bytes, soff, C2, up, ssbase, kbase, kstride;
};

template <typename T, int TH>
__global__ __launch_bounds__(768) void conv_tap9s(const FusedArgs a, const int tiles_x, const int tps, const int ntn) {
    using v8 = typename TT<T>::v8;
    using v4 = typename TT<T>::v4;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    constexpr int TW = 16, PW = TW + 2, PH = TH + 2;
    constexpr int NPIECE = PH * PW * 8;                // 16-byte pieces per patch chunk
    constexpr int NT = 512;                            // consumer threads (and the epilogue's thread count)
    constexpr int NP = 256;                            // producer threads
    constexpr int NROUND = (NPIECE + NP - 1) / NP;     // patch rounds per chunk and producer thread
    constexpr int NREMW = (NPIECE - (NROUND - 1) * NP + 63) / 64;    // producer waves with pieces in the last round
    constexpr int DUMP_OFF = (NROUND - 1) * NP * 16 + NREMW * 1024;
    constexpr int PATCH_BYTES = DUMP_OFF + (NREMW < 4 ? 1024 : 0);
    constexpr int NFULL = NROUND - (NREMW < 4 ? 1 : 0);
    static_assert(NFULL <= 10, "normalisation schedule: two full rounds after each of B_3 .. B_7");
    constexpr int BM = TH * TW;
    constexpr int TM = BM / (4 * 32), TN = 2;
    static_assert(TM >= 1 && (TH / 4) % 2 == 0, "wave tiling");
    constexpr int WSTAGES = 4, W_BYTES = 128 * 128;
    constexpr int OFF_W = 2 * PATCH_BYTES;
    constexpr int OFF_SS = OFF_W + WSTAGES * W_BYTES;
    constexpr int OFF_TAB = OFF_SS + 8192;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l = tid & 63;
    const bool producer = w >= 8;                      // wave-uniform
    const int pw = w - 8, ptid = tid - NT;

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


    // ---- producers: patch piece descriptors (piece = round * 256 + ptid) ---------------------------------
    int p_full[NROUND];
    int p_valid = 0;
    uint64_t p_lcpack = 0;
#pragma unroll
    for (int r = 0; r < NROUND; ++r) {
        const int piece = r * NP + (producer ? ptid : 0);
        const int pc = piece < NPIECE ? piece : NPIECE - 1;
        const int pp = pc >> 3, pch = pc & 7;
        const int pyy = pp / PW, pxx = pp - pyy * PW;
        const int iy = y0 - 1 + pyy, ix = x0 - 1 + pxx;
        const bool ok = piece < NPIECE && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)Wd;
        p_full[r] = ok ? (b * H + iy) * Wd + ix : -1;
        p_valid |= ok ? (1 << r) : 0;
        p_lcpack |= (uint64_t)(pch ^ ((pxx >> 1) & 7)) << (3 * r);
    }
    auto patch_dma = [&](auto rc, const Chunk &c, int buf) {
        constexpr int r = decltype(rc)::value;
        int pix = p_full[r];
        if (c.up) {
            const int ix = pix & (Wd - 1), iy = (pix >> lgW) & (H - 1), bb = pix >> (lgW + lgH);
            pix = pix < 0 ? -1 : ((((bb << (lgH - 1)) + (iy >> 1)) << (lgW - 1)) + (ix >> 1));
        }
        const int lc16 = (int)((p_lcpack >> (3 * r)) & 7) << 4;
        const unsigned voff = (unsigned)(pix * c.C2 + lc16);
        char *dst = smem + buf * PATCH_BYTES + ((r < NROUND - 1 || pw < NREMW) ? r * (NP * 16) + pw * 1024 : DUMP_OFF);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(uniform_rsrc(c.src, c.bytes), (lds_ptr_t)dst, 16, voff,
                                                 __builtin_amdgcn_readfirstlane(c.soff), 0, 0);
    };
    const float *ssL = reinterpret_cast<const float *>(smem + OFF_SS);
    // GroupNorm scale/shift + SiLU of one round, in place, by the thread that issued the piece (padding and tail
    // pieces are rewritten unchanged: the reference pads after the activation)
    auto xf_round = [&](auto rc, const Chunk &c, int buf) {
        constexpr int r = decltype(rc)::value;
        if (r == NROUND - 1 && pw >= NREMW) return;                      // wave-uniform: no piece in the partial round
        char *addr = smem + buf * PATCH_BYTES + r * (NP * 16) + ptid * 16;
        const u32x4 x = *reinterpret_cast<const u32x4 *>(addr);
        const int lc = (int)((p_lcpack >> (3 * r)) & 7);
        const bool valid = (p_valid >> r) & 1;
        const float *sc = ssL + c.ssbase + lc * 8;
        const f32x4 s0 = *reinterpret_cast<const f32x4 *>(sc), s1 = *reinterpret_cast<const f32x4 *>(sc + 4);
        const f32x4 h0 = *reinterpret_cast<const f32x4 *>(sc + a.ssC), h1 = *reinterpret_cast<const f32x4 *>(sc + a.ssC + 4);
        const v8 vin = __builtin_bit_cast(v8, x);
        v8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float f = fmaf((float)vin[e], e < 4 ? s0[e & 3] : s1[e & 3], e < 4 ? h0[e & 3] : h1[e & 3]);
            o[e] = (T)(f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * f)));
        }
        const u32x4 ou = __builtin_bit_cast(u32x4, o);
        u32x4 res;
#pragma unroll
        for (int e = 0; e < 4; ++e) res[e] = valid ? ou[e] : x[e];
        *reinterpret_cast<u32x4 *>(addr) = res;
    };

    // ---- consumers: weight tiles, fragment addresses, accumulators ----------------------------------------
    const __amdgpu_buffer_rsrc_t wrs = uniform_rsrc((const char *)a.Wgt + (size_t)n0 * a.Ktot * 2, 128 * a.Ktot * 2);
    const unsigned wvoff0 = (unsigned)(((tid >> 3) * a.Ktot + (((tid & 7) ^ ((tid >> 4) & 7)) << 3)) * 2);
    const unsigned wvstep = (unsigned)(64 * a.Ktot * 2);
    auto w_issue = [&](int slot, int kofs) {
        char *base = smem + OFF_W + slot * W_BYTES + w * 1024;
        const int so = __builtin_amdgcn_readfirstlane(kofs * 2);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lds_ptr_t)base, 16, wvoff0, so, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lds_ptr_t)(base + 8192), 16, wvoff0 + wvstep, so, 0, 0);
    };
    f32x16 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const int wm = w & 3, wn = (w >> 2) & 1;
    const int q = l & 31, kh = l >> 5;
    const int row_base = wm * (TH / 4);
    const int lr = q >> 4, lcx = q & 15;
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
#pragma unroll
        for (int i = 0; i < TN; ++i) fa[set][i] = *reinterpret_cast<const v8 *>(smem + wa[ks] + i * 4096);
#pragma unroll
        for (int j = 0; j < TM; ++j)
            fb[set][j] = *reinterpret_cast<const v8 *>(smem + pa[kx][ks] + (ky * PW + kx + 2 * j * PW) * 128);
    };
    auto multiply = [&](auto sc) {
        constexpr int set = decltype(sc)::value;
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j) acc[i][j] = TT<T>::mfma(fa[set][i], fb[set][j], acc[i][j]);
    };
    auto for_rounds = [&](auto self, auto rc, auto f) {          // f(IC<r>) for r = rc .. NROUND-1
        constexpr int r = decltype(rc)::value;
        if constexpr (r < NROUND) {
            f(rc);
            self(self, IC<r + 1>{}, f);
        }
    };

    // ---- prologue -----------------------------------------------------------------------------------------
    Chunk cur = make_chunk(0, 0);
    Chunk nxt = cur;
    if (!producer) {
        const __amdgpu_buffer_rsrc_t srs =
            uniform_rsrc(a.ss ? a.ss + (size_t)b * 2 * a.ssC : (const float *)a.zeros, a.ss ? 2 * a.ssC * 4 : 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srs, (lds_ptr_t)(smem + OFF_SS + w * 1024), 16, (unsigned)(tid * 16), 0,
                                                 0, 0);
        w_issue(0, cur.kbase);
        w_issue(1, cur.kbase + cur.kstride);
        w_issue(2, cur.kbase + 2 * cur.kstride);
        w_issue(3, cur.kbase + 3 * cur.kstride);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");          // scale/shift table landed
    } else {
        for_rounds(for_rounds, IC<0>{}, [&](auto rc) { patch_dma(rc, cur, 0); });
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // own patch pieces landed
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // chunk table written
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (producer) {
        if (cur.ssbase >= 0) for_rounds(for_rounds, IC<0>{}, [&](auto rc) { xf_round(rc, cur, 0); });
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (nchunk9 > 1) nxt = load_chunk(1);

    // ---- 3x3 chunks ---------------------------------------------------------------------------------------
    int slot = 0, pbuf = 0;
    auto advance_chunk = [&](int c) {
        pbuf ^= 1;
        cur = nxt;
        nxt = load_chunk(c + 2 < nchunk9 ? c + 2 : nchunk9 - 1);
    };
    if (!producer) {
        read_frags(IC<0>{}, IC<0>{}, IC<0>{});
        read_frags(IC<0>{}, IC<1>{}, IC<1>{});
        for (int c = 0; c < nchunk9; ++c) {
            auto step = [&](auto tc) {
                constexpr int t = decltype(tc)::value;
                constexpr int p0 = 4 * t;
                read_frags(tc, IC<2>{}, IC<(p0 + 2) % 3>{});
                multiply(IC<p0 % 3>{});
                read_frags(tc, IC<3>{}, IC<(p0 + 3) % 3>{});
                multiply(IC<(p0 + 1) % 3>{});
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
                    advance_chunk(c);
                }
                asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                read_frags(IC<(t + 1) % 9>{}, IC<0>{}, IC<(p0 + 4) % 3>{});
                {
                    const int kofs = t + 4 < 9 ? cur.kbase + (t + 4) * cur.kstride
                                     : t == 8  ? cur.kbase + 3 * cur.kstride
                                               : nxt.kbase + (t + 4 - 9) * nxt.kstride;
                    w_issue((slot + 3) & (WSTAGES - 1), kofs);
                }
                multiply(IC<(p0 + 2) % 3>{});
                read_frags(IC<(t + 1) % 9>{}, IC<1>{}, IC<(p0 + 5) % 3>{});
                multiply(IC<(p0 + 3) % 3>{});
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
        }
    } else {
        for (int c = 0; c < nchunk9; ++c) {
            const bool has_next = c + 1 < nchunk9;
            const bool dox = has_next && nxt.ssbase >= 0;
            auto pstep = [&](auto tc) {
                constexpr int t = decltype(tc)::value;
                if constexpr (t == 8) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // every normalised piece is written
                    advance_chunk(c);
                }
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                if constexpr (t == 8) return;
                const int nb = pbuf ^ 1;                                  // patch buffer of the next chunk
                if (has_next) {
                    // rounds 0..5 after B_0, the rest after B_1
                    if constexpr (t == 0 || t == 1) {
                        constexpr int R0 = t == 0 ? 0 : 6, R1 = t == 0 ? (NROUND < 6 ? NROUND : 6) : NROUND;
                        for_rounds(for_rounds, IC<R0>{}, [&](auto rc) {
                            if constexpr (decltype(rc)::value < R1) patch_dma(rc, nxt, nb);
                        });
                    }
                    // two rounds after each of B_3 .. B_7 (+ the partial round after B_7)
                    if constexpr (t >= 3 && t <= 7) {
                        constexpr int ra = 2 * (t - 3), rb = ra + 1;
                        if (dox) {
                            if constexpr (ra < NFULL) {
                                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NROUND - 1 - ra) : "memory");
                                xf_round(IC<ra>{}, nxt, nb);
                            }
                            if constexpr (rb < NFULL) {
                                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NROUND - 1 - (rb < NROUND ? rb : 0)) : "memory");
                                xf_round(IC<(rb < NROUND ? rb : 0)>{}, nxt, nb);
                            }
                            if constexpr (t == 7 && NFULL < NROUND) {
                                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                                xf_round(IC<NROUND - 1>{}, nxt, nb);
                            }
                        }
                    }
                }
            };
            pstep(IC<0>{});
            pstep(IC<1>{});
            pstep(IC<2>{});
            pstep(IC<3>{});
            pstep(IC<4>{});
            pstep(IC<5>{});
            pstep(IC<6>{});
            pstep(IC<7>{});
            pstep(IC<8>{});
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    // ---- 1x1 chunks (raw centre pixels): producers fetch the patch, consumers the weight tile and multiply ----
    if (nchunk1 > 0) {
        Chunk c1 = load_chunk(nchunk9);
        auto issue1 = [&](const Chunk &c, int buf) {
            if (producer) for_rounds(for_rounds, IC<0>{}, [&](auto rc) { patch_dma(rc, c, buf); });
            else w_issue(buf, c.kbase);
        };
        issue1(c1, 0);
        for (int n = 0; n < nchunk1; ++n) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            const Chunk c2 = load_chunk(nchunk9 + (n + 1 < nchunk1 ? n + 1 : nchunk1 - 1));
            if (n + 1 < nchunk1) issue1(c2, (n + 1) & 1);
            const int buf = n & 1;
            if (!producer) {
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
            }
            c1 = c2;
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }

    // ---- epilogue (consumers; the producers only keep the barriers company) ------------------------------
    char *stg = smem;
    f32x4 addv[TN][4];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) addv[i][g] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (a.bias && !producer) {
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                addv[i][g] = *reinterpret_cast<const f32x4 *>(a.bias + n0 + wn * 64 + i * 32 + 8 * g + 4 * kh);
    }
    if (a.temb && !producer) {
        const float *tembp = a.temb + (size_t)b * a.temb_bstride + a.temb_off;
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                addv[i][g] += *reinterpret_cast<const f32x4 *>(tembp + n0 + wn * 64 + i * 32 + 8 * g + 4 * kh);
    }
    if (!producer)
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
    if (a.resid && !producer) {
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
    if (!producer)
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
        if (!producer)
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


template <typename T, int TH>
int launch_tap9s_t(const FusedArgs &a, hipStream_t st) {
    constexpr int NP = 256;
    constexpr int NPIECE = (TH + 2) * 18 * 8, NROUND = (NPIECE + NP - 1) / NP;
    constexpr int NREMW = (NPIECE - (NROUND - 1) * NP + 63) / 64;
    constexpr int PATCH_BYTES = (NROUND - 1) * NP * 16 + NREMW * 1024 + (NREMW < 4 ? 1024 : 0);
    constexpr int main_bytes = 2 * PATCH_BYTES + 4 * 16384 + 8192 + TAP9S_MAX_CHUNKS * 32;
    constexpr int epi_bytes = TH * 16 * 256 + (512 / 16) * 128 * 2 * 4;
    constexpr int smem = main_bytes > epi_bytes ? main_bytes : epi_bytes;
    static_assert(smem <= 160 * 1024, "LDS budget");
    static bool attr = false;
    if (!attr) {
        BNDM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_tap9s<T, TH>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr = true;
    }
    const int tiles_x = a.W / 16, tiles_y = a.H / TH, tps = tiles_x * tiles_y, ntn = a.Cout / 128;
    dim3 grid(a.B * tps * ntn);
    hipLaunchKernelGGL((conv_tap9s<T, TH>), grid, dim3(768), smem, st, a, tiles_x, tps, ntn);
    return launch_status("conv_tap9s");
}

}  // namespace

int launch_conv_tap9s(int dtype, int TH, const FusedArgs &a, hipStream_t st) {
    if (dtype == BNDM_DTYPE_F16) return TH == 16 ? launch_tap9s_t<_Float16, 16>(a, st) : launch_tap9s_t<_Float16, 8>(a, st);
    return TH == 16 ? launch_tap9s_t<__bf16, 16>(a, st) : launch_tap9s_t<__bf16, 8>(a, st);
}

}  // namespace bndm
