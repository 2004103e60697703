// UNet2DModel forward kernels for gfx950 (diffusers UNet2DModel as built at iadb_bn.py:205-282).
//
//   conv_igemm      implicit-GEMM convolution on v_mfma_f32_32x32x16_{f16,bf16}: NHWC 16-bit in,
//                   fp32 accumulate.  K runs over up to 4 "segments" (source tensor x taps), which
//                   expresses cat([h, skip]) inputs, the fused 1x1 conv_shortcut of ResnetBlock2D,
//                   stride-2 Downsample2D and nearest-2x Upsample2D without materialising anything.
//                   Both operands are staged by 16-byte global_load_lds into XOR-swizzled,
//                   double-buffered LDS tiles ([rows][64 k] of 128 B; chunk ^= (row>>1)&7 makes the
//                   ds_read_b128 fragment reads conflict-free); padding taps read a zero page.
//                   D = W . X^T, so a lane owns 4 consecutive output channels of one pixel and the
//                   epilogue (bias + time-embedding + residual) stores 8 B per lane into NHWC.
//   conv_in         3x3 conv from the fp32 NCHW sample (K = 9*Cin <= 64) straight onto MFMA.
//   gn_*            GroupNorm(32): slab partial sums -> per-(sample, channel) scale/shift -> apply(+SiLU)
//   attention       softmax(q k^T / sqrt(8)) v for 8-wide heads over <= a few hundred tokens
//   temb_mlp        sinusoidal Timesteps -> Linear -> SiLU -> Linear -> SiLU (fp32)
#include "unet_kernels.hpp"
#include "unet_types.hpp"
#include <cmath>

namespace bndm {
namespace {

// ------------------------------------------------------------------------------------------------
// implicit-GEMM convolution
// ------------------------------------------------------------------------------------------------
struct KIter {
    int seg, tap, chunk, nchunk, taps;
};

__device__ __forceinline__ void kiter_load(KIter &k, const ConvArgs &a) {
    k.nchunk = a.seg[k.seg].C >> 6;
    k.taps = a.seg[k.seg].taps;
}
__device__ __forceinline__ void kiter_init(KIter &k, const ConvArgs &a, int ks) {
    k.seg = 0;
    for (;;) {
        const int n = a.seg[k.seg].taps * (a.seg[k.seg].C >> 6);
        if (ks < n || k.seg == a.nseg - 1) break;
        ks -= n;
        ++k.seg;
    }
    kiter_load(k, a);
    k.tap = ks / k.nchunk;
    k.chunk = ks - k.tap * k.nchunk;
}
__device__ __forceinline__ void kiter_next(KIter &k, const ConvArgs &a) {
    if (++k.chunk == k.nchunk) {
        k.chunk = 0;
        if (++k.tap == k.taps) {
            k.tap = 0;
            if (k.seg < a.nseg - 1) ++k.seg;
            kiter_load(k, a);
        }
    }
}

// FAST: no segment is upsampled -> per-step addressing comes from the host-built step table
// (ConvArgs::steps: {segment | tap << 8, pixel delta dy*Ws+dx, channel offset, -}) plus a per-piece base
// pixel and 9-bit tap-validity mask computed once; the generic path walks segments/taps in-kernel.
template <typename T, int WAVES_M, int WAVES_N, int TM, int TN, int EPI, int STAGES, bool FAST>
__global__ __launch_bounds__(WAVES_M *WAVES_N * 64) void conv_igemm(const ConvArgs a, const int ksteps,
                                                                     const int logW, const int logH,
                                                                     const int ntm, const int ntn) {
    constexpr int NT = WAVES_M * WAVES_N * 64;   // threads per block
    constexpr int RPI = NT / 8;                  // tile rows covered by one block-wide glds instruction
    constexpr int BM = WAVES_M * TM * 32;        // output pixels per block
    constexpr int BN = WAVES_N * TN * 32;        // output channels per block
    constexpr int X_BYTES = BM * 128;
    constexpr int W_BYTES = BN * 128;
    constexpr int STAGE = X_BYTES + W_BYTES;
    constexpr int NXP = BM / RPI;                // 16-B pieces per thread per stage (activations)
    constexpr int NWP = BN / RPI;                // (weights)
    constexpr int NLD = NXP + NWP;               // glds instructions per thread per stage
    static_assert(BM % RPI == 0 && BN % RPI == 0, "tile rows must be a multiple of the staging stride");
    static_assert(STAGES >= 2 && NLD * (STAGES - 1) <= 63, "vmcnt field is 6 bits");
    using v8 = typename TT<T>::v8;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l = tid & 63;
    // profiling aid (BNDM_IGEMM_TRACE): block (0, 0) records s_memtime marks of its first 24 K-steps and of its life
    const bool tracing = a.counters != nullptr && blockIdx.x == 0 && blockIdx.y == 0;
    auto xmark = [&](int k) {
        if (tracing) {
            const unsigned tm = (unsigned)__builtin_amdgcn_s_memtime();
            if (l == 0) a.counters[16 * 24 * 5 + w * 4 + k] = tm;
        }
    };
    xmark(0);

    // ---- XCD-aware tile id: consecutive tiles (shared halos / shared weights) stay on one XCD -----
    const int nblk = ntm * ntn;
    int tix;
    {
        const int bid = blockIdx.x;
        const int xcd = bid & 7, q = nblk >> 3, r = nblk & 7;
        tix = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int mt = a.wmajor ? tix % ntm : tix / ntn, nt = a.wmajor ? tix / ntm : tix - mt * ntn;
    const int m0 = mt * BM, n0 = nt * BN;
    const int M = a.B << (logW + logH);
    const int H = a.H, Wd = a.W;

    // ---- split-K range ------------------------------------------------------------------------------
    int ks_begin = 0, ks_end = ksteps;
    if (a.splitk > 1) {
        const int per = (ksteps + a.splitk - 1) / a.splitk;
        ks_begin = blockIdx.y * per;
        ks_end = min(ksteps, ks_begin + per);
    }
    const int nsteps = max(0, ks_end - ks_begin);

    // ---- per-thread staging descriptors -----------------------------------------------------------
    const int prow = tid >> 3;                       // 0..RPI-1
    const int lchunk = (tid & 7) ^ ((tid >> 4) & 7); // logical 16-B chunk this lane fetches
    int px[NXP], py[NXP], pb[NXP];
#pragma unroll
    for (int i = 0; i < NXP; ++i) {
        const int m = m0 + prow + RPI * i;
        px[i] = m & (Wd - 1);
        py[i] = (m >> logW) & (H - 1);
        pb[i] = (m < M) ? (m >> (logW + logH)) : -1;
    }
    const char *wsrc[NWP];
    const size_t wstep = a.wtiled ? (size_t)BN * 128 : 128;          // bytes between consecutive K-steps
#pragma unroll
    for (int i = 0; i < NWP; ++i)
        wsrc[i] = a.wtiled ? (const char *)a.Wgt + ((size_t)nt * ksteps * BN + prow + RPI * i) * 128 + (tid & 7) * 16
                           : (const char *)a.Wgt + ((size_t)(n0 + prow + RPI * i) * a.Ktot + lchunk * 8) * 2;
    // FAST path: centre-tap source pixel and tap-validity mask of every piece.  The per-step part of the address (tap
    // shift, channel chunk) is a scalar from the host-built step table; the source is read through a buffer descriptor
    // whose base sits one row + one pixel BEFORE the tensor, so that every tap shift is a non-negative scalar offset, and
    // padding taps get an out-of-range offset (the descriptor returns zeros): one multiply-add and one select per piece.
    int pbase[NXP], vmask[NXP];
    const int Hs = H * a.stride, Ws = Wd * a.stride;
    if (FAST) {
#pragma unroll
        for (int i = 0; i < NXP; ++i) {
            const int cy = py[i] * a.stride, cx = px[i] * a.stride;
            pbase[i] = pb[i] >= 0 ? (pb[i] * Hs + cy) * Ws + cx : 0;
            int vm = 0;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int dy = t / 3 - 1, dx = t % 3 - 1;
                const bool ok = (unsigned)(cy + dy) < (unsigned)Hs && (unsigned)(cx + dx) < (unsigned)Ws && pb[i] >= 0;
                vm |= ok ? (1 << t) : 0;
            }
            vmask[i] = vm;
        }
    }
    // The step table is read through the constant address space: a uniform load from plain global memory is emitted as
    // a VECTOR load + readfirstlane, and the s_waitcnt vmcnt(0) in front of that readfirstlane drains every LDS-DMA
    // stage in flight -- the ring would run one stage deep.  As s_load it only touches lgkmcnt.
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(4))) i32x4 *const_int4_ptr;
    const const_int4_ptr stab = (const_int4_ptr)a.steps;
    // segment descriptors live in SGPRs (selected by value): indexing the kernel argument dynamically would put a
    // scalar load + lgkmcnt(0) in front of every stage
    static_assert(CONV_MAX_SEG == 4, "segment select below is written out for four segments");
#define BNDM_SEG_SGPRS(i)                                                                          \
    const uint32_t sg_lo##i = __builtin_amdgcn_readfirstlane((uint32_t)(uint64_t)a.seg[i].src);    \
    const uint32_t sg_hi##i = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)a.seg[i].src >> 32)); \
    const uint32_t sg_c2##i = __builtin_amdgcn_readfirstlane(a.seg[i].C * 2);
    BNDM_SEG_SGPRS(0) BNDM_SEG_SGPRS(1) BNDM_SEG_SGPRS(2) BNDM_SEG_SGPRS(3)
#undef BNDM_SEG_SGPRS
    // the descriptor of the NEXT staged step is fetched while the current one is used, so the scalar load
    // never sits right in front of an lgkmcnt wait
    i32x4 dsc = {0, 0, 0, 0};
    if (FAST) dsc = stab[min(ks_begin, ksteps - 1)];
    auto stage_fast = [&](int buf, int ks) __attribute__((always_inline)) {
        char *base = smem + buf * STAGE;
        const int kc = min(ks, ksteps - 1);
        const i32x4 d = dsc;
        dsc = stab[min(ks + 1, ksteps - 1)];
        const int si = d.x & 0xff;
        // scalar selects written as instructions: a C++ select over lambda-captured values is folded into an indexed
        // load from the closure object, which then keeps the closure AND the whole argument struct in scratch
        uint32_t blo = sg_lo0, bhi = sg_hi0, C2 = sg_c20;
#define BNDM_SSEL(k)                                                                                           \
    asm("s_cmp_eq_u32 %3, " #k "\n\ts_cselect_b32 %0, %4, %0\n\ts_cselect_b32 %1, %5, %1\n\ts_cselect_b32 %2, %6, %2" \
        : "+s"(blo), "+s"(bhi), "+s"(C2)                                                                      \
        : "s"(si), "s"(sg_lo##k), "s"(sg_hi##k), "s"(sg_c2##k)                                                \
        : "scc");
        BNDM_SSEL(1) BNDM_SSEL(2) BNDM_SSEL(3)
#undef BNDM_SSEL
        const int tap = (d.x >> 8) & 0xf;
        const int lead = (Ws + 1) * (int)C2;                             // bytes of one row + one pixel
        const __amdgpu_buffer_rsrc_t rs =
            uniform_rsrc((const char *)(((uint64_t)bhi << 32) | blo) - lead, a.B * Hs * Ws * (int)C2 + lead);
        const int soff = __builtin_amdgcn_readfirstlane(d.w);            // (tap shift + row + pixel) * C2 + chunk * 128
#pragma unroll
        for (int i = 0; i < NXP; ++i) {
            const bool ok = (vmask[i] >> tap) & 1;
            const unsigned voff = ok ? (unsigned)(pbase[i] * (int)C2 + lchunk * 16) : 0x80000000u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(base + i * (RPI * 128) + w * 1024), 16, voff, soff, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NWP; ++i)
            glds16(wsrc[i] + (size_t)kc * wstep, base + X_BYTES + i * (RPI * 128) + w * 1024);
    };

    auto stage = [&](int buf, const KIter &k, int ks) __attribute__((always_inline)) {
        char *base = smem + buf * STAGE;
        const ConvSeg sg = a.seg[k.seg];
        int dy = 0, dx = 0;
        if (sg.taps == 9) {
            dy = k.tap / 3 - 1;
            dx = k.tap - (dy + 1) * 3 - 1;
        }
        // unified addressing: coordinate c = p*mul + d is bounds-checked against lim (the upsampled /
        // strided extent), the source pixel is c >> sh in an Hs x Ws tensor
        const int sh = sg.up ? 1 : 0;
        const int mul = sg.up ? 1 : a.stride;
        const int limY = sg.up ? H : H * a.stride, limX = sg.up ? Wd : Wd * a.stride;
        const int Hs = limY >> sh, Ws = limX >> sh;
        const int coff = k.chunk * 64 + lchunk * 8;
        const char *sbase = (const char *)sg.src;
#pragma unroll
        for (int i = 0; i < NXP; ++i) {
            const int cy = py[i] * mul + dy, cx = px[i] * mul + dx;
            const bool ok = (unsigned)cy < (unsigned)limY && (unsigned)cx < (unsigned)limX && pb[i] >= 0;
            const long off = ((long)((pb[i] * Hs + (cy >> sh)) * Ws + (cx >> sh)) * sg.C + coff) * 2;
            const char *src = ok ? sbase + off : (const char *)a.zeros;
            glds16(src, base + i * (RPI * 128) + w * 1024);
        }
#pragma unroll
        for (int i = 0; i < NWP; ++i)
            glds16(wsrc[i] + (size_t)min(ks, ksteps - 1) * wstep, base + X_BYTES + i * (RPI * 128) + w * 1024);
    };

    f32x16 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int wm = w % WAVES_M, wn = w / WAVES_M;
    const int frow = l & 31, kh = l >> 5, key = (l >> 1) & 7;

    // ---- STAGES-deep ring: stages s+1 .. s+STAGES-1 are in flight while step s is multiplied.
    // Every thread issues exactly NLD loads per stage, so "stage s has landed" is a counted wait on the
    // stages issued after it; near the end of the K range fewer stages are in flight and the count
    // shrinks accordingly (no dummy stages are issued).  One raw s_barrier per K-step both publishes
    // stage s to all waves and retires the buffer that the next issue overwrites.
    KIter kit;
    if (!FAST) kiter_init(kit, a, ks_begin);
    int issued = 0;
    // the descriptors of the prologue's stages (and of the first in-loop stage) are fetched together: one scalar-load
    // latency instead of one per stage
    i32x4 dpre[STAGES];
    if (FAST) {
#pragma unroll
        for (int p = 0; p < STAGES; ++p) dpre[p] = stab[min(ks_begin + p, ksteps - 1)];
    }
#pragma unroll
    for (int p = 0; p < STAGES - 1; ++p) {
        if (issued < nsteps) {
            if (FAST) {
                dsc = dpre[p];
                stage_fast(p, ks_begin + issued);
            } else {
                stage(p, kit, ks_begin + issued);
                if (issued + 1 < nsteps) kiter_next(kit, a);
            }
            ++issued;
        }
    }
    int cur = 0, nxt = STAGES - 1;
    constexpr bool dephase = WAVES_M * WAVES_N >= 8;
    auto mark = [&](int it, int k) {
        if (tracing && it < 24) {
            const unsigned tm = (unsigned)__builtin_amdgcn_s_memtime();
            if (l == 0) a.counters[(w * 24 + it) * 5 + k] = tm;
        }
    };
    for (int it = 0; it < nsteps; ++it) {
        mark(it, 0);
        const int younger = issued - it - 1;           // stages in flight behind the one needed now (uniform)
        if (younger >= STAGES - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD * (STAGES - 2)) : "memory");
        else if (STAGES > 3 && younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        mark(it, 1);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        mark(it, 2);
        // The two waves of a SIMD are de-phased: the first issues the next stage BEFORE its MFMAs, its partner AFTER
        // them, so that one wave's MFMAs cover the other's address/issue time (all waves in lock-step would serialise
        // the 32 KiB of LDS-DMA issue and the MFMA work of every K-step).
        auto issue_next = [&]() __attribute__((always_inline)) {
            if (issued < nsteps) {
                if (FAST) {
                    stage_fast(nxt, ks_begin + issued);
                } else {
                    stage(nxt, kit, ks_begin + issued);
                    if (issued + 1 < nsteps) kiter_next(kit, a);
                }
                ++issued;
            }
        };
        const bool late = dephase && w >= (WAVES_M * WAVES_N) / 2;
        if (!late) issue_next();
        mark(it, 3);
        const char *Xt = smem + cur * STAGE;
        const char *Wt = Xt + X_BYTES;
        // all fragment reads of the K-step are issued up front; the MFMAs of slice s start as soon as its own
        // reads have returned (in-order lgkmcnt) while the later slices are still in flight
        v8 af[4][TN], bf[4][TM];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int pc = ((2 * s + kh) ^ key) * 16;
#pragma unroll
            for (int i = 0; i < TN; ++i)
                af[s][i] = *reinterpret_cast<const v8 *>(Wt + (wn * TN * 32 + i * 32 + frow) * 128 + pc);
#pragma unroll
            for (int j = 0; j < TM; ++j)
                bf[s][j] = *reinterpret_cast<const v8 *>(Xt + (wm * TM * 32 + j * 32 + frow) * 128 + pc);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j) acc[i][j] = TT<T>::mfma(af[s][i], bf[s][j], acc[i][j]);
        if (late) issue_next();
        mark(it, 4);
        cur = cur + 1 == STAGES ? 0 : cur + 1;
        nxt = nxt + 1 == STAGES ? 0 : nxt + 1;
    }

    xmark(1);
    // ---- epilogue: lane owns pixel (l&31) of each M-tile and channels 8g + 4*kh + {0..3} -----------
    const int HW = 1 << (logW + logH);
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int m = m0 + wm * TM * 32 + j * 32 + frow;
        if (m >= M) continue;
        const int b = m >> (logW + logH);
#pragma unroll
        for (int i = 0; i < TN; ++i) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int co = n0 + wn * TN * 32 + i * 32 + 8 * g + 4 * kh;
                if (co >= a.Cout) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g + e];
                if (EPI == EPI_NCHW32) {
                    float *o = (float *)a.out;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (co + e < a.Cout)
                            o[((size_t)b * a.Cout + co + e) * HW + (m & (HW - 1))] =
                                v[e] + (a.bias ? a.bias[co + e] : 0.f);
                    continue;
                }
                if (a.splitk <= 1) {
                    if (a.bias) {
                        const f32x4 bv = *reinterpret_cast<const f32x4 *>(a.bias + co);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += bv[e];
                    }
                    if (a.temb) {
                        const f32x4 tv = *reinterpret_cast<const f32x4 *>(a.temb + (size_t)b * a.temb_bstride +
                                                                           a.temb_off + co);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += tv[e];
                    }
                }
                if (EPI == EPI_F32_ROWS) {
                    float *o = (float *)a.out + (a.splitk > 1 ? (size_t)blockIdx.y * M * a.Cout : 0);
                    f32x4 ov = {v[0], v[1], v[2], v[3]};
                    *reinterpret_cast<f32x4 *>(o + (size_t)m * a.Cout + co) = ov;
                } else {
                    using v4 = typename TT<T>::v4;
                    if (a.resid) {
                        const v4 rv = *reinterpret_cast<const v4 *>((const T *)a.resid + (size_t)m * a.Cout + co);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += (float)rv[e];
                    }
                    v4 ov;
#pragma unroll
                    for (int e = 0; e < 4; ++e) ov[e] = (T)v[e];
                    *reinterpret_cast<v4 *>((T *)a.out + (size_t)m * a.Cout + co) = ov;
                }
            }
        }
    }

    if (tracing) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        xmark(2);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float *__restrict__ part, int splitk,
                                                            const ConvArgs a, int logHW) {
    using v4 = typename TT<T>::v4;
    const size_t M = (size_t)a.B << logHW;
    const int C4 = a.Cout >> 2;
    const size_t total = M * C4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const size_t m = i / C4;
        const int co = (int)(i - m * C4) * 4;
        f32x4 s = *reinterpret_cast<const f32x4 *>(part + m * a.Cout + co);
        for (int z0 = 1; z0 < splitk; z0 += 8) {                 // eight slabs in flight; sums stay in z order
            f32x4 u[8];
#pragma unroll
            for (int z = 0; z < 8; ++z)
                if (z0 + z < splitk) u[z] = *reinterpret_cast<const f32x4 *>(part + ((size_t)(z0 + z) * M + m) * a.Cout + co);
#pragma unroll
            for (int z = 0; z < 8; ++z)
                if (z0 + z < splitk) s += u[z];
        }
        if (a.bias) s += *reinterpret_cast<const f32x4 *>(a.bias + co);
        if (a.temb) {
            const size_t b = m >> logHW;
            s += *reinterpret_cast<const f32x4 *>(a.temb + b * a.temb_bstride + a.temb_off + co);
        }
        if (a.resid) {
            const v4 rv = *reinterpret_cast<const v4 *>((const T *)a.resid + m * a.Cout + co);
#pragma unroll
            for (int e = 0; e < 4; ++e) s[e] += (float)rv[e];
        }
        v4 ov;
#pragma unroll
        for (int e = 0; e < 4; ++e) ov[e] = (T)s[e];
        *reinterpret_cast<v4 *>((T *)a.out + m * a.Cout + co) = ov;
    }
}

// ------------------------------------------------------------------------------------------------
// conv_in: fp32 NCHW -> NHWC 16-bit, 3x3 pad 1, K = 9*Cin padded to KP (multiple of 16, <= 64)
// one wave per 32 consecutive pixels; W16 is [C0][KP] 16-bit (k = ci*9 + ky*3 + kx)
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void conv_in_kernel(const float *__restrict__ x, int Cx,
                                                      const float *__restrict__ extra, int Ce,
                                                      const T *__restrict__ W16, const float *__restrict__ bias,
                                                      T *__restrict__ out, float *__restrict__ stats, int B,
                                                      int logH, int logW, int C0, int KP) {
    // block = 128 consecutive pixels of one sample (H*W is a multiple of 128); wave = 32 pixels.
    // The fp32 NCHW input rows the block touches (its rows +-1, zero padded) are first copied to LDS as
    // 16-bit with coalesced loads; every lane then builds its MFMA operand (27..54 taps) from LDS.
    // The output tile is staged in LDS too: rows are stored 16 B per lane and the per-channel sums for
    // the first GroupNorm come out of the same pass.
    using v8 = typename TT<T>::v8;
    using v4 = typename TT<T>::v4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, l = tid & 63, wv = tid >> 6;
    const int H = 1 << logH, Wd = 1 << logW, HW = H * Wd;
    const int m0 = blockIdx.x * 128;
    const int b = m0 >> (logW + logH);
    const int Cin = Cx + Ce;
    const int kh = l >> 5;
    // ---- input patch: rows [ya-1, yb+1] x cols [xa-1, xb+1] of every input channel ----------------------
    const int p0 = m0 & (HW - 1);
    const int ya = p0 >> logW, yb = (p0 + 127) >> logW;                 // first / last image row of the block
    const int xa = Wd >= 128 ? (p0 & (Wd - 1)) : 0;                     // a block is part of one row when W >= 128
    const int pw = (Wd >= 128 ? 128 : Wd) + 2, ph = yb - ya + 3;
    T *patch = reinterpret_cast<T *>(smem);                             // [Cin][ph][pw]
    for (int i = tid; i < Cin * ph * pw; i += 256) {
        const int px_ = i % pw, t = i / pw, py_ = t % ph, ci = t / ph;
        const int iy = ya - 1 + py_, ix = xa - 1 + px_;
        float v = 0.f;
        if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)Wd)
            v = ci < Cx ? x[((size_t)b * Cx + ci) * HW + iy * Wd + ix]
                        : extra[((size_t)b * Ce + (ci - Cx)) * HW + iy * Wd + ix];
        patch[i] = (T)v;
    }
    __syncthreads();
    const int m = m0 + wv * 32 + (l & 31);
    const int xx = m & (Wd - 1), yy = (m >> logW) & (H - 1);
    const int nks = KP >> 4;
    v8 bf[4];
    const int pbase = (yy - ya) * pw + (xx - xa);
#pragma unroll
    for (int s = 0; s < 4; ++s) {                      // fully unrolled: bf[] must stay in registers
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 16 * s + 8 * kh + j;
            const int kc = k < 9 * Cin ? k : 0;
            const int ci = kc / 9, t = kc - ci * 9, dy = t / 3, dx = t - dy * 3;        // dy, dx in 0..2
            const T v = patch[(ci * ph + dy) * pw + dx + pbase];
            bf[s][j] = k < 9 * Cin ? v : (T)0.f;
        }
    }
    __syncthreads();                                                   // patch is dead: the tile is staged over it
    const int rowB = C0 * 2;                       // bytes per staged pixel row
    const int pl = wv * 32 + (l & 31);             // pixel inside the block
    // the 16-byte chunks of a staged row are XOR-swizzled with the pixel index; the key must stay inside the row's C0 / 8
    // chunks (8 at C0 = 64: with a 4-bit key pixel 15's chunks landed in pixel 16's row -- found by tests/gfx950sim)
    const int swz = ((C0 >> 3) - 1) & 15;
    for (int n0 = 0; n0 < C0; n0 += 32) {
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s)
            if (s < nks) {
                const v8 af = *reinterpret_cast<const v8 *>(W16 + (size_t)(n0 + (l & 31)) * KP + 16 * s + 8 * kh);
                acc = TT<T>::mfma(af, bf[s], acc);
            }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int co = n0 + 8 * g + 4 * kh;
            const f32x4 bv = *reinterpret_cast<const f32x4 *>(bias + co);
            v4 ov;
#pragma unroll
            for (int e = 0; e < 4; ++e) ov[e] = (T)(acc[4 * g + e] + bv[e]);
            *reinterpret_cast<v4 *>(smem + pl * rowB + ((((co >> 3) ^ (pl & swz)) << 4) | ((co & 7) * 2))) = ov;
        }
    }
    __syncthreads();
    const int CH = C0 >> 3, RP = 256 / CH;         // 16-B chunks per row, pixel rows per pass
    const int c16 = tid % CH, prw = tid / CH;
    float s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
    for (int p = prw; p < 128; p += RP) {
        const v8 v = *reinterpret_cast<const v8 *>(smem + p * rowB + ((c16 ^ (p & swz)) << 4));
        store_wt(reinterpret_cast<v8 *>(out + (size_t)(m0 + p) * C0 + c16 * 8), v);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float f = (float)v[e];
            s1[e] += f;
            s2[e] = fmaf(f, f, s2[e]);
        }
    }
    if (stats) {
        float *red = reinterpret_cast<float *>(smem + 128 * rowB);     // [RP][C0][2]
        if (prw < RP)                                                  // (256 % CH != 0 would leave spare threads)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            red[((prw * C0) + c16 * 8 + e) * 2 + 0] = s1[e];
            red[((prw * C0) + c16 * 8 + e) * 2 + 1] = s2[e];
        }
        __syncthreads();
        const int nslab = HW >> 7, b0 = m0 >> (logW + logH), slab = (m0 & (HW - 1)) >> 7;
        for (int i = tid; i < 2 * C0; i += 256) {
            float t = 0.f;
            for (int r = 0; r < RP; ++r) t += red[r * C0 * 2 + i];
            store_wt(stats + ((size_t)(b0 * nslab + slab) * C0) * 2 + i, t);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// GroupNorm
// ------------------------------------------------------------------------------------------------
// partial[((b*nslab + slab)*C + c)*2 + {0,1}] = sum / sum of squares over the slab's pixels
template <typename T>
__global__ __launch_bounds__(256) void gn_stats_kernel(const T *__restrict__ x1, int C1, const T *__restrict__ x2,
                                                       int C2, int HW, float *__restrict__ partial, int nslab) {
    using v8 = typename TT<T>::v8;
    __shared__ float red[256 * 16];
    const int C = C1 + C2, CH = C >> 3;          // 16-byte chunks per pixel (<= 256)
    const int slab = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x;
    const int chunk = tid % CH, prow = tid / CH, RP = 256 / CH;
    const int pps = HW / nslab;
    float s[8], ss[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = ss[e] = 0.f;
    const int c0 = chunk * 8;
    const T *src;
    int Cs;
    if (c0 < C1) { src = x1 + c0; Cs = C1; } else { src = x2 + (c0 - C1); Cs = C2; }
    if (prow < RP) {
        for (int p = slab * pps + prow; p < (slab + 1) * pps; p += RP) {
            const v8 v = *reinterpret_cast<const v8 *>(src + ((size_t)b * HW + p) * Cs);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float f = (float)v[e];
                s[e] += f;
                ss[e] = fmaf(f, f, ss[e]);
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        red[tid * 16 + e] = s[e];
        red[tid * 16 + 8 + e] = ss[e];
    }
    __syncthreads();
    if (tid < CH) {
        for (int r = 1; r < RP; ++r)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                s[e] += red[(r * CH + tid) * 16 + e];
                ss[e] += red[(r * CH + tid) * 16 + 8 + e];
            }
        float *o = partial + ((size_t)(b * nslab + slab) * C + tid * 8) * 2;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            o[2 * e] = s[e];
            o[2 * e + 1] = ss[e];
        }
    }
}

__global__ __launch_bounds__(256) void gn_finalize_kernel(const float *__restrict__ partial, int nslab, int HW,
                                                          int C, int groups, float eps,
                                                          const float *__restrict__ gamma,
                                                          const float *__restrict__ beta,
                                                          float *__restrict__ scale_shift) {
    __shared__ double cs[1024], css[1024];
    const int b = blockIdx.x;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        double s = 0, q = 0;
        for (int k = 0; k < nslab; ++k) {
            const float *p = partial + ((size_t)(b * nslab + k) * C + c) * 2;
            s += p[0];
            q += p[1];
        }
        cs[c] = s;
        css[c] = q;
    }
    __syncthreads();
    const int Cg = C / groups;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const int g0 = (c / Cg) * Cg;
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

template <typename T>
__global__ __launch_bounds__(256) void gn_apply_kernel(const T *__restrict__ x1, int C1, const T *__restrict__ x2,
                                                       int C2, const float *__restrict__ scale_shift, int HW,
                                                       size_t total_chunks, int silu, T *__restrict__ out) {
    using v8 = typename TT<T>::v8;
    const int C = C1 + C2, CH = C >> 3;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total_chunks;
         i += (size_t)gridDim.x * blockDim.x) {
        const size_t m = i / CH;
        const int c0 = (int)(i - m * CH) * 8;
        const size_t b = m / HW;
        const v8 v = c0 < C1 ? *reinterpret_cast<const v8 *>(x1 + m * C1 + c0)
                             : *reinterpret_cast<const v8 *>(x2 + m * C2 + (c0 - C1));
        const float *sc = scale_shift + (b * 2) * C + c0;
        const float *sh = sc + C;
        const f32x4 s0 = *reinterpret_cast<const f32x4 *>(sc), s1 = *reinterpret_cast<const f32x4 *>(sc + 4);
        const f32x4 h0 = *reinterpret_cast<const f32x4 *>(sh), h1 = *reinterpret_cast<const f32x4 *>(sh + 4);
        v8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float f = fmaf((float)v[e], e < 4 ? s0[e & 3] : s1[e & 3], e < 4 ? h0[e & 3] : h1[e & 3]);
            if (silu) f = silu_f(f);
            o[e] = (T)f;
        }
        *reinterpret_cast<v8 *>(out + m * C + c0) = o;
    }
}

// ------------------------------------------------------------------------------------------------
// attention core: one thread per (sample, head, query)
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void attention_kernel(const T *__restrict__ qkv, T *__restrict__ out, int B, int Tn,
                                                        int C) {
    using v8 = typename TT<T>::v8;
    const int heads = C >> 3;
    const size_t total = (size_t)B * heads * Tn;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int h = (int)(idx % heads);
    const size_t bt = idx / heads;          // b*T + tq
    const size_t b = bt / Tn;
    const v8 qv = *reinterpret_cast<const v8 *>(qkv + bt * 3 * C + h * 8);
    float q[8], o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        q[e] = (float)qv[e] * 0.35355339059327373f;   // head_dim ** -0.5
        o[e] = 0.f;
    }
    float mx = -INFINITY, den = 0.f;
    for (int tk = 0; tk < Tn; ++tk) {
        const T *row = qkv + (b * Tn + tk) * 3 * C + h * 8;
        const v8 kv = *reinterpret_cast<const v8 *>(row + C);
        const v8 vv = *reinterpret_cast<const v8 *>(row + 2 * C);
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) s = fmaf(q[e], (float)kv[e], s);
        const float nm = fmaxf(mx, s);
        const float corr = __expf(mx - nm), p = __expf(s - nm);
        den = den * corr + p;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = o[e] * corr + p * (float)vv[e];
        mx = nm;
    }
    const float inv = 1.0f / den;
    v8 ov;
#pragma unroll
    for (int e = 0; e < 8; ++e) ov[e] = (T)(o[e] * inv);
    *reinterpret_cast<v8 *>(out + bt * C + h * 8) = ov;
}

// ------------------------------------------------------------------------------------------------
// time embedding MLP (fp32): act_emb = SiLU(W2 SiLU(W1 [cos|sin](t f) + b1) + b2)
// ------------------------------------------------------------------------------------------------
// grid (D/64, B): each block recomputes the 128->D hidden layer of its sample (cheap) and produces 64
// outputs of the second layer with 4 k-slices per output (coalesced reads of the transposed weights).
template <typename T>
__global__ __launch_bounds__(256) void temb_mlp_kernel(const float *__restrict__ t, int C0, int D,
                                                       const float *__restrict__ W1t, const float *__restrict__ b1,
                                                       const float *__restrict__ W2t, const float *__restrict__ b2,
                                                       T *__restrict__ act) {
    __shared__ float emb[256];
    __shared__ float h1[1024];
    __shared__ float part[4][64];
    const int b = blockIdx.y, n0 = blockIdx.x * 64, tid = threadIdx.x;
    const float tv = t[b];
    const int half = C0 >> 1;
    for (int i = tid; i < C0; i += 256) {
        const int k = i < half ? i : i - half;
        const float f = expf(-9.210340371976184f * (float)k / (float)half);   // ln(10000)
        const float ang = tv * f;
        emb[i] = i < half ? cosf(ang) : sinf(ang);                            // flip_sin_to_cos
    }
    __syncthreads();
    for (int n = tid; n < D; n += 256) {
        float s = b1[n];
#pragma unroll 8
        for (int k = 0; k < C0; ++k) s = fmaf(emb[k], W1t[(size_t)k * D + n], s);
        h1[n] = s / (1.0f + expf(-s));
    }
    __syncthreads();
    const int nl = tid & 63, ksl = tid >> 6, kper = D >> 2;
    float s = 0.f;
#pragma unroll 8
    for (int k = ksl * kper; k < (ksl + 1) * kper; ++k) s = fmaf(h1[k], W2t[(size_t)k * D + n0 + nl], s);
    part[ksl][nl] = s;
    __syncthreads();
    if (tid < 64) {
        const float v = b2[n0 + tid] + ((part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]));
        act[(size_t)b * D + n0 + tid] = (T)(v / (1.0f + expf(-v)));
    }
}

inline int ilog2(int v) {
    int r = 0;
    while ((1 << r) < v) ++r;
    return r;
}

template <typename T, int WM, int WN, int TM, int TN, int EPI, int STAGES, bool FAST>
int launch_conv_cfg2(const ConvArgs &a, hipStream_t st) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int smem = STAGES * (BM + BN) * 128;
    int ksteps = 0;
    for (int i = 0; i < a.nseg; ++i) ksteps += a.seg[i].taps * (a.seg[i].C / 64);
    const int M = a.B * a.H * a.W;
    const int ntm = ceil_div(M, BM), ntn = ceil_div(a.Cout, BN);
    static bool attr = false;
    if (!attr) {
        BNDM_CHECK_HIP(hipFuncSetAttribute(
            reinterpret_cast<const void *>(&conv_igemm<T, WM, WN, TM, TN, EPI, STAGES, FAST>),
            hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr = true;
    }
    dim3 grid(ntm * ntn, a.splitk > 1 ? a.splitk : 1);
#ifdef BNDM_ABLATION      // profiling builds only (tools/ablate.sh)
#include "ablation_igemm_trace.inc"
#endif
    hipLaunchKernelGGL((conv_igemm<T, WM, WN, TM, TN, EPI, STAGES, FAST>), grid, dim3(WM * WN * 64), smem, st, a,
                       ksteps, ilog2(a.W), ilog2(a.H), ntm, ntn);
    return launch_status("conv_igemm");
}

template <typename T, int WM, int WN, int TM, int TN, int EPI, int STAGES>
int launch_conv_cfg(const ConvArgs &a, hipStream_t st) {
    // the table-driven path pays ~100 VALU of per-piece set-up: only worth it with enough K-steps per block
    int ksteps = 0;
    for (int i = 0; i < a.nseg; ++i) ksteps += a.seg[i].taps * (a.seg[i].C / 64);
    constexpr int fast_min = 4;
    bool fast = a.steps != nullptr && ksteps / (a.splitk > 1 ? a.splitk : 1) >= fast_min;
    for (int i = 0; i < a.nseg; ++i) {
        fast = fast && !a.seg[i].up;
        // the table-driven path addresses a source through a buffer descriptor with 32-bit offsets
        const long long bytes = (long long)a.B * a.H * a.stride * a.W * a.stride * a.seg[i].C * 2 +
                                (long long)(a.W * a.stride + 1) * a.seg[i].C * 2;
        fast = fast && bytes < (1LL << 31);
    }
    return fast ? launch_conv_cfg2<T, WM, WN, TM, TN, EPI, STAGES, true>(a, st)
                : launch_conv_cfg2<T, WM, WN, TM, TN, EPI, STAGES, false>(a, st);
}

template <typename T>
int launch_conv_t(int tile, int epi, const ConvArgs &a, hipStream_t st) {
    if (tile == TILE_256x128) {      // 8 waves, 3-stage ring (144 KB LDS, one block per CU)
        if (epi == EPI_NHWC16) return launch_conv_cfg<T, 4, 2, 2, 2, EPI_NHWC16, 3>(a, st);
        if (epi == EPI_F32_ROWS) return launch_conv_cfg<T, 4, 2, 2, 2, EPI_F32_ROWS, 3>(a, st);
    } else if (tile == TILE_128x128) {   // 4-stage ring (128 KB LDS), 8 waves (32x64 wave tiles)
        if (epi == EPI_NHWC16) return launch_conv_cfg<T, 4, 2, 1, 2, EPI_NHWC16, 4>(a, st);
        if (epi == EPI_F32_ROWS) return launch_conv_cfg<T, 4, 2, 1, 2, EPI_F32_ROWS, 4>(a, st);
    } else if (tile == TILE_128x32) {    // 4 waves, 4-stage ring (80 KB LDS)
        if (epi == EPI_NCHW32) return launch_conv_cfg<T, 4, 1, 1, 1, EPI_NCHW32, 4>(a, st);
        if (epi == EPI_NHWC16) return launch_conv_cfg<T, 4, 1, 1, 1, EPI_NHWC16, 4>(a, st);
    }
    set_error("launch_conv: unsupported tile/epilogue combination %d/%d", tile, epi);
    return BNDM_E_ARG;
}

inline int grid_for(size_t n) {
    size_t b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

}  // namespace

#define DISPATCH_T(dtype, expr_f16, expr_bf16) ((dtype) == BNDM_DTYPE_F16 ? (expr_f16) : (expr_bf16))

int launch_conv(int dtype, int tile, int epi, const ConvArgs &a, hipStream_t st) {
    if (a.wtiled && tile == TILE_128x32) {
        set_error("conv_igemm: tile-contiguous weights are packed for 128-row tiles");
        return BNDM_E_ARG;
    }
    for (int i = 0; i < a.nseg; ++i)
        if (a.seg[i].C % 64 != 0 || (a.seg[i].taps != 1 && a.seg[i].taps != 9)) {
            set_error("launch_conv: segment %d has C=%d taps=%d", i, a.seg[i].C, a.seg[i].taps);
            return BNDM_E_ARG;
        }
    if ((a.H & (a.H - 1)) || (a.W & (a.W - 1))) {
        set_error("launch_conv: H, W must be powers of two (%d x %d)", a.H, a.W);
        return BNDM_E_ARG;
    }
    return DISPATCH_T(dtype, launch_conv_t<_Float16>(tile, epi, a, st), launch_conv_t<__bf16>(tile, epi, a, st));
}

int launch_splitk_reduce(int dtype, const float *part, int splitk, const ConvArgs &a, hipStream_t st) {
    const size_t total = (size_t)a.B * a.H * a.W * (a.Cout / 4);
    const int lg = ilog2(a.H * a.W);
    if (dtype == BNDM_DTYPE_F16)
        hipLaunchKernelGGL(splitk_reduce_kernel<_Float16>, dim3(grid_for(total)), dim3(256), 0, st, part, splitk, a, lg);
    else
        hipLaunchKernelGGL(splitk_reduce_kernel<__bf16>, dim3(grid_for(total)), dim3(256), 0, st, part, splitk, a, lg);
    return launch_status("splitk_reduce");
}

int launch_conv_in(int dtype, const float *x, int Cx, const float *extra, int Ce, const void *W16, const float *bias,
                   void *out, float *stats, int B, int H, int W, int C0, int KP, hipStream_t st) {
    const int M = B * H * W;
    if ((H * W) % 128 || (C0 != 64 && C0 != 128 && C0 != 256 && C0 != 512)) {
        set_error("conv_in: H*W=%d must be a multiple of 128 and C0=%d one of 64/128/256/512", H * W, C0);
        return BNDM_E_ARG;
    }
    const int blocks = M / 128;
    const int rows = (W >= 128 ? 1 : 128 / W) + 2, cols = (W >= 128 ? 128 : W) + 2;
    const int patch_bytes = (Cx + Ce) * rows * cols * 2;
    const int tile_bytes = 128 * C0 * 2 + (256 / (C0 / 8)) * C0 * 2 * 4;
    const int smem = patch_bytes > tile_bytes ? patch_bytes : tile_bytes;
    if (smem > 64 * 1024) {
        static bool attr = false;
        if (!attr) {
            BNDM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_in_kernel<_Float16>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            BNDM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_in_kernel<__bf16>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr = true;
        }
    }
    if (dtype == BNDM_DTYPE_F16)
        hipLaunchKernelGGL(conv_in_kernel<_Float16>, dim3(blocks), dim3(256), smem, st, x, Cx, extra, Ce,
                           (const _Float16 *)W16, bias, (_Float16 *)out, stats, B, ilog2(H), ilog2(W), C0, KP);
    else
        hipLaunchKernelGGL(conv_in_kernel<__bf16>, dim3(blocks), dim3(256), smem, st, x, Cx, extra, Ce,
                           (const __bf16 *)W16, bias, (__bf16 *)out, stats, B, ilog2(H), ilog2(W), C0, KP);
    return launch_status("conv_in");
}

namespace {
// one wave per row; lane l owns the 16-byte chunks l, l+64, ... of the row (at most 8 of them)
template <typename T>
__global__ __launch_bounds__(256) void softmax_rows_kernel(T *__restrict__ s, int rows, int n, float scale) {
    using v8 = typename TT<T>::v8;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), l = threadIdx.x & 63;
    if (row >= rows) return;
    T *p = s + (size_t)row * n;
    const int nch = n >> 3;
    float v[8][8];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = l + 64 * i;
        if (c < nch) {
            const v8 x = *reinterpret_cast<const v8 *>(p + c * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                v[i][e] = (float)x[e] * scale;
                mx = fmaxf(mx, v[i][e]);
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
        if (l + 64 * i < nch) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                v[i][e] = __expf(v[i][e] - mx);
                sum += v[i][e];
            }
        }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = l + 64 * i;
        if (c < nch) {
            v8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (T)(v[i][e] * inv);
            *reinterpret_cast<v8 *>(p + c * 8) = o;
        }
    }
}

__global__ __launch_bounds__(256) void pointwise_f32_kernel(const float *__restrict__ z, const float *__restrict__ w,
                                                            const float *__restrict__ bias, float *__restrict__ out,
                                                            int Cin, int Cout, int HW, size_t total) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int p = (int)(i % HW);
        const size_t bm = i / HW;
        const int m = (int)(bm % Cout);
        const size_t b = bm / Cout;
        float acc = bias[m];
        for (int c = 0; c < Cin; ++c) acc += w[m * Cin + c] * z[(b * Cin + c) * HW + p];
        out[i] = acc;
    }
}
}  // namespace

int launch_softmax_rows(int dtype, void *s, int rows, int n, float scale, hipStream_t st) {
    if (n % 8 || n > 4096 || rows < 1) {
        set_error("softmax_rows: row length %d unsupported (multiple of 8, <= 4096)", n);
        return BNDM_E_ARG;
    }
    const dim3 grid((rows + 3) / 4);
    if (dtype == BNDM_DTYPE_F16) hipLaunchKernelGGL(softmax_rows_kernel<_Float16>, grid, dim3(256), 0, st, (_Float16 *)s, rows, n, scale);
    else hipLaunchKernelGGL(softmax_rows_kernel<__bf16>, grid, dim3(256), 0, st, (__bf16 *)s, rows, n, scale);
    return launch_status("softmax_rows");
}

int launch_pointwise_f32(const float *z, const float *w, const float *bias, float *out, int B, int Cin, int Cout, int HW,
                         hipStream_t st) {
    const size_t total = (size_t)B * Cout * HW;
    hipLaunchKernelGGL(pointwise_f32_kernel, dim3(grid_for(total)), dim3(256), 0, st, z, w, bias, out, Cin, Cout, HW, total);
    return launch_status("pointwise_f32");
}

// Host: per-K-step table of the FAST addressing path (4 ints per step)
std::vector<int> build_conv_steps(const ConvSeg *seg, int nseg, int W_out, int stride) {
    std::vector<int> out;
    const int Ws = W_out * stride;
    for (int i = 0; i < nseg; ++i)
        for (int t = 0; t < seg[i].taps; ++t)
            for (int c = 0; c < seg[i].C / 64; ++c) {
                const int tap = seg[i].taps == 9 ? t : 4;
                const int dy = tap / 3 - 1, dx = tap % 3 - 1;
                out.push_back(i | (tap << 8));
                out.push_back(dy * Ws + dx);
                out.push_back(c * 64);
                out.push_back(((dy + 1) * Ws + (dx + 1)) * seg[i].C * 2 + c * 128);     // scalar byte offset of the step
            }
    return out;
}

int conv_tile_bm(int tile) { return tile == TILE_256x128 ? 256 : 128; }

int gn_num_slabs(int HW) {
    int n = HW / 256;
    return n < 1 ? 1 : (n > 16 ? 16 : n);
}

int launch_gn_stats(int dtype, const void *x1, int C1, const void *x2, int C2, int B, int HW, float *partial,
                    int nslab, hipStream_t st) {
    if ((C1 + C2) / 8 > 256 || (C1 % 8) || (C2 % 8)) {
        set_error("gn_stats: unsupported channel split %d+%d", C1, C2);
        return BNDM_E_ARG;
    }
    dim3 grid(nslab, B);
    if (dtype == BNDM_DTYPE_F16)
        hipLaunchKernelGGL(gn_stats_kernel<_Float16>, grid, dim3(256), 0, st, (const _Float16 *)x1, C1,
                           (const _Float16 *)x2, C2, HW, partial, nslab);
    else
        hipLaunchKernelGGL(gn_stats_kernel<__bf16>, grid, dim3(256), 0, st, (const __bf16 *)x1, C1,
                           (const __bf16 *)x2, C2, HW, partial, nslab);
    return launch_status("gn_stats");
}

int launch_gn_finalize(const float *partial, int nslab, int B, int HW, int C, int groups, float eps,
                       const float *gamma, const float *beta, float *scale_shift, hipStream_t st) {
    if (C > 1024 || C % groups) {
        set_error("gn_finalize: C=%d groups=%d unsupported", C, groups);
        return BNDM_E_ARG;
    }
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(B), dim3(256), 0, st, partial, nslab, HW, C, groups, eps, gamma,
                       beta, scale_shift);
    return launch_status("gn_finalize");
}

int launch_gn_apply(int dtype, const void *x1, int C1, const void *x2, int C2, const float *scale_shift, int B,
                    int HW, int silu, void *out, hipStream_t st) {
    const size_t total = (size_t)B * HW * ((C1 + C2) / 8);
    if (dtype == BNDM_DTYPE_F16)
        hipLaunchKernelGGL(gn_apply_kernel<_Float16>, dim3(grid_for(total)), dim3(256), 0, st, (const _Float16 *)x1,
                           C1, (const _Float16 *)x2, C2, scale_shift, HW, total, silu, (_Float16 *)out);
    else
        hipLaunchKernelGGL(gn_apply_kernel<__bf16>, dim3(grid_for(total)), dim3(256), 0, st, (const __bf16 *)x1, C1,
                           (const __bf16 *)x2, C2, scale_shift, HW, total, silu, (__bf16 *)out);
    return launch_status("gn_apply");
}

int launch_attention(int dtype, const void *qkv, void *out, int B, int T, int C, hipStream_t st) {
    const size_t total = (size_t)B * (C / 8) * T;
    const int blocks = (int)((total + 255) / 256);
    if (dtype == BNDM_DTYPE_F16)
        hipLaunchKernelGGL(attention_kernel<_Float16>, dim3(blocks), dim3(256), 0, st, (const _Float16 *)qkv,
                           (_Float16 *)out, B, T, C);
    else
        hipLaunchKernelGGL(attention_kernel<__bf16>, dim3(blocks), dim3(256), 0, st, (const __bf16 *)qkv,
                           (__bf16 *)out, B, T, C);
    return launch_status("attention");
}

int launch_temb_mlp(int dtype, const float *t, int B, int C0, int D, const float *W1t, const float *b1,
                    const float *W2t, const float *b2, void *act_emb16, hipStream_t st) {
    if (C0 > 256 || D > 1024 || D % 64) {
        set_error("temb_mlp: C0=%d D=%d unsupported", C0, D);
        return BNDM_E_ARG;
    }
    const dim3 grid(D / 64, B);
    if (dtype == BNDM_DTYPE_F16)
        hipLaunchKernelGGL(temb_mlp_kernel<_Float16>, grid, dim3(256), 0, st, t, C0, D, W1t, b1, W2t, b2,
                           (_Float16 *)act_emb16);
    else
        hipLaunchKernelGGL(temb_mlp_kernel<__bf16>, grid, dim3(256), 0, st, t, C0, D, W1t, b1, W2t, b2,
                           (__bf16 *)act_emb16);
    return launch_status("temb_mlp");
}

}  // namespace bndm
