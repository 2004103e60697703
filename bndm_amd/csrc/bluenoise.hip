// Tiled Gaussian blue-noise generator for gfx950: out = unvec(L . vec(z)) per 64x64 tile, fused
// with the reference's tile gather / batch-major re-read / noise_padding / white<->blue lerp.
//
// Stands in for get_noise_v2 (bluenoise/get_noise_recent.py:23-196): torch.matmul(cov_mat_L, noise)
// (:88,:113,:146) + view/permute/contiguous (:111,:143-146) + torch.cat tile split and noise_padding
// (:131-133, :7-19) + lerp (:91,:116,:160).
//
// Two launches:
//   1. bluenoise_gemm<NT>: exact-f32 MFMA (v_mfma_f32_32x32x2_f32 == an fmaf chain) over the LOWER
//      TRIANGLE of L only.  A block owns 64 rows of L x (64*NT) z-columns x one 1024-wide k segment;
//      L and z tiles are staged with 16-byte global_load_lds into XOR-swizzled LDS images
//      (double-buffered), so every byte of L is fetched from HBM in full 256-B row pieces and
//      ds_read_b128 fragment reads are bank-conflict free.  Partial sums go to per-segment slabs
//      (deterministic: no atomics).
//   2. bluenoise_finish: sums the <=4 (<=32 behind bluenoise_small) slabs in fixed order and writes noise / noise_bn / noise_wn
//      in the reference's output layouts (32-px crop, 128-px slot placement and the scrambled
//      white-noise view), with the per-sample lerp evaluated in the reference's operation order.
//
// Algorithmic work: 4*T(4096) = 33,562,624 B of L read once, 2*T(4096)*n flop.
#include "common.hpp"

namespace bndm {

namespace {

constexpr int NPIX = 4096;
constexpr int BM = 64;       // rows of L per block
constexpr int BK = 64;       // k per LDS stage
constexpr int KSEG = 1024;   // k per work unit
constexpr int NSEG = NPIX / KSEG;

struct ZSrc {
    const float *z;
    int layout;
    int B_global;
    int C;
};

// &z(fcol, c, j).  For j % 4 == 0 the 4 floats j..j+3 are contiguous in every layout.
__device__ __forceinline__ const float *z_addr(const ZSrc &s, int fcol, int c, int j) {
    const int jy = j >> 6, jx = j & 63;
    if (s.layout == BNDM_Z_COLUMNS) return s.z + ((size_t)(fcol * s.C + c)) * NPIX + j;
    if (s.layout == BNDM_Z_IMAGE32)
        return s.z + ((size_t)(fcol * s.C + c)) * 1024 + (jy & 31) * 32 + (jx & 31);
    const int k = fcol / s.B_global, b = fcol - k * s.B_global;      // tile-major: f = k*B + b
    const int r0 = (k >> 1) * 64, c0 = (k & 1) * 64;                 // TL, TR, BL, BR
    return s.z + (((size_t)(b * s.C + c)) * 128 + r0 + jy) * 128 + c0 + jx;
}

// offset (in floats) of tile row jy relative to tile row 0, per layout
__device__ __forceinline__ int z_row_off(int layout, int jy) {
    if (layout == BNDM_Z_COLUMNS) return jy * 64;
    if (layout == BNDM_Z_IMAGE32) return (jy & 31) * 32;
    return jy * 128;
}

template <int NT>
__global__ __launch_bounds__(256) void bluenoise_gemm(const float *__restrict__ L, ZSrc zs,
                                                      float *__restrict__ part, int ncols,
                                                      int f_begin, int dense) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BN = 64 * NT;
    constexpr int L_BYTES = BM * BK * 4;
    constexpr int Z_BYTES = BN * BK * 4;
    constexpr int STAGE = L_BYTES + Z_BYTES;

    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l = tid & 63;

    // ---- work unit: (row panel p, k segment), heaviest panels first --------------------------
    const int u = blockIdx.x;
    int p, seg;
    if (dense || u < 64) { p = 63 - (u >> 2); seg = u & 3; }
    else if (u < 112)    { const int v = u - 64;  p = 47 - v / 3;    seg = v % 3; }
    else if (u < 144)    { const int v = u - 112; p = 31 - (v >> 1); seg = v & 1; }
    else                 { p = 15 - (u - 144); seg = 0; }
    const int i0 = p * BM;
    const int kbeg = seg * KSEG;
    const int kend = dense ? kbeg + KSEG : min(kbeg + KSEG, i0 + BM);
    const int nchunks = (kend - kbeg) / BK;
    const int col0 = blockIdx.y * BN;

    // ---- per-thread staging descriptors ------------------------------------------------------
    const int q = w * 64 + l;                 // 16-byte piece index inside one 4-KB block-instruction
    const int prow = q >> 4;                  // 0..15
    const int pchunk = q & 15;                // physical 16-B chunk inside the 256-B LDS row
    const int lchunk = pchunk ^ (prow & 15);  // logical chunk stored there (rows r*16+prow: same key)
    const float *lsrc[4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
        lsrc[r] = L + (size_t)(i0 + r * 16 + prow) * NPIX + kbeg + 4 * lchunk;
    const float *zsrc[4 * NT];
#pragma unroll
    for (int r = 0; r < 4 * NT; ++r) {
        int lc = col0 + r * 16 + prow;
        lc = lc < ncols ? lc : ncols - 1;
        const int fl = lc / zs.C;
        zsrc[r] = z_addr(zs, f_begin + fl, lc - fl * zs.C, 4 * lchunk);
    }
    const int jy0 = kbeg >> 6;

    auto stage = [&](int buf, int t) {
        char *base = smem + buf * STAGE;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            glds16(lsrc[r] + t * BK, base + r * 4096 + w * 1024);
        const int zoff = z_row_off(zs.layout, jy0 + t);
#pragma unroll
        for (int r = 0; r < 4 * NT; ++r)
            glds16(zsrc[r] + zoff, base + L_BYTES + r * 4096 + w * 1024);
    };

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    const int wi = w & 1, wj = w >> 1;
    const int frow = l & 31, kh = l >> 5;
    const int key = l & 15;
    const int lrow = wi * 32 + frow;

    if (nchunks > 0) {
        stage(0, 0);
        wait_vmem_all();
        __syncthreads();
    }
    int cur = 0;
    for (int t = 0; t < nchunks; ++t) {
        if (t + 1 < nchunks) stage(cur ^ 1, t + 1);
        const float *Lt = reinterpret_cast<const float *>(smem + cur * STAGE);
        const float *Zt = reinterpret_cast<const float *>(smem + cur * STAGE + L_BYTES);
#pragma unroll
        for (int s = 0; s < BK / 8; ++s) {
            const int pc = ((2 * s + kh) ^ key) * 4;
            const f32x4 lf = *reinterpret_cast<const f32x4 *>(Lt + lrow * BK + pc);
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) {
                const int zrow = wj * 32 * NT + tt * 32 + frow;
                const f32x4 zf = *reinterpret_cast<const f32x4 *>(Zt + zrow * BK + pc);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    acc[tt] = __builtin_amdgcn_mfma_f32_32x32x2f32(zf[e], lf[e], acc[tt], 0, 0, 0);
            }
        }
        wait_vmem_all();
        __syncthreads();
        cur ^= 1;
    }

    // ---- D rows = z columns, D cols = L rows: 128-B contiguous stores along i ----------------
    float *pseg = part + (size_t)seg * ncols * NPIX;
#pragma unroll
    for (int tt = 0; tt < NT; ++tt)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int lc = col0 + wj * 32 * NT + tt * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
            if (lc < ncols) pseg[(size_t)lc * NPIX + i0 + lrow] = acc[tt][e];
        }
}


// ------------------------------------------------------------------------------------------------
// HBM regime (at most 32 z-columns, i.e. B <= 10 RGB images of 64 px): the transform is bound by the ONE read of L's
// lower triangle (33.5 MB), and bluenoise_gemm's 160 work units of 64 rows x 1024 k leave 96 CUs idle and spend their
// time in MFMAs on padding columns.  bluenoise_small cuts the triangle into 528 units of 128 rows x 128 k (1024 for a
// dense L), four resident per CU: a workgroup of 4 waves stages 32-k slices of its 128 L rows (128-B row pieces,
// 16-B LDS-DMA, XOR-swizzled) and of the z columns through a double buffer; wave w multiplies rows 32w..32w+31 with
// the exact-f32 MFMA (32x32x2 for 17..32 columns, 16x16x4 for <= 16 columns: half the matrix-pipe time), skipping the
// slices of the diagonal unit that lie wholly above its rows.  Partial sums go to per-segment slabs (no atomics);
// bluenoise_finish sums the i/128 + 1 slabs of row i in segment order.
constexpr int SM_BM = 128, SM_BK = 32, SM_KSEG = 128;
constexpr int SM_NSEG = NPIX / SM_KSEG;          // 32

typedef float f32x4v __attribute__((ext_vector_type(4)));

constexpr int SM_NST = 2;                        // LDS stages (deeper rings measured slower: the per-CU miss queue, not the ring, bounds what is in flight)
constexpr int SM_STAGE = SM_BM * SM_BK * 4 + 32 * SM_BK * 4;     // 16 KiB of L (128 rows x 128 B) + 4 KiB of z (32 cols x 128 B)

template <bool W16>
__global__ __launch_bounds__(256, 4) void bluenoise_small(const float *__restrict__ L, ZSrc zs, float *__restrict__ part,
                                                          int ncols, int f_begin, int dense) {
    __shared__ __attribute__((aligned(16))) char smem[SM_NST * SM_STAGE];
    constexpr int L_BYTES = SM_BM * SM_BK * 4;
    constexpr int STAGE = SM_STAGE;
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l = tid & 63;

    // ---- work unit: (128-row panel P, 128-k segment S), S <= P unless dense.  A lower-triangular L has 32 diagonal
    // units (which read 10/16 of a unit: rows above a 32-k slice hold only zeros and are not fetched) and 496
    // off-diagonal ones: 528 workgroups, all resident at once (40 KiB of LDS each).  A workgroup's life is a chain of
    // memory latencies (two slices requested, ~1.8 us until the first lands, four slices, ~1 us until the slab stores
    // retire), so every workgroup gets exactly ONE unit: pairing the cheap diagonal units made those workgroups the
    // critical path of the launch.
    int P, S;
    {
        const int u = blockIdx.x;
        if (dense) {
            P = 31 - (u >> 5);
            S = u & 31;
        } else if (u < 32) {                       // diagonal units
            P = S = 31 - u;
        } else {                                   // off-diagonal units, longest panels first: panel P has P of them
            int rem = u - 32;
            P = 31;
            while (rem >= P) {
                rem -= P;
                --P;
            }
            S = rem;
        }
    }
    P = __builtin_amdgcn_readfirstlane(P);
    S = __builtin_amdgcn_readfirstlane(S);
    const bool diag = !dense && S == P;
    const int i0 = P * SM_BM, kbeg = S * SM_KSEG;
    constexpr int nslice = SM_KSEG / SM_BK;          // 4

    // ---- staging: piece q of a 4-KiB block instruction = row q >> 3, physical chunk q & 7 holds logical chunk
    // (q & 7) ^ ((row >> 1) & 7)
    const int prow = tid >> 3, pchunk = tid & 7;
    const float *lsrc[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = r * 32 + prow;
        lsrc[r] = L + (size_t)(i0 + row) * NPIX + kbeg + 4 * (pchunk ^ ((row >> 1) & 7));
    }
    int zc = prow < ncols ? prow : ncols - 1;                        // z column of this thread's piece (32 rows x 8 chunks)
    const int fl = zc / zs.C;
    const float *zsrc = z_addr(zs, f_begin + fl, zc - fl * zs.C, 0);
    const int zchunk = 4 * (pchunk ^ ((prow >> 1) & 7));
    auto stage = [&](int t) __attribute__((always_inline)) {        // t: 32-k slice of the segment
        char *base = smem + (t % SM_NST) * STAGE;
#pragma unroll
        for (int r = 0; r < 4; ++r)                                 // (diagonal unit: rows 32r .. 32r+31 are zero from slice r+1 on:
            glds16(diag && t > r ? lsrc[r] : lsrc[r] + t * SM_BK,   //  re-read slice 0 of the row -- an L2 hit -- instead)
                   base + r * 4096 + w * 1024);
        const int j = kbeg + t * SM_BK + zchunk;                     // 4 contiguous floats in every layout (j % 4 == 0)
        glds16(zsrc + z_row_off(zs.layout, j >> 6) + (zs.layout == BNDM_Z_IMAGE32 ? (j & 31) : (j & 63)),
               base + L_BYTES + w * 1024);
    };

    f32x16 acc32, acc32b;
    f32x4v acc16[2][2];              // [k half of the slice][row half]: the two k halves are added when the unit is done
#pragma unroll
    for (int e = 0; e < 16; ++e) acc32[e] = acc32b[e] = 0.f;
    acc16[0][0] = acc16[0][1] = acc16[1][0] = acc16[1][1] = f32x4v{0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int t = 0; t < SM_NST; ++t) stage(t);
    for (int t = 0; t < nslice; ++t) {
        if (t + 1 < nslice) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");     // slice t landed, slice t+1 may fly
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const float *Lt = reinterpret_cast<const float *>(smem + (t % SM_NST) * STAGE);
        const float *Zt = reinterpret_cast<const float *>(smem + (t % SM_NST) * STAGE + L_BYTES);
        // slices of the diagonal unit above this wave's rows hold only zeros of L: skipped (all waves still stage them)
        if (!diag || t <= w) {
            if constexpr (W16) {
                // 16x16x4: lane (i = l & 15, kq = l >> 4) holds 4 consecutive k of its row; MFMA e multiplies element e.
                // All six fragments are read first; the 16 MFMAs of a slice run as four independent accumulation chains
                // (k half x row half) instead of two chains of eight dependent instructions.
                const int i = l & 15, kq = l >> 4;
                f32x4v zf[2], lf[2][2];
#pragma unroll
                for (int b16 = 0; b16 < 2; ++b16) {
                    const int c = 4 * b16 + kq;
                    zf[b16] = *reinterpret_cast<const f32x4v *>(Zt + i * SM_BK + 4 * (c ^ ((i >> 1) & 7)));
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int row = w * 32 + h * 16 + i;
                        lf[b16][h] = *reinterpret_cast<const f32x4v *>(Lt + row * SM_BK + 4 * (c ^ ((row >> 1) & 7)));
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int b16 = 0; b16 < 2; ++b16)
#pragma unroll
                        for (int h = 0; h < 2; ++h)
                            acc16[b16][h] = __builtin_amdgcn_mfma_f32_16x16x4f32(zf[b16][e], lf[b16][h][e], acc16[b16][h], 0, 0, 0);
            } else {
                const int q = l & 31, kh = l >> 5;
                const int row = w * 32 + q;
#pragma unroll
                for (int sx = 0; sx < SM_BK / 8; ++sx) {
                    const int c = 2 * sx + kh;
                    const f32x4v lf = *reinterpret_cast<const f32x4v *>(Lt + row * SM_BK + 4 * (c ^ ((row >> 1) & 7)));
                    const f32x4v zf = *reinterpret_cast<const f32x4v *>(Zt + q * SM_BK + 4 * (c ^ ((q >> 1) & 7)));
#pragma unroll
                    for (int e = 0; e < 4; ++e) {                    // two accumulation chains (even / odd k steps)
                        if (e & 1) acc32b = __builtin_amdgcn_mfma_f32_32x32x2f32(zf[e], lf[e], acc32b, 0, 0, 0);
                        else acc32 = __builtin_amdgcn_mfma_f32_32x32x2f32(zf[e], lf[e], acc32, 0, 0, 0);
                    }
                }
            }
        }
        if (t + SM_NST < nslice) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                                        // buffer t % SM_NST may be overwritten
            asm volatile("" ::: "memory");
            stage(t + SM_NST);
        }
    }

    // ---- D rows = z columns, D cols = L rows: contiguous stores along i ------------------------------------
    float *pseg = part + (size_t)S * ncols * NPIX;
    if constexpr (W16) {
        const int i = l & 15, kq = l >> 4;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int col = 4 * kq + e;
                if (col < ncols) pseg[(size_t)col * NPIX + i0 + w * 32 + h * 16 + i] = acc16[0][h][e] + acc16[1][h][e];
            }
    } else {
        const int q = l & 31, kh = l >> 5;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int col = (e & 3) + 8 * (e >> 2) + 4 * kh;
            if (col < ncols) pseg[(size_t)col * NPIX + i0 + w * 32 + q] = acc32[e] + acc32b[e];
        }
    }
}

#pragma clang fp contract(off)   // lerp evaluated as torch does: two rounded products, one add
__global__ __launch_bounds__(256) void bluenoise_finish(const float *__restrict__ part, ZSrc zs,
                                                        const float *__restrict__ alpha,
                                                        float *__restrict__ noise,
                                                        float *__restrict__ noise_bn,
                                                        float *__restrict__ noise_wn, int ncols,
                                                        int b_begin, int b_count, int res, int mode,
                                                        int dense, int kseg) {
    const int C = zs.C;
    const size_t total = (size_t)b_count * C * res * res;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int X = (int)(idx % res);
        const int Y = (int)((idx / res) % res);
        const int c = (int)((idx / ((size_t)res * res)) % C);
        const int bl = (int)(idx / ((size_t)res * res * C));
        const int b = b_begin + bl;
        int i, fcol, lc;
        if (res == 128) {
            const int slot = (Y >> 6) + 2 * (X >> 6);          // noise_padding placement (:10-14)
            i = (Y & 63) * 64 + (X & 63);
            fcol = 4 * b + slot;                               // batch-major re-read (:144,:146)
            lc = (fcol - 4 * b_begin) * C + c;
        } else {
            i = Y * 64 + X;                                    // 32 px: crop of the 64 grid (:97-99)
            fcol = b;
            lc = bl * C + c;
        }
        float wn;
        if (res == 128) {                                      // [4096, C] buffer re-read as [C, 4096]
            const int qq = c * NPIX + i;
            const int j = qq / C;
            wn = *z_addr(zs, fcol, qq - j * C, j);
        } else {
            wn = *z_addr(zs, fcol, c, i);
        }
        float out;
        if (mode == BNDM_NOISE_SCRAMBLE) {
            out = wn;
        } else {
            const int nseg = dense ? NPIX / kseg : i / kseg + 1;
            float bn = part[(size_t)lc * NPIX + i];
            for (int s0 = 1; s0 < nseg; s0 += 31) {             // all slabs in flight at once; the sum keeps the segment order
                float v[31];
#pragma unroll
                for (int k = 0; k < 31; ++k) v[k] = s0 + k < nseg ? part[((size_t)(s0 + k) * ncols + lc) * NPIX + i] : 0.f;
#pragma unroll
                for (int k = 0; k < 31; ++k)
                    if (s0 + k < nseg) bn = bn + v[k];
            }
            if (noise_bn) noise_bn[idx] = bn;
            if (mode == BNDM_NOISE_BLEND) {
                const float a = alpha[b];
                out = bn * (1.0f - a) + wn * a;
            } else {
                out = bn;
            }
        }
        if (noise_wn) noise_wn[idx] = wn;
        if (noise) noise[idx] = out;
    }
}

template <int NT>
int launch_gemm(const float *L, const ZSrc &zs, float *part, int ncols, int f_begin, int dense,
                hipStream_t st) {
    constexpr int BN = 64 * NT;
    constexpr int smem = 2 * (BM * BK * 4 + BN * BK * 4);
    static bool attr_set = false;
    if (!attr_set) {
        BNDM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&bluenoise_gemm<NT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr_set = true;
    }
    dim3 grid(dense ? 256 : 160, ceil_div(ncols, BN));
    hipLaunchKernelGGL(bluenoise_gemm<NT>, grid, dim3(256), smem, st, L, zs, part, ncols, f_begin, dense);
    return launch_status("bluenoise_gemm");
}

}  // namespace
}  // namespace bndm

using namespace bndm;

extern "C" size_t bndm_bluenoise_workspace_bytes(int b_count, int C, int res) {
    if (b_count <= 0 || C <= 0) return 0;
    const size_t ncols = (size_t)b_count * C * (res == 128 ? 4 : 1);
    return (size_t)(ncols <= 32 ? SM_NSEG : NSEG) * ncols * NPIX * sizeof(float);     // one slab per k segment
}

extern "C" int bndm_bluenoise(const float *L, int l_dense, const float *z, int z_layout,
                              const float *alpha, float *noise, float *noise_bn, float *noise_wn,
                              int B_global, int b_begin, int b_count, int C, int res, int mode,
                              void *workspace, size_t workspace_bytes, void *stream) {
    BNDM_REQUIRE(res == 32 || res == 64 || res == 128,
                 "bndm_bluenoise: unsupported width %d (reference raises NotImplementedError, "
                 "get_noise_recent.py:166-167)", res);
    BNDM_REQUIRE(z != nullptr, "bndm_bluenoise: z is NULL");
    BNDM_REQUIRE(B_global > 0 && C > 0 && b_begin >= 0 && b_count >= 0 && b_begin + b_count <= B_global,
                 "bndm_bluenoise: bad batch range [%d,+%d) of %d", b_begin, b_count, B_global);
    BNDM_REQUIRE(mode == BNDM_NOISE_BLEND || mode == BNDM_NOISE_PURE_BN || mode == BNDM_NOISE_SCRAMBLE,
                 "bndm_bluenoise: bad mode %d", mode);
    BNDM_REQUIRE(z_layout == BNDM_Z_COLUMNS || (z_layout == BNDM_Z_IMAGE32 && res == 32) ||
                     (z_layout == BNDM_Z_IMAGE128 && res == 128),
                 "bndm_bluenoise: z_layout %d does not fit width %d", z_layout, res);
    BNDM_REQUIRE(mode != BNDM_NOISE_SCRAMBLE || res == 128, "bndm_bluenoise: scramble is 128-px only");
    BNDM_REQUIRE(mode != BNDM_NOISE_BLEND || alpha != nullptr, "bndm_bluenoise: alpha is NULL");
    if (b_count == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const int ncols = b_count * C * (res == 128 ? 4 : 1);
    const int f_begin = res == 128 ? 4 * b_begin : b_begin;
    ZSrc zs{z, z_layout, B_global, C};
    float *part = (float *)workspace;
    if (mode != BNDM_NOISE_SCRAMBLE) {
        BNDM_REQUIRE(L != nullptr, "bndm_bluenoise: L is NULL");
        BNDM_REQUIRE(workspace != nullptr &&
                         workspace_bytes >= bndm_bluenoise_workspace_bytes(b_count, C, res),
                     "bndm_bluenoise: workspace too small (%zu < %zu)", workspace_bytes,
                     bndm_bluenoise_workspace_bytes(b_count, C, res));
        int rc;
        if (ncols <= 32) {
            const dim3 grid(l_dense ? 32 * 32 : 528);
            if (ncols <= 16)
                hipLaunchKernelGGL(bluenoise_small<true>, grid, dim3(256), 0, st, L, zs, part, ncols, f_begin, l_dense ? 1 : 0);
            else
                hipLaunchKernelGGL(bluenoise_small<false>, grid, dim3(256), 0, st, L, zs, part, ncols, f_begin, l_dense ? 1 : 0);
            rc = launch_status("bluenoise_small");
        } else if (ncols <= 64) rc = launch_gemm<1>(L, zs, part, ncols, f_begin, l_dense ? 1 : 0, st);
        else if (ncols <= 128) rc = launch_gemm<2>(L, zs, part, ncols, f_begin, l_dense ? 1 : 0, st);
        else rc = launch_gemm<3>(L, zs, part, ncols, f_begin, l_dense ? 1 : 0, st);
        if (rc) return rc;
    }
    const size_t total = (size_t)b_count * C * res * res;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(bluenoise_finish, dim3(blocks), dim3(256), 0, st, part, zs, alpha, noise, noise_bn,
                       noise_wn, ncols, b_begin, b_count, res, mode, l_dense ? 1 : 0,
                       mode != BNDM_NOISE_SCRAMBLE && ncols <= 32 ? SM_KSEG : KSEG);
    return launch_status("bluenoise_finish");
}
