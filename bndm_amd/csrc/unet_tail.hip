// conv_s: the convolutions of the <= 8x8 levels (down_blocks.3-5, mid_block, up_blocks.0-2 of the UNet2DModel built at
// iadb_bn.py:205-282), one launch per convolution -- no split-K slabs, no separate GroupNorm launch.
//
// Why a second conv kernel.  At 8x8 / 4x4 / 2x2 a layer is 1-15 GFLOP against 2-9 MB of weights: a launch is its
// latency chain (first loads, reduction, stores), not its MFMA time.  The implicit-GEMM kernel needed split-K (fp32
// slabs + a consumer that sums them) to fill the chip and a gn_small launch per GroupNorm: ~100 dependent launches.
// conv_s removes both:
//   * a workgroup (8 waves, one per CU) owns TM = 64 / 128 rows = WHOLE samples (all H*W pixels of 1-16 samples) x 32
//     (96 for q|k|v) output channels and the FULL K range, so the GroupNorm(32) statistics of its outputs are local:
//     the epilogue writes the raw 16-bit tensor AND, for every consuming GroupNorm (group size 8 / 16 / 32 channels),
//     the normalised (+SiLU) tensor.  Consumers read already-normalised activations: their input side is pure DMA.
//   * K is split across the 8 WAVES, not across workgroups: each wave multiplies its own (32-channel sub-chunk, tap)
//     steps of a round against the shared activation patch and streams its own weight fragments global -> VGPR through
//     a D-deep register ring (no LDS for weights, no barrier inside a round); the 8 partial tiles are added through LDS
//     in a fixed order.
//   * a round = up to 256 channels of one source tensor for all TM rows ([TM][256] 16-bit, 16-byte slots XOR-swizzled
//     with the row index so that fragment reads at any tap shift are bank-conflict free), brought in by LDS-DMA one
//     round ahead (two buffers).  3x3 taps read the same patch through a per-lane address table (row + tap shift, or a
//     zero row for padding); nearest-2x upsampled and stride-2 (four phase rounds) sources only change the DMA row map.
//   * weights are packed on the host in exactly the order each wave consumes them (pack in build_tail_plan), so the
//     kernel's weight addressing is "pointer += 1 KiB".
// The attention blocks use the same kernel: q|k|v projection of four heads per workgroup + softmax(q k^T / sqrt(8)) v
// in the epilogue (fp32), then the output projection + residual as a plain 1x1 conv_s.
#include "unet_kernels.hpp"
#include "unet_types.hpp"
#include <cstring>

namespace bndm {
namespace {

constexpr int TS_NW = 8;                 // waves per workgroup
constexpr int TS_NT = TS_NW * 64;
constexpr int TS_ROWB = 512;             // bytes per patch row (256 channels)
constexpr int TS_ZSLOT = 9;              // table slot whose entries all point at the zero row

template <int TM> constexpr int ts_patch_bytes() { return (TM + 2) * TS_ROWB; }      // + zero row, 1 KiB multiple
template <int TM> constexpr int ts_tab_bytes() { return 10 * (TM / 32) * 64 * 4; }
template <int NB> constexpr int ts_slab_row() { return NB * 32 + 4; }                  // floats per staged row (padded)
// partial tiles staged for the cross-wave sum: all 8 when they fit, else 4 (waves 4..7 first, waves 0..3 add theirs)
template <int TM, int NB> constexpr int ts_nslab() { return 8 * TM * ts_slab_row<NB>() * 4 <= 148 * 1024 ? 8 : 4; }
template <int TM, int NB> constexpr int ts_epi_bytes() {
    return ts_nslab<TM, NB>() * TM * ts_slab_row<NB>() * 4 + TM * NB * 4 * 8 + 3 * 64 * 8;   // slabs, item sums, statistics
}
template <int TM, int NB> constexpr int ts_rows_off() {
    constexpr int main_b = 2 * ts_patch_bytes<TM>() + ts_tab_bytes<TM>();
    return main_b > ts_epi_bytes<TM, NB>() ? main_b : ts_epi_bytes<TM, NB>();
}
// additive rows [16 samples][TN] + gamma / beta of up to three consumers [3][2][TN], alive from prologue to epilogue
template <int TM, int NB> constexpr int ts_smem_bytes() { return ts_rows_off<TM, NB>() + 22 * NB * 32 * 4; }

template <int N> struct IC {
    static constexpr int value = N;
};

typedef const __attribute__((address_space(4))) uint32_t *const_u32_ptr;
typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// vmcnt(k * NL), k = 1..D (k is wave-uniform)
template <int NL, int D> __device__ __forceinline__ void wait_vm_steps(int k) {
    static_assert(NL * D <= 63, "vmcnt range");
    if (k > D) return;      // more than D entries since the DMA: the weight waits have retired it already
    switch (k) {
        case 1: wait_vm<NL * 1>(); break;
        case 2: wait_vm<(D >= 2 ? NL * 2 : 0)>(); break;
        case 3: wait_vm<(D >= 3 ? NL * 3 : 0)>(); break;
        case 4: wait_vm<(D >= 4 ? NL * 4 : 0)>(); break;
        case 5: wait_vm<(D >= 5 ? NL * 5 : 0)>(); break;
        case 6: wait_vm<(D >= 6 ? NL * 6 : 0)>(); break;
        case 7: wait_vm<(D >= 7 ? NL * 7 : 0)>(); break;
        case 8: wait_vm<(D >= 8 ? NL * 8 : 0)>(); break;
        default: break;
    }
}

template <typename T, int TM, int NB, int D>
__global__ __launch_bounds__(TS_NT) void conv_s(const TailArgs a) {
    using v8 = typename TT<T>::v8;
    constexpr int NMB = TM / 32;                 // 32-row MFMA blocks along M
    constexpr int NL = 2 * NB;                   // weight fragments (1 KiB loads) per step
    constexpr int TN = NB * 32;
    constexpr int PB = ts_patch_bytes<TM>();
    constexpr int OFF_TAB = 2 * PB;
    constexpr int OFF_ROWS = ts_rows_off<TM, NB>();
    constexpr int NDMA = TM / 16;                // patch DMA instructions per wave and round (1 KiB each)
    static_assert(D == 8 || D == 4 || D == 2, "descriptor fetch width");
    static_assert(D * NL + NDMA <= 63, "vmcnt range");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l = tid & 63;
    const int q = l & 31, kh = l >> 5;
    // profiling aid (tools/ubench/tail_bench): s_memtime marks of waves 0 and 7 of workgroup 0
    auto mark = [&](int k) {
        if (a.dbg && blockIdx.x == 0 && blockIdx.y == 0 && (w == 0 || w == 7)) {
            const unsigned long long tm = __builtin_amdgcn_s_memtime();
            if (l == 0) a.dbg[(w ? 16 : 0) + k] = tm;
        }
    };
    mark(0);

    // ---- tile id: grid (n-tiles, m-tiles).  Workgroups are dealt to the 8 XCDs in linear order, so with a multiple of 8
    // n-tiles every workgroup of an n-tile (same weights) lands on the same XCD / L2
    const int nt = blockIdx.x, mt = blockIdx.y;
    const int hwlog = a.hwlog, wlog = a.wlog, HW = 1 << hwlog, Wd = 1 << wlog, Hd = HW >> wlog;
    const int M = a.B << hwlog;
    const int m0 = mt * TM;
    const int Cout = a.Cout, n0 = nt * TN;
    const bool attn = a.epi == TAIL_EPI_ATTN;

    // ---- patch DMA of one round --------------------------------------------------------------------------------------
    // DMA instruction i of wave w fills LDS rows 2 (w NDMA + i), +1 (lane >> 5); physical slot p = lane & 31 of row R
    // holds logical 16-byte slot p ^ (R & 15)
    const TailRound *__restrict__ rtab = a.rounds;
    auto round_dma_desc = [&](uint64_t src, int row_bytes, int cbyte, int mode, int phase, int nsub, int buf) {
        const int rows_src = mode == 0 ? M : (mode == 1 ? M >> 2 : M << 2);
        const __amdgpu_buffer_rsrc_t rs = uniform_rsrc((const void *)src, rows_src * row_bytes);
        const int py = phase >> 1, px = phase & 1;
        if (mode == 0) {                                 // same resolution: the tile's rows are consecutive source rows
            const int R0 = w * NDMA * 2 + kh;
            int rowoff = (m0 + R0) * row_bytes + cbyte;
#pragma unroll
            for (int i = 0; i < NDMA; ++i) {
                const int R = R0 + 2 * i, s = q ^ (R & 15);
                const bool ok = m0 + R < M && s < nsub * 4;
                const unsigned voff = ok ? (unsigned)(rowoff + s * 16) : 0x80000000u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(smem + buf * PB + (w * NDMA + i) * 1024), 16, voff,
                                                         0, 0, 0);
                rowoff += 2 * row_bytes;
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < NDMA; ++i) {
            const int R = (w * NDMA + i) * 2 + kh;
            const int s = q ^ (R & 15);
            const int m = m0 + R;
            const int b = m >> hwlog, pix = m & (HW - 1), y = pix >> wlog, x = pix & (Wd - 1);
            const int srow = mode == 1 ? ((b << (hwlog - 2)) + ((y >> 1) << (wlog - 1))) + (x >> 1)
                                       : ((((b * Hd + y) * 2 + py) << (wlog + 1)) + 2 * x + px);
            const bool ok = m < M && s < nsub * 4;
            const unsigned voff = ok ? (unsigned)(srow * row_bytes + cbyte + s * 16) : 0x80000000u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(smem + buf * PB + (w * NDMA + i) * 1024), 16, voff, 0,
                                                     0, 0);
        }
    };
    // rounds 0 and 1 are described in the kernel arguments (no dependent scalar load in front of the first DMA), later
    // ones in the device table
    auto round_dma = [&](int r, int buf) {
        const const_u32_ptr rp = (const_u32_ptr)(rtab + r);
        const u32x8 e = *reinterpret_cast<const __attribute__((address_space(4))) u32x8 *>(rp);
        round_dma_desc(((uint64_t)e[1] << 32) | e[0], (int)e[2], (int)e[3], (int)e[4], (int)e[5], (int)e[6], buf);
    };

    const int nrounds = a.nrounds;
    round_dma_desc((uint64_t)a.r0.src, a.r0.row_bytes, a.r0.cbyte, a.r0.mode, a.r0.phase, a.r0.nsub, 0);
    mark(8);

    // ---- epilogue rows, requested right behind the first patch round: bias + time-embedding row per (sample of the tile, channel),
    // gamma / beta of the consuming GroupNorms; stored to LDS further down
    // channel of column c: plain tiles n0 + c; q|k|v tiles (c / 32) * C + 32 nt + c % 32
    float *rows = reinterpret_cast<float *>(smem + OFF_ROWS);         // [16][TN] additive, then [3][2][TN] gamma / beta
    constexpr int NADD = (16 * TN + TS_NT - 1) / TS_NT;
    float addv[NADD], addt[NADD], gbv = 0.f;      // (summed when they are stored: no wait in front of the DMAs)
    const int nsamp = TM >> hwlog;
#pragma unroll
    for (int k = 0; k < NADD; ++k) {
        const int e = tid + k * TS_NT, sidx = e / TN, c = e - sidx * TN;
        const int cb = attn ? (c >> 5) * Cout + nt * 32 + (c & 31) : n0 + c;
        const int b = (m0 >> hwlog) + sidx;
        addv[k] = addt[k] = 0.f;
        if (e < 16 * TN && sidx < nsamp && b < a.B) {
            if (a.bias) addv[k] = a.bias[cb];
            if (a.temb) addt[k] = a.temb[(size_t)b * a.temb_bstride + a.temb_off + cb];
        }
    }
    if (tid < 6 * TN) {
        const int r = tid / (2 * TN), rem = tid - r * 2 * TN, isb = rem / TN, c = rem - isb * TN;
        // (selected by value: indexing the kernel argument per lane would fetch the pointers with vector loads)
        const float *g0 = a.req[0].gamma, *g1 = a.req[1].gamma, *g2 = a.req[2].gamma;
        const float *b0 = a.req[0].beta, *b1 = a.req[1].beta, *b2 = a.req[2].beta;
        const float *gp = r == 0 ? (isb ? b0 : g0) : (r == 1 ? (isb ? b1 : g1) : (isb ? b2 : g2));
        if (r < a.nreq) gbv = gp[n0 + c];
    }

    mark(9);

    // ---- weight stream of this wave: fragments in consumption order, D steps ahead in registers ---------------------
    // Read through a buffer descriptor that ends behind the wave's last useful entry: the loads of the padding entries
    // (the list is padded to a multiple of D, and the ring runs D entries ahead) fall outside and cost no traffic.
    const int nuse = w == 0 ? a.nuse[0] : w == 1 ? a.nuse[1] : w == 2 ? a.nuse[2] : w == 3 ? a.nuse[3] : w == 4 ? a.nuse[4]
                   : w == 5 ? a.nuse[5] : w == 6 ? a.nuse[6] : a.nuse[7];
    const __amdgpu_buffer_rsrc_t wrs =
        uniform_rsrc((const char *)a.wgt + (size_t)nt * a.tile_bytes + (size_t)w * a.wave_bytes, nuse * NL * 1024);
    int wofs = 0;                                // byte offset of the next entry to request
    auto wload = [&](int frag) {
        return __builtin_bit_cast(v8, __builtin_amdgcn_raw_buffer_load_b128(wrs, l * 16 + frag * 1024, wofs, 0));
    };
    v8 Wr[D][NL];
    // issued in ring order (the compiler's vmcnt bookkeeping at the loop head takes the minimum over the prologue and the
    // back edge: a reordered prologue would make every iteration drain the ring)
    // Only the first half of the ring is requested in front of the first barrier: the first patch round shares the CU's
    // 64 B/clk load path with whatever is requested next to it, and nothing can start before it has landed in every wave.
    constexpr int DH = D / 2;
#pragma unroll
    for (int d = 0; d < DH; ++d)
#pragma unroll
        for (int f = 0; f < NL; ++f) {
            Wr[d][f] = wload(d * NL + f);
            asm volatile("" ::: "memory");
        }
    mark(10);

    // ---- fragment address table: [slot 0..9][m-block][lane] -> byte offset of the lane's 16 bytes inside a patch
    // buffer for (sub-chunk 0, k16 slice 0); slot = 3 (dy + 1) + (dx + 1), slot 9 = zero row
    {
        int *tabw = reinterpret_cast<int *>(smem + OFF_TAB);
        for (int e = tid; e < 10 * NMB * 64; e += TS_NT) {
            const int ln = e & 63, i = (e >> 6) % NMB, ts = e / (64 * NMB);
            const int R = 32 * i + (ln & 31), pix = R & (HW - 1), y = pix >> wlog, x = pix & (Wd - 1);
            const int dy = ts / 3 - 1, dx = ts % 3 - 1;
            const bool ok = ts < 9 && (unsigned)(y + dy) < (unsigned)Hd && (unsigned)(x + dx) < (unsigned)Wd;
            const int Rs = ok ? R + dy * Wd + dx : TM;
            tabw[e] = Rs * TS_ROWB + ((((Rs & 15) ^ (ln >> 5)) & 15) << 4);
        }
        // zero rows of both buffers (rows TM, TM + 1: 1 KiB)
        if (tid < 128) {
            const u32x4 z = {0u, 0u, 0u, 0u};
            *reinterpret_cast<u32x4 *>(smem + (tid >> 6) * PB + TM * TS_ROWB + (tid & 63) * 16) = z;
        }
#pragma unroll
        for (int k = 0; k < NADD; ++k)
            if (tid + k * TS_NT < 16 * TN) rows[tid + k * TS_NT] = addv[k] + addt[k];
        if (tid < 6 * TN) rows[16 * TN + tid] = gbv;
    }

    f32x16 acc[NB][NMB];
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
        for (int i = 0; i < NMB; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[n][i][e] = 0.f;

    mark(1);
    // round 0 of the patch is older than the DH * NL weight loads
    wait_vm<DH * NL>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // second half of the ring, then the second round's patch (it is needed at the first round boundary)
#pragma unroll
    for (int d = DH; d < D; ++d)
#pragma unroll
        for (int f = 0; f < NL; ++f) {
            Wr[d][f] = wload(d * NL + f);
            asm volatile("" ::: "memory");
        }
    wofs = D * NL * 1024;
    if (nrounds > 1) round_dma_desc((uint64_t)a.r1.src, a.r1.row_bytes, a.r1.cbyte, a.r1.mode, a.r1.phase, a.r1.nsub, 1);
    mark(11);

    mark(2);
    // ---- main loop: this wave's step list ------------------------------------------------------------------------------
    // entry: [3:0] table slot, [6:4] 32-channel sub-chunk of the round, [7] patch buffer, [8] round boundary after this
    // entry, [12:9] entries of this wave in the round (capped at D + 1), [13] no work (padding of the list,
    // or the placeholder of a wave without a step in a round: it only carries the boundary).  The fragments of entry k + 1 are read while the
    // MFMAs of entry k run (two register sets), except across a round boundary.
    const const_u32_ptr dp = (const_u32_ptr)a.desc + (size_t)w * a.maxsteps;
    const int *tab = reinterpret_cast<const int *>(smem + OFF_TAB) + l;
    int cur_round = 0;
    typedef uint32_t u32xD __attribute__((ext_vector_type(D)));
    u32xD dnext = *reinterpret_cast<const __attribute__((address_space(4))) u32xD *>(dp);
    v8 fb[2][2][NMB];
    auto read_frags = [&](uint32_t e, v8 (&f)[2][NMB]) __attribute__((always_inline)) {
        const int ts = e & 15, jx = ((e >> 4) & 7) << 6, bufoff = ((e >> 7) & 1) * PB;
        int ta[NMB];
#pragma unroll
        for (int i = 0; i < NMB; ++i) ta[i] = tab[(ts * NMB + i) * 64];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < NMB; ++i) f[ks][i] = *reinterpret_cast<const v8 *>(smem + ((ta[i] ^ (jx | (ks << 5))) + bufoff));
    };
    if (!(dnext[0] & 0x2000u)) read_frags(__builtin_amdgcn_readfirstlane(dnext[0]), fb[0]);
    for (int s0 = 0; s0 < a.maxsteps; s0 += D) {
        const u32xD dd = dnext;
        {
            const int sn = s0 + D < a.maxsteps ? s0 + D : s0;
            dnext = *reinterpret_cast<const __attribute__((address_space(4))) u32xD *>(dp + sn);
        }
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const uint32_t e = __builtin_amdgcn_readfirstlane(dd[d]);
            const uint32_t en = __builtin_amdgcn_readfirstlane(d + 1 < D ? dd[d + 1 < D ? d + 1 : 0] : dnext[0]);
            const bool boundary = (e & 0x100u) != 0;
            if (!boundary && !(en & 0x2000u)) read_frags(en, fb[(d + 1) & 1]);
            if (!(e & 0x2000u)) {                        // (padding and placeholder entries: no LDS reads, no MFMAs)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int n = 0; n < NB; ++n)
#pragma unroll
                        for (int i = 0; i < NMB; ++i)
                            acc[n][i] = TT<T>::mfma(Wr[d][ks * NB + n], fb[d & 1][ks][i], acc[n][i]);
            }
#pragma unroll
            for (int f = 0; f < NL; ++f) {
                Wr[d][f] = wload(f);
                asm volatile("" ::: "memory");
            }
            wofs += NL * 1024;
            if (boundary) {
                // round boundary: my pieces of the next round's patch have landed (they are older than the weights of
                // the entries of this round), every wave is done with the current buffer -> refill it two rounds ahead
                wait_vm_steps<NL, D>((int)((e >> 9) & 15));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                ++cur_round;
                if (cur_round + 1 < nrounds) round_dma(cur_round + 1, (cur_round + 1) & 1);
                if (!(en & 0x2000u)) read_frags(en, fb[(d + 1) & 1]);
            }
        }
    }
    mark(3);

    // ---- cross-wave sum through LDS -----------------------------------------------------------------------------------------
    constexpr int RS = ts_slab_row<NB>();        // floats per row
    constexpr int SLAB = TM * RS;                // floats per slab
    constexpr int NSLAB = ts_nslab<TM, NB>();
    constexpr int CH8 = TN / 8, ITEMS = TM * CH8;
    float *slab = reinterpret_cast<float *>(smem);
    float2 *part = reinterpret_cast<float2 *>(slab + NSLAB * SLAB);          // [ITEMS] (sum, sum of squares) of 8 channels
    float *stat = reinterpret_cast<float *>(part + ITEMS);                    // [3][64][2] mean, rstd
    // the residual of this thread's first item is requested before the barriers
    v8 rres;
    const bool has_res = a.resid && tid < ITEMS && m0 + tid / CH8 < M;
    if (has_res) rres = *reinterpret_cast<const v8 *>((const T *)a.resid + (size_t)(m0 + tid / CH8) * Cout + n0 + (tid % CH8) * 8);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                // every wave is done with the patch buffers
    asm volatile("" ::: "memory");
    mark(4);
    // accumulator element (n, i, g, e) of lane (q, kh): row 32 i + q, channel 32 n + 8 g + 4 kh + e
    auto acc_addr = [&](int sl, int n, int i, int g) { return slab + sl * SLAB + (32 * i + q) * RS + 32 * n + 8 * g + 4 * kh; };
    if (NSLAB == 8 || w >= 4) {
#pragma unroll
        for (int n = 0; n < NB; ++n)
#pragma unroll
            for (int i = 0; i < NMB; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v = {acc[n][i][4 * g], acc[n][i][4 * g + 1], acc[n][i][4 * g + 2], acc[n][i][4 * g + 3]};
                    *reinterpret_cast<f32x4 *>(acc_addr(NSLAB == 8 ? w : w - 4, n, i, g)) = v;
                }
    }
    __syncthreads();
    if constexpr (NSLAB == 4) {
        if (w < 4) {
#pragma unroll
            for (int n = 0; n < NB; ++n)
#pragma unroll
                for (int i = 0; i < NMB; ++i)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        float *p = acc_addr(w, n, i, g);
                        f32x4 v = *reinterpret_cast<const f32x4 *>(p);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += acc[n][i][4 * g + e];
                        *reinterpret_cast<f32x4 *>(p) = v;
                    }
        }
        __syncthreads();
    }
    mark(5);

    // ---- outputs: item = (row, 8 channels) ------------------------------------------------------------------------------
    if constexpr (NB == 3) {
        // q|k|v tile: + bias into the staged tile, then softmax(q k^T / sqrt(8)) v per (row, head); q | k | v of the tile's
        // four heads sit at columns 0 / 32 / 64
        for (int it = tid; it < ITEMS; it += TS_NT) {
            const int R = it / CH8, c8 = (it % CH8) * 8;
            float *p = slab + R * RS + c8;
            f32x4 v0 = *reinterpret_cast<const f32x4 *>(p), v1 = *reinterpret_cast<const f32x4 *>(p + 4);
#pragma unroll
            for (int k = 1; k < NSLAB; ++k) {
                v0 += *reinterpret_cast<const f32x4 *>(p + k * SLAB);
                v1 += *reinterpret_cast<const f32x4 *>(p + k * SLAB + 4);
            }
            const float *ar = rows + (R >> hwlog) * TN + c8;
            v0 += *reinterpret_cast<const f32x4 *>(ar);
            v1 += *reinterpret_cast<const f32x4 *>(ar + 4);
            *reinterpret_cast<f32x4 *>(p) = v0;
            *reinterpret_cast<f32x4 *>(p + 4) = v1;
        }
        mark(6);
        __syncthreads();
        const int T_ = HW;
        for (int it = tid; it < TM * 4; it += TS_NT) {
            const int R = it >> 2, hh = it & 3, m = m0 + R;
            const int Rs = R & ~(T_ - 1);
            const float *qp = slab + R * RS + hh * 8;
            float qv[8], o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                qv[e] = qp[e] * 0.35355339059327373f;
                o[e] = 0.f;
            }
            float mx = -INFINITY, den = 0.f;
            for (int tk = 0; tk < T_; ++tk) {
                const float *kp = slab + (Rs + tk) * RS + 32 + hh * 8;
                float sc = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) sc = fmaf(qv[e], kp[e], sc);
                const float nm = fmaxf(mx, sc);
                const float corr = __expf(mx - nm), pw = __expf(sc - nm);
                den = den * corr + pw;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = o[e] * corr + pw * kp[32 + e];
                mx = nm;
            }
            const float inv = 1.0f / den;
            if (m < M) {
                v8 ov;
#pragma unroll
                for (int e = 0; e < 8; ++e) ov[e] = (T)(o[e] * inv);
                store_wt(reinterpret_cast<v8 *>((T *)a.attn_out + (size_t)m * Cout + nt * 32 + hh * 8), ov);
            }
        }
        return;
    } else {
        // one item per thread (TM * 4 <= 512): its 8 values stay in registers through the GroupNorm of every consumer
        static_assert(ITEMS <= TS_NT, "one item per thread");
        const bool active = tid < ITEMS;
        const int R = tid >> 2, c8 = (tid & 3) * 8, m = m0 + R;
        float v[8];
        if (active) {
            const float *p = slab + R * RS + c8;
            f32x4 v0 = *reinterpret_cast<const f32x4 *>(p), v1 = *reinterpret_cast<const f32x4 *>(p + 4);
#pragma unroll
            for (int k = 1; k < NSLAB; ++k) {
                v0 += *reinterpret_cast<const f32x4 *>(p + k * SLAB);
                v1 += *reinterpret_cast<const f32x4 *>(p + k * SLAB + 4);
            }
            const float *ar = rows + (R >> hwlog) * TN + c8;
            v0 += *reinterpret_cast<const f32x4 *>(ar);
            v1 += *reinterpret_cast<const f32x4 *>(ar + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = v0[e];
                v[4 + e] = v1[e];
            }
            if (has_res) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += (float)rres[e];
            }
            if (a.raw_out && m < M) {
                v8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (T)v[e];
                store_wt(reinterpret_cast<v8 *>((T *)a.raw_out + (size_t)m * Cout + n0 + c8), o);
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = 0.f;
        }
        mark(6);
        if (a.nreq == 0) return;

        // GroupNorm (+SiLU) of the tile for each consumer; a (sample, group) = HW rows x gs channels lives in lanes
        // {bits 0..1: 8-channel column, bits 2..5: row} of 1 (HW <= 16) or 4 (HW = 64) waves.  Two passes (mean, then
        // centred sum of squares), each a fixed-order butterfly: DPP inside 16-lane rows, swizzle / permute across them,
        // LDS across the four waves of an 8x8 sample.
        auto dpp_add = [](float x, auto ctrl) {
            constexpr int C = decltype(ctrl)::value;
            return x + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), C, 0xf, 0xf, false));
        };
        auto group_sum = [&](float x, int gs) {
            if (gs >= 16) x = dpp_add(x, IC<0xB1>{});          // quad_perm [1,0,3,2]: lane ^ 1
            if (gs >= 32) x = dpp_add(x, IC<0x4E>{});          // quad_perm [2,3,0,1]: lane ^ 2
            x = dpp_add(x, IC<0x124>{});                       // row_ror:4, row_ror:8: the four lanes = l (mod 4) of a row
            x = dpp_add(x, IC<0x128>{});
            if (hwlog >= 4) {
                // rows 0+1 / 2+3, then the two halves: v_permlane16_swap / v_permlane32_swap exchange rows between two
                // registers holding the same value (written as instructions: the builtin's two results were folded into
                // one by the compiler)
                unsigned a = __builtin_bit_cast(unsigned, x), b;
                asm volatile("v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "=&v"(b));
                x = __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
                a = __builtin_bit_cast(unsigned, x);
                asm volatile("v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "=&v"(b));
                x = __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
            }
            return x;
        };
        float *xw = stat;                                      // [2 passes][3 consumers][8 waves][4 columns]
        const int nreq = a.nreq;
        float mean[3], rstd[3];
        int gsz[3];
        gsz[0] = a.req[0].gs;
        gsz[1] = a.req[1].gs;
        gsz[2] = a.req[2].gs;
        float s8 = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) s8 += v[e];
#pragma unroll
        for (int rq = 0; rq < 3; ++rq) {
            if (rq < nreq) {
                mean[rq] = group_sum(s8, gsz[rq]);
                if (hwlog == 6 && l < 4) xw[(rq * 8 + w) * 4 + l] = mean[rq];
            }
        }
        if (hwlog == 6) {
            __syncthreads();
#pragma unroll
            for (int rq = 0; rq < 3; ++rq)
                if (rq < nreq) {
                    const float *pw = xw + (rq * 8 + (w & 4)) * 4 + (l & 3);
                    mean[rq] = (pw[0] + pw[4]) + (pw[8] + pw[12]);
                }
        }
#pragma unroll
        for (int rq = 0; rq < 3; ++rq) {
            if (rq < nreq) {
                mean[rq] *= 1.0f / (float)(gsz[rq] << hwlog);
                float d2 = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float dl = v[e] - mean[rq];
                    d2 = fmaf(dl, dl, d2);
                }
                rstd[rq] = group_sum(d2, gsz[rq]);
                if (hwlog == 6 && l < 4) xw[96 + (rq * 8 + w) * 4 + l] = rstd[rq];
            }
        }
        if (hwlog == 6) {
            __syncthreads();
#pragma unroll
            for (int rq = 0; rq < 3; ++rq)
                if (rq < nreq) {
                    const float *pw = xw + 96 + (rq * 8 + (w & 4)) * 4 + (l & 3);
                    rstd[rq] = (pw[0] + pw[4]) + (pw[8] + pw[12]);
                }
        }
        if (active && m < M) {
#pragma unroll
            for (int rq = 0; rq < 3; ++rq) {
                if (rq < nreq) {
                    const float rs_ = 1.0f / sqrtf(rstd[rq] / (float)(gsz[rq] << hwlog) + a.eps);
                    const float *gp = rows + 16 * TN + rq * 2 * TN + c8;
                    const f32x4 g0 = *reinterpret_cast<const f32x4 *>(gp), g1 = *reinterpret_cast<const f32x4 *>(gp + 4);
                    const f32x4 b0 = *reinterpret_cast<const f32x4 *>(gp + TN), b1 = *reinterpret_cast<const f32x4 *>(gp + TN + 4);
                    const int silu = rq == 0 ? a.req[0].silu : (rq == 1 ? a.req[1].silu : a.req[2].silu);
                    void *outp = rq == 0 ? a.req[0].out : (rq == 1 ? a.req[1].out : a.req[2].out);
                    v8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float gm = e < 4 ? g0[e & 3] : g1[e & 3], bt = e < 4 ? b0[e & 3] : b1[e & 3];
                        float f = fmaf((v[e] - mean[rq]) * rs_, gm, bt);
                        if (silu) f = f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * f));
                        o[e] = (T)f;
                    }
                    store_wt(reinterpret_cast<v8 *>((T *)outp + (size_t)m * Cout + n0 + c8), o);
                }
            }
        }
        mark(7);
    }
}

template <typename T, int TM, int NB, int D> int launch_tail_t(const TailArgs &a, hipStream_t st) {
    constexpr int smem = ts_smem_bytes<TM, NB>();
    static_assert(smem <= 160 * 1024, "LDS budget");
    static bool attr = false;
    if (!attr) {
        BNDM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_s<T, TM, NB, D>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr = true;
    }
    const int M = a.B << a.hwlog, ntm = (M + TM - 1) / TM;
    hipLaunchKernelGGL((conv_s<T, TM, NB, D>), dim3(a.ntn, ntm), dim3(TS_NT), smem, st, a);
    return launch_status("conv_s");
}

}  // namespace

int tail_ring_depth(int nb) { return nb == 1 ? 8 : 2; }

int launch_conv_tail(int dtype, int TM, int NB, const TailArgs &a, hipStream_t st) {
    if (!((TM == 128 && NB == 1) || (TM == 64 && NB == 1) || (TM == 64 && NB == 3))) {
        set_error("conv_s: tile %d x %d not built", TM, NB * 32);
        return BNDM_E_ARG;
    }
    if (dtype == BNDM_DTYPE_F16) {
        if (NB == 3) return launch_tail_t<_Float16, 64, 3, 2>(a, st);
        return TM == 128 ? launch_tail_t<_Float16, 128, 1, 8>(a, st) : launch_tail_t<_Float16, 64, 1, 8>(a, st);
    }
    if (NB == 3) return launch_tail_t<__bf16, 64, 3, 2>(a, st);
    return TM == 128 ? launch_tail_t<__bf16, 128, 1, 8>(a, st) : launch_tail_t<__bf16, 64, 1, 8>(a, st);
}

// ------------------------------------------------------------------------------------------------
// Host side: rounds, per-wave step lists and the weight stream in consumption order
// ------------------------------------------------------------------------------------------------
TailPlan build_tail_plan(const std::vector<TailSeg> &segs, int Cout_rows, int NB, int D,
                         const std::function<float(int, int, int, int)> &w_of,
                         const std::function<int(int, int, int)> &row_of) {
    TailPlan p;
    struct Tap { int ts, wt; };
    struct Rnd { int seg, c0, nsub, mode, phase; std::vector<Tap> taps; };
    std::vector<Rnd> rounds;
    auto add_rounds = [&](int si, int mode, int phase, const std::vector<Tap> &taps) {
        for (int c0 = 0; c0 < segs[si].C; c0 += 256) {
            const int n = std::min(256, segs[si].C - c0);
            rounds.push_back(Rnd{si, c0, n / 32, mode, phase, taps});
        }
    };
    for (int pass = 0; pass < 2; ++pass)                  // 3x3 sources first, 1x1 sources after them
        for (int si = 0; si < (int)segs.size(); ++si) {
            const TailSeg &s = segs[si];
            if ((s.kind == TAIL_SEG_1x1) != (pass == 1)) continue;
            if (s.kind == TAIL_SEG_1x1) {
                add_rounds(si, 0, 0, {Tap{4, 0}});
            } else if (s.kind == TAIL_SEG_3x3 || s.kind == TAIL_SEG_3x3_UP) {
                std::vector<Tap> t9;
                for (int t = 0; t < 9; ++t) t9.push_back(Tap{t, t});
                add_rounds(si, s.kind == TAIL_SEG_3x3_UP ? 1 : 0, 0, t9);
            } else {
                // stride 2, pad 1: tap d reads source coordinate 2 y + d - 1 = 2 (y + o) + phase with
                // d = 0 -> (o, phase) = (-1, 1), d = 1 -> (0, 0), d = 2 -> (0, 1): four phase images, 4 / 2 / 2 / 1 taps
                for (int ph : {3, 2, 1, 0}) {
                    const int py = ph >> 1, px = ph & 1;
                    std::vector<Tap> tp;
                    for (int dy = 0; dy < 3; ++dy)
                        for (int dx = 0; dx < 3; ++dx) {
                            const int phy = dy == 1 ? 0 : 1, phx = dx == 1 ? 0 : 1;
                            if (phy != py || phx != px) continue;
                            const int oy = dy == 0 ? -1 : 0, ox = dx == 0 ? -1 : 0;
                            tp.push_back(Tap{(oy + 1) * 3 + (ox + 1), dy * 3 + dx});
                        }
                    add_rounds(si, 2, ph, tp);
                }
            }
        }
    p.nrounds = (int)rounds.size();
    for (const Rnd &r : rounds) p.rounds.push_back(TailPlanRound{r.seg, r.c0, r.nsub, r.mode, r.phase});

    // steps dealt round-robin to the waves; every wave has at least one entry per round (a zero-weight entry on the zero
    // row if it got no step) and the last entry of a round carries the boundary flag
    struct Ent { uint32_t d; int rnd, j, wt; bool real; };
    std::vector<Ent> lists[TS_NW];
    int g = 0;
    for (int r = 0; r < (int)rounds.size(); ++r) {
        const Rnd &rd = rounds[r];
        size_t first[TS_NW];
        for (int w = 0; w < TS_NW; ++w) first[w] = lists[w].size();
        for (int j = 0; j < rd.nsub; ++j)
            for (const Tap &t : rd.taps) {
                const int w = g++ % TS_NW;
                lists[w].push_back(Ent{(uint32_t)(t.ts | (j << 4) | ((r & 1) << 7)), r, j, t.wt, true});
            }
        for (int w = 0; w < TS_NW; ++w) {
            if (lists[w].size() == first[w])
                lists[w].push_back(Ent{(uint32_t)(TS_ZSLOT | ((r & 1) << 7) | 0x2000u), r, 0, 0, false});
            if (r + 1 < (int)rounds.size()) {
                const int cnt = (int)std::min<size_t>(lists[w].size() - first[w], (size_t)D + 1);
                lists[w].back().d |= 0x100u | ((uint32_t)cnt << 9);
            }
        }
    }
    size_t mx = 0;
    for (int w = 0; w < TS_NW; ++w) {
        mx = std::max(mx, lists[w].size());
        p.nuse[w] = (int)lists[w].size();                 // every list ends with a real or a boundary entry
    }
    p.maxsteps = (int)((mx + D - 1) / D) * D;
    for (int w = 0; w < TS_NW; ++w)
        while ((int)lists[w].size() < p.maxsteps) lists[w].push_back(Ent{(uint32_t)(TS_ZSLOT | 0x2000u), 0, 0, 0, false});
    p.desc.resize((size_t)TS_NW * p.maxsteps);
    for (int w = 0; w < TS_NW; ++w)
        for (int s = 0; s < p.maxsteps; ++s) p.desc[(size_t)w * p.maxsteps + s] = lists[w][s].d;

    // weight stream: [n-tile][wave][entry][fragment ks * NB + nb][lane][8]
    const int NL = 2 * NB, TN = NB * 32, ntn = (Cout_rows + TN - 1) / TN;
    p.ntn = ntn;
    p.wave_elems = (size_t)p.maxsteps * NL * 512;
    p.tile_elems = p.wave_elems * TS_NW;
    p.wgt.assign(p.tile_elems * ntn, 0.f);
    for (int nt = 0; nt < ntn; ++nt)
        for (int w = 0; w < TS_NW; ++w)
            for (int s = 0; s < p.maxsteps; ++s) {
                const Ent &en = lists[w][s];
                if (!en.real) continue;
                const Rnd &rd = rounds[en.rnd];
                float *dst = &p.wgt[(size_t)nt * p.tile_elems + (size_t)w * p.wave_elems + (size_t)s * NL * 512];
                for (int ks = 0; ks < 2; ++ks)
                    for (int nb = 0; nb < NB; ++nb)
                        for (int ln = 0; ln < 64; ++ln) {
                            const int row = row_of(nt, nb, ln & 31);
                            if (row < 0) continue;
                            const int c = rd.c0 + 32 * en.j + 16 * ks + 8 * (ln >> 5);
                            float *o = dst + ((size_t)(ks * NB + nb) * 64 + ln) * 8;
                            for (int e = 0; e < 8; ++e) o[e] = w_of(rd.seg, row, c + e, en.wt);
                        }
            }
    return p;
}

}  // namespace bndm
