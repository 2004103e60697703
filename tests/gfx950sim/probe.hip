// TEST INFRASTRUCTURE: probe kernels for tests/test_gfx950sim_isa.py.  Everything here is written with HIP-level operations whose
// result is defined by the language / the programming guide, NOT by this repository's kernels: whatever instructions hipcc picks
// for them (DPP, ds_bpermute, permlane swaps, SDWA, v_div_* sequences, v_rcp_iflag, v_mul_hi, MFMA ...) the simulator must
// reproduce the defined result.  An independent check of tests/gfx950sim's instruction semantics.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// out[k][tid]: k-th probe of each thread (block of 256 threads = 4 waves; each wave's lanes see only their wave)
extern "C" __global__ void probe_lanes(const int *in, int *out, unsigned long long *out64) {
    const int tid = threadIdx.x, n = blockDim.x;
    const int v = in[tid];
    int k = 0;
    for (int m = 1; m < 64; m <<= 1) out[(k++) * n + tid] = __shfl_xor(v, m);
    out[(k++) * n + tid] = __shfl_up(v, 1);
    out[(k++) * n + tid] = __shfl_up(v, 3);
    out[(k++) * n + tid] = __shfl_up(v, 16);
    out[(k++) * n + tid] = __shfl_down(v, 1);
    out[(k++) * n + tid] = __shfl_down(v, 5);
    out[(k++) * n + tid] = __shfl_down(v, 32);
    out[(k++) * n + tid] = __shfl(v, 17);
    out[(k++) * n + tid] = __shfl(v, (tid * 7 + 3) & 63);
    out[(k++) * n + tid] = __shfl_xor(v, 1, 16);           // width 16
    out[(k++) * n + tid] = __shfl_up(v, 2, 8);             // width 8: lanes below 2 in their group keep their own value
    const unsigned long long b = __ballot(v & 1);
    out64[tid] = b;
    out[(k++) * n + tid] = __popcll(b);
    out[(k++) * n + tid] = __lane_id();
    out[(k++) * n + tid] = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0));
    // a wave-level sum by the classic butterfly
    int s = v;
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
    out[(k++) * n + tid] = s;
    // divergent: only odd lanes vote
    unsigned long long b2 = 0;
    if (tid & 1) b2 = __ballot(v > 0);
    out64[n + tid] = b2;
    out[(k++) * n + tid] = __builtin_amdgcn_readfirstlane(v);
}

// integer / float arithmetic whose results are defined by C: a[i] op b[i]
extern "C" __global__ void probe_arith(const int *ia, const int *ib, const float *fa, const float *fb, int *iout, float *fout, double *dout,
                                       unsigned long long *lout, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int a = ia[i], b = ib[i];
    const unsigned ua = (unsigned)a, ub = (unsigned)b;
    int k = 0;
    iout[(k++) * n + i] = b ? a / b : 0;
    iout[(k++) * n + i] = b ? a % b : 0;
    iout[(k++) * n + i] = ub ? (int)(ua / ub) : 0;
    iout[(k++) * n + i] = ub ? (int)(ua % ub) : 0;
    iout[(k++) * n + i] = a * b;
    iout[(k++) * n + i] = __mulhi(a, b);
    iout[(k++) * n + i] = (int)__umulhi(ua, ub);
    iout[(k++) * n + i] = a >> (b & 31);
    iout[(k++) * n + i] = (int)(ua >> (b & 31));
    iout[(k++) * n + i] = a << (b & 31);
    iout[(k++) * n + i] = __clz(a);
    iout[(k++) * n + i] = __popc(ua);
    iout[(k++) * n + i] = __brev(ua);
    iout[(k++) * n + i] = min(a, b);
    iout[(k++) * n + i] = (int)max(ua, ub);
    iout[(k++) * n + i] = abs(a);
    iout[(k++) * n + i] = (a & 0xffff) * (b & 0xffff) + 7;                  // 16 / 24-bit multiply forms
    iout[(k++) * n + i] = (short)a + (short)b;                              // SDWA candidates
    iout[(k++) * n + i] = (unsigned char)(a >> 8) + (unsigned char)(b >> 16);
    iout[(k++) * n + i] = __byte_perm(ua, ub, 0x5140);
    iout[(k++) * n + i] = __byte_perm(ua, ub, 0x3276);
    const long long la = ((long long)a << 20) + b, lb = ((long long)b << 7) - a;
    lout[0 * n + i] = (unsigned long long)(la * lb);
    lout[1 * n + i] = (unsigned long long)(la >> (b & 63));
    lout[2 * n + i] = (unsigned long long)la << (a & 63);
    lout[3 * n + i] = (unsigned long long)(la + lb);
    lout[4 * n + i] = lb ? (unsigned long long)(la / lb) : 0;
    const float x = fa[i], y = fb[i];
    k = 0;
    fout[(k++) * n + i] = x / y;
    fout[(k++) * n + i] = sqrtf(fabsf(x));
    fout[(k++) * n + i] = fmaf(x, y, 0.25f);
    fout[(k++) * n + i] = exp2f(x * 0.125f);
    fout[(k++) * n + i] = __expf(x * 0.125f);
    fout[(k++) * n + i] = rsqrtf(fabsf(y) + 1.0f);
    fout[(k++) * n + i] = fminf(x, y);
    fout[(k++) * n + i] = fmaxf(fmaxf(x, y), 0.5f);
    fout[(k++) * n + i] = floorf(x * 3.7f);
    fout[(k++) * n + i] = rintf(x * 3.7f);
    fout[(k++) * n + i] = truncf(x * 3.7f);
    fout[(k++) * n + i] = (float)a;
    fout[(k++) * n + i] = (float)ua;
    fout[(k++) * n + i] = (float)(int)(x * 1000.f);
    fout[(k++) * n + i] = (float)(unsigned)(fabsf(x) * 1000.f);
    fout[(k++) * n + i] = __half2float(__float2half(x));
    fout[(k++) * n + i] = (float)(__bf16)x;
    fout[(k++) * n + i] = (float)((_Float16)x + (_Float16)y);
    fout[(k++) * n + i] = ldexpf(x, b & 7);
    fout[(k++) * n + i] = x < y ? x : (x == y ? 0.f : y);
    const double dx = x, dy = y;
    dout[0 * n + i] = dx / dy;
    dout[1 * n + i] = sqrt(fabs(dx) + 1.0);
    dout[2 * n + i] = dx * dy + 0.5;
    dout[3 * n + i] = 1.0 / sqrt(fabs(dy) + 2.0);
    dout[4 * n + i] = (double)a * 0.5;
}

// MFMA with the programming guide's documented fragment layouts (cdna_hip_programming.md section 3):
//   32x32x16: A[i = l & 31][k = 8 (l >> 5) + e], B[k = 8 (l >> 5) + e][j = l & 31], D: col = l & 31, row = (r & 3) + 8 (r >> 2) + 4 (l >> 5)
//   16x16x32: A[i = l & 15][k = 8 (l >> 4) + e], B[k][j = l & 15],                  D: col = l & 15, row = 4 (l >> 4) + r
//   32x32x2 f32: A[i = l & 31][k = l >> 5], B[k][j = l & 31];  16x16x4 f32: A[l & 15][k = l >> 4], B[k][l & 15]
// (the k mapping of A and B only has to be the SAME for both; the row / column mapping is what this checks)
extern "C" __global__ void probe_mfma(const _Float16 *A32, const _Float16 *B32, const float *C32, float *D32,      // [32][16], [16][32], [32][32]
                                      const _Float16 *A16, const _Float16 *B16, float *D16,                       // [16][32], [32][16], [16][16]
                                      const float *Af, const float *Bf, float *Df32, float *Df16,                // [32][4] / [4][32] shared by both f32 shapes
                                      const __bf16 *Ab, const __bf16 *Bb, float *Db) {                           // bf16 32x32x16
    const int l = threadIdx.x;
    {
        f16x8 a, b;
        for (int e = 0; e < 8; ++e) {
            a[e] = A32[(l & 31) * 16 + 8 * (l >> 5) + e];
            b[e] = B32[(8 * (l >> 5) + e) * 32 + (l & 31)];
        }
        f32x16 c;
        for (int r = 0; r < 16; ++r) c[r] = C32[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)];
        const f32x16 d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
        for (int r = 0; r < 16; ++r) D32[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = d[r];
        bf16x8 ab, bb;
        for (int e = 0; e < 8; ++e) {
            ab[e] = Ab[(l & 31) * 16 + 8 * (l >> 5) + e];
            bb[e] = Bb[(8 * (l >> 5) + e) * 32 + (l & 31)];
        }
        f32x16 z;
        for (int r = 0; r < 16; ++r) z[r] = 0.f;
        const f32x16 db = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, z, 0, 0, 0);
        for (int r = 0; r < 16; ++r) Db[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = db[r];
    }
    {
        f16x8 a, b;
        for (int e = 0; e < 8; ++e) {
            a[e] = A16[(l & 15) * 32 + 8 * (l >> 4) + e];
            b[e] = B16[(8 * (l >> 4) + e) * 16 + (l & 15)];
        }
        f32x4 c = {0.f, 0.f, 0.f, 0.f};
        const f32x4 d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
        for (int r = 0; r < 4; ++r) D16[(4 * (l >> 4) + r) * 16 + (l & 15)] = d[r];
    }
    {
        f32x16 c;
        for (int r = 0; r < 16; ++r) c[r] = 0.f;
        // two k-steps of 32x32x2 cover k = 0..3
        for (int s = 0; s < 2; ++s) c = __builtin_amdgcn_mfma_f32_32x32x2f32(Af[(l & 31) * 4 + 2 * s + (l >> 5)], Bf[(2 * s + (l >> 5)) * 32 + (l & 31)], c, 0, 0, 0);
        for (int r = 0; r < 16; ++r) Df32[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
        f32x4 c4 = {0.f, 0.f, 0.f, 0.f};
        c4 = __builtin_amdgcn_mfma_f32_16x16x4f32(Af[(l & 15) * 4 + (l >> 4)], Bf[(l >> 4) * 32 + (l & 15)], c4, 0, 0, 0);
        for (int r = 0; r < 4; ++r) Df16[(4 * (l >> 4) + r) * 16 + (l & 15)] = c4[r];
    }
}

// LDS: transposition through shared memory with a barrier, 16-bit and 128-bit accesses, an atomic histogram
extern "C" __global__ void probe_lds(const float *in, float *out, int *hist) {
    __shared__ float tile[32][33];
    __shared__ int h[16];
    __shared__ __attribute__((aligned(16))) unsigned short hs[256];
    const int t = threadIdx.x, x = t & 31, y = t >> 5;          // 256 threads: 8 rows per pass
    if (t < 16) h[t] = 0;
    for (int r = 0; r < 32; r += 8) tile[y + r][x] = in[(y + r) * 32 + x];
    hs[t] = (unsigned short)(t * 257);
    __syncthreads();
    for (int r = 0; r < 32; r += 8) out[(y + r) * 32 + x] = tile[x][y + r] + (float)hs[255 - t];
    atomicAdd(&h[(t * 7) & 15], t);
    __syncthreads();
    if (t < 16) hist[t] = h[t];
    if (t < 32) {
        const uint4 v = reinterpret_cast<const uint4 *>(hs)[t];
        reinterpret_cast<uint4 *>(out + 1024)[t] = v;
    }
}

// deliberately WRONG: the loaded register is used before any s_waitcnt covers the load (the hazard recogniser of the simulator must fire;
// hardware would read the stale register here)
extern "C" __global__ void probe_missing_wait(const int *in, int *out) {
    int v;
    asm volatile("global_load_dword %0, %1, off\n\tv_add_u32 %0, %0, %0\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(in + threadIdx.x) : "memory");
    out[threadIdx.x] = v;
}

// deliberately WRONG: an LDS-DMA lands 1 KiB per wave in shared memory, and the wave reads it back without waiting for vmcnt
extern "C" __global__ void probe_lds_dma_race(const float *in, float *out) {
    __shared__ __attribute__((aligned(16))) float buf[256];
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(in + 4 * threadIdx.x),
                                     (__attribute__((address_space(3))) void *)buf, 16, 0, 0);
    out[threadIdx.x] = buf[threadIdx.x];                  // (no wait: the compiler does not track the DMA's destination)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
