// Microbenchmark: how fast can one CU bring L2-resident bytes in, as a function of the number of waves issuing,
// for (0) buffer_load ... lds b128, (1) global_load_lds b128, (2) global_load_dwordx4 into VGPRs.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench/dma_rate.hip -o gpurun_out/dma_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

typedef __attribute__((address_space(3))) void *lds_ptr_t;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int NW>
__global__ __launch_bounds__(NW * 64) void rate_kernel(const char *__restrict__ src, int iters, int win_bytes,
                                                       unsigned long long *__restrict__ cyc, unsigned *__restrict__ sink) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const char *wg_src = src + (size_t)blockIdx.x * win_bytes;
    const int wave_span = win_bytes / NW;                    // bytes of the window this wave walks
    const char *wsrc = wg_src + wave * wave_span;
    char *wlds = lds + wave * 8192;                          // 8 KiB landing zone per wave (overwritten every iteration)
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)(uint64_t)wsrc);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)wsrc >> 32));
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(((uint64_t)hi << 32) | lo), 0, wave_span, 0x00020000);
    u32x4 acc = {0, 0, 0, 0};
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    int off = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int o = off + u * 1024 + lane * 16;
            if (MODE == 0) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(wlds + u * 1024), 16, o, 0, 0, 0);
            } else if (MODE == 3 || MODE == 4) {
                // 8 rows x 128 B per instruction, rows 16 KiB (MODE 3) or 2 KiB (MODE 4) apart, as a row-panel of a matrix is read
                const int rstride = MODE == 3 ? 16384 : 2048;
                const char *p = src + (size_t)(blockIdx.x & 63) * 128 * rstride / 64 + (size_t)((lane >> 3) + 8 * u + 64 * wave) * rstride + (lane & 7) * 16 + (off & (rstride - 1) & ~127);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)p, (lds_ptr_t)(wlds + u * 1024), 16, 0, 0);
            } else if (MODE == 1) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(wsrc + o),
                                                 (lds_ptr_t)(wlds + u * 1024), 16, 0, 0);
            } else {
                const u32x4 v = *(const u32x4 *)(wsrc + o);
                acc ^= v;
            }
        }
        if (MODE != 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        off += (MODE >= 3 ? 128 : 8192);
        if (off >= wave_span) off = 0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    if (MODE == 2 && (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
    if (MODE != 2 && lds[threadIdx.x * 4] == 77 && iters < 0) sink[1] = 1;
}

template <int MODE, int NW>
void run(const char *name, int nwg, const char *src, int win, unsigned long long *cyc, unsigned *sink) {
    const int iters = 400;
    const size_t smem = NW * 8192;
    hipFuncSetAttribute((const void *)rate_kernel<MODE, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((rate_kernel<MODE, NW>), dim3(nwg), dim3(NW * 64), smem, 0, src, iters, win, cyc, sink);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
    }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(nwg);
    hipMemcpy(h.data(), cyc, nwg * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto c : h) avg += (double)c; avg /= nwg;
    const double bytes_wg = (double)iters * 8192 * NW;
    const double wg_per_cu = nwg / 256.0;
    printf("%-22s NW=%2d wgs=%4d  %8.1f us  %7.2f TB/s  counter ticks/WG %9.0f  B/tick/CU %6.1f  B/clk/CU@%.0fus*2.1GHz %5.1f\n", name, NW, nwg,
           ms * 1e3, bytes_wg * nwg / (ms * 1e-3) / 1e12, avg, bytes_wg * (wg_per_cu < 1 ? 1 : wg_per_cu) / avg, ms * 1e3,
           bytes_wg * (wg_per_cu < 1 ? 1 : wg_per_cu) / (ms * 1e-3 * 2.1e9));
}


// One "K-step" of an implicit-GEMM workgroup: 16 MFMAs + 16 fragment reads per compute wave and 8 KiB of LDS-DMA per SIMD,
// with one barrier per step.  SPLIT=0: the four compute waves issue their own DMAs.  SPLIT=1: four extra waves (one per
// SIMD) issue them.
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
template <int SPLIT, int NDMA>
__global__ __launch_bounds__(SPLIT ? 512 : 256) void mix_kernel(const char *__restrict__ src, int iters, int win_bytes,
                                                                unsigned long long *__restrict__ cyc, float *__restrict__ sink) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool dma_wave = SPLIT ? wave >= 4 : true, mma_wave = SPLIT ? wave < 4 : true;
    const int w4 = wave & 3;
    const char *wsrc = src + (size_t)blockIdx.x * win_bytes + w4 * (win_bytes / 4);
    const int wave_span = win_bytes / 4;
    char *wlds = lds + w4 * 16384;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)(uint64_t)wsrc);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)wsrc >> 32));
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(((uint64_t)hi << 32) | lo), 0, wave_span, 0x00020000);
    f16v acc[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    int off = 0;
    for (int it = 0; it < iters; ++it) {
        if (dma_wave) {
#pragma unroll
            for (int u = 0; u < NDMA; ++u)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(wlds + ((it & 1) * NDMA + u) * 1024), 16,
                                                         off + u * 1024 + lane * 16, 0, 0, 0);
            off += NDMA * 1024;
            if (off >= wave_span) off = 0;
        }
        if (mma_wave) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                h8 a[2], b[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    a[i] = *(const h8 *)(lds + ((k * 4 + i) * 1024 + lane * 16));
                    b[i] = *(const h8 *)(lds + 32768 + ((k * 4 + i) * 1024 + lane * 16));
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i * 2 + j], 0, 0, 0);
            }
        }
        if (dma_wave) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
        __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    if (s == 12345.f) sink[0] = s;
}

template <int SPLIT, int NDMA>
void run_mix(int nwg, const char *src, int win, unsigned long long *cyc, unsigned *sink) {
    const int iters = 400;
    const size_t smem = 65536;
    hipFuncSetAttribute((const void *)mix_kernel<SPLIT, NDMA>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((mix_kernel<SPLIT, NDMA>), dim3(nwg), dim3(SPLIT ? 512 : 256), smem, 0, src, iters, win, cyc, (float *)sink);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
    }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(nwg);
    hipMemcpy(h.data(), cyc, nwg * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto c : h) avg += (double)c; avg /= nwg;
    printf("mix split=%d dma/step/wave=%d wgs=%4d  %8.1f us  ticks/step %7.1f  ns/step %7.1f  (16 MFMA 32x32x16 per wave per step = 512 pipe cycles)\n",
           SPLIT, NDMA, nwg, ms * 1e3, avg / iters, ms * 1e6 / iters);
}

int main() {
    const int win = 65536;                      // bytes per workgroup window: 256 wgs * 64 KiB = 16 MiB, L2-resident per XCD
    char *src; unsigned long long *cyc; unsigned *sink;
    hipMalloc(&src, (size_t)1024 * win);
    hipMemset(src, 1, (size_t)1024 * win);
    hipMalloc(&cyc, 1024 * 8);
    hipMalloc(&sink, 16);
    hipMemset(sink, 0, 16);
#define ROW(M, name)                                              \
    run<M, 1>(name, 256, src, win, cyc, sink);                    \
    run<M, 2>(name, 256, src, win, cyc, sink);                    \
    run<M, 4>(name, 256, src, win, cyc, sink);                    \
    run<M, 8>(name, 256, src, win, cyc, sink);                    \
    run<M, 16>(name, 256, src, win, cyc, sink);                   \
    run<M, 4>(name, 512, src, win, cyc, sink);                    \
    run<M, 4>(name, 1024, src, win, cyc, sink);
    ROW(0, "buffer_load_lds_b128")
    ROW(1, "global_load_lds_b128")
    ROW(2, "global_load_dwordx4")
    ROW(3, "lds_b128 8rows x128B @16K")
    ROW(4, "lds_b128 8rows x128B @2K")
    run_mix<0, 0>(256, src, win, cyc, sink);
    run_mix<0, 2>(256, src, win, cyc, sink);
    run_mix<0, 4>(256, src, win, cyc, sink);
    run_mix<0, 8>(256, src, win, cyc, sink);
    run_mix<1, 2>(256, src, win, cyc, sink);
    run_mix<1, 4>(256, src, win, cyc, sink);
    run_mix<1, 8>(256, src, win, cyc, sink);
    run_mix<0, 8>(512, src, win, cyc, sink);
    run_mix<1, 8>(512, src, win, cyc, sink);
    unsigned hs[4]; hipMemcpy(hs, sink, 16, hipMemcpyDeviceToHost);
    return hs[0] + hs[1] > 100;
}
