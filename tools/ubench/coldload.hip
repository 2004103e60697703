// How long does the FIRST global load of a dependent kernel take?  A reader kernel (512 workgroups x 256 threads, every
// wave loads 16 B per lane from a 64 MB buffer at its own place) is launched behind: nothing (idle GPU), a writer of
// the same 64 MB with plain stores, the same writer with write-through (sc0 sc1) stores, itself, an empty kernel.
// Prints the s_memtime clocks from the reader's first instruction to the data, per workgroup: min / mean / max.
//   hipcc -O3 --offload-arch=gfx950 coldload.hip -o coldload.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void writer(u32x4 *buf, size_t n16, int wt) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        u32x4 v = {(unsigned)i, 1u, 2u, 3u};
        if (wt) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(buf + i), "v"(v) : "memory");
        else buf[i] = v;
    }
}
__global__ void empty() {}
__global__ __launch_bounds__(256, 2) void reader(const u32x4 *buf, size_t n16, unsigned *out, unsigned *sink, int second) {
    extern __shared__ char smem[];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const size_t i = ((size_t)blockIdx.x * 131 + (threadIdx.x >> 6) * 37) * 4096 % (n16 - 64) + (threadIdx.x & 63);
    u32x4 v = buf[i];
    asm volatile("s_waitcnt vmcnt(0)" ::"v"(v) : "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned d2 = 0;
    if (second) {                                   // a second, dependent round trip
        u32x4 w = buf[(i + (v[0] & 1) + 77777) % n16];
        asm volatile("s_waitcnt vmcnt(0)" ::"v"(w) : "memory");
        d2 = (unsigned)(__builtin_amdgcn_s_memtime() - t1);
        v[1] += w[1];
    }
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = (unsigned)(t1 - t0); out[blockIdx.x * 2 + 1] = d2; }
    if (v[1] == 0xdeadbeef) sink[0] = v[2];
    smem[threadIdx.x] = 0;
}
static void report(const char *name, unsigned *d_out, int G) {
    std::vector<unsigned> h(G * 2);
    hipMemcpy(h.data(), d_out, G * 8, hipMemcpyDeviceToHost);
    for (int k = 0; k < 2; ++k) {
        unsigned mn = ~0u, mx = 0; double s = 0;
        for (int b = 0; b < G; ++b) { mn = std::min(mn, h[b * 2 + k]); mx = std::max(mx, h[b * 2 + k]); s += h[b * 2 + k]; }
        printf("%-44s %s load: min %6u  mean %8.0f  max %6u clk\n", name, k ? "second" : "first ", mn, s / G, mx);
    }
}
int main() {
    const size_t bytes = 64u << 20, n16 = bytes / 16;
    const int G = 512;
    u32x4 *buf; unsigned *out, *sink;
    hipMalloc(&buf, bytes); hipMalloc(&out, G * 8); hipMalloc(&sink, 64);
    hipFuncSetAttribute((const void *)reader, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    auto R = [&]() { hipLaunchKernelGGL(reader, dim3(G), dim3(256), 80 * 1024, 0, buf, n16, out, sink, 1); };
    hipLaunchKernelGGL(writer, dim3(2048), dim3(256), 0, 0, buf, n16, 0); hipDeviceSynchronize();
    for (int rep = 0; rep < 2; ++rep) {
        hipDeviceSynchronize(); R(); hipDeviceSynchronize(); report("behind nothing (idle, data in HBM/MALL)", out, G);
        hipLaunchKernelGGL(writer, dim3(2048), dim3(256), 0, 0, buf, n16, 0); R(); hipDeviceSynchronize(); report("behind a 64 MB writer, plain stores", out, G);
        hipLaunchKernelGGL(writer, dim3(2048), dim3(256), 0, 0, buf, n16, 1); R(); hipDeviceSynchronize(); report("behind a 64 MB writer, write-through stores", out, G);
        R(); R(); hipDeviceSynchronize(); report("behind itself", out, G);
        hipLaunchKernelGGL(empty, dim3(1), dim3(64), 0, 0); R(); hipDeviceSynchronize(); report("behind an empty kernel", out, G);
    }
    return 0;
}
