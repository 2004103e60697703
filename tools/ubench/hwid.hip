// Which hardware slots do the workgroups of a 2-per-CU launch get?  Prints, per (xcc, se, cu), the blocks resident there
// and the HW_ID.wave_id of their first wave.   hipcc -O3 --offload-arch=gfx950 hwid.hip -o hwid.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
#include <algorithm>
__global__ __launch_bounds__(256, 2) void probe(unsigned *out) {
    extern __shared__ char smem[];
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);          // HW_REG_HW_ID
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u;   // HW_REG_XCC_ID
    const unsigned long long t = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) {
        out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + 0] = hw;
        out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + 1] = xcc;
        out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + 2] = (unsigned)(t >> 6);
    }
    smem[threadIdx.x] = 1;
    for (int i = 0; i < 200; ++i) __builtin_amdgcn_s_sleep(100);            // stay resident
}
int main() {
    const int G = 1024;
    unsigned *d;
    hipMalloc(&d, G * 16 * 4);
    hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    hipLaunchKernelGGL(probe, dim3(G), dim3(256), 80 * 1024, 0, d);
    hipDeviceSynchronize();
    std::vector<unsigned> h(G * 16);
    hipMemcpy(h.data(), d, G * 16 * 4, hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<std::pair<unsigned, int>>> cu;
    unsigned t0 = ~0u;
    for (int b = 0; b < G; ++b) t0 = std::min(t0, h[b * 16 + 2]);
    int hist[16] = {0};
    for (int b = 0; b < G; ++b) {
        const unsigned hw = h[b * 16], xcc = h[b * 16 + 1];
        const unsigned key = (xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15);
        cu[key].push_back({h[b * 16 + 2] - t0, b});
        if (b < 512) hist[hw & 15]++;
    }
    printf("CUs seen: %zu; wave_id histogram of blocks 0..511 (wave 0):", cu.size());
    for (int i = 0; i < 16; ++i) printf(" %d", hist[i]);
    printf("\n");
    int n = 0;
    for (auto &kv : cu) {
        if (n++ % 37) continue;
        std::sort(kv.second.begin(), kv.second.end());
        printf("xcc %u se %u sh %u cu %2u:", kv.first >> 8, (kv.first >> 5) & 7, (kv.first >> 4) & 1, kv.first & 15);
        for (auto &p : kv.second) {
            const unsigned hw = h[p.second * 16];
            printf("  b%d(t=%u,wid=%u,simd=%u | w1 wid=%u simd=%u)", p.second, p.first, hw & 15, (hw >> 4) & 3, h[p.second * 16 + 4] & 15,
                   (h[p.second * 16 + 4] >> 4) & 3);
        }
        printf("\n");
    }
    return 0;
}
