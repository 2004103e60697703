// Which CUs does a stream created with hipExtStreamCreateWithCUMask really run on?  For each mask layout that
// tools/experiments/lanes.patch uses (chain k of n gets bits [k*32/n, (k+1)*32/n) of EVERY 32-bit mask word) the probe launches
// 2048 one-wave workgroups on the masked stream and prints, per XCC, how many distinct (se, sh, cu) slots they ran on -- the patch
// assumes "an equal, disjoint share of the CUs of every XCD".  Also prints the overlap between the shares of two chains.
//   hipcc -O3 --offload-arch=gfx950 cumask_probe.hip -o cumask_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <set>
#include <vector>
__global__ void probe(unsigned *out) {
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);          // HW_REG_HW_ID: cu [11:8], sh [12], se [15:13]
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u;   // HW_REG_XCC_ID
    if (threadIdx.x == 0) out[blockIdx.x] = (xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15);
    for (int i = 0; i < 20; ++i) __builtin_amdgcn_s_sleep(100);             // stay resident so that the launch spreads
}
static std::set<unsigned> run(hipStream_t st, unsigned *d, int G) {
    hipLaunchKernelGGL(probe, dim3(G), dim3(64), 0, st, d);
    hipStreamSynchronize(st);
    std::vector<unsigned> h(G);
    hipMemcpy(h.data(), d, G * 4, hipMemcpyDeviceToHost);
    return std::set<unsigned>(h.begin(), h.end());
}
static void show(const char *what, const std::set<unsigned> &s) {
    std::map<unsigned, int> per;
    for (unsigned k : s) per[k >> 8]++;
    printf("%-34s %3zu CUs; per XCC:", what, s.size());
    for (auto &kv : per) printf(" %u:%d", kv.first, kv.second);
    printf("\n");
}
int main() {
    const int G = 2048;
    unsigned *d;
    hipMalloc(&d, G * 4);
    show("unmasked (NULL stream)", run(nullptr, d, G));
    for (int n : {2, 4}) {
        std::vector<std::set<unsigned>> share;
        for (int k = 0; k < n; ++k) {
            const int per = 32 / n;
            uint32_t words[8];
            for (uint32_t &w : words) w = (uint32_t)(((1ull << per) - 1) << (k * per));
            hipStream_t st;
            if (hipExtStreamCreateWithCUMask(&st, 8, words) != hipSuccess) {
                printf("hipExtStreamCreateWithCUMask failed\n");
                return 1;
            }
            char what[64];
            snprintf(what, sizeof what, "%d chains, chain %d (word %08x)", n, k, words[0]);
            share.push_back(run(st, d, G));
            show(what, share.back());
            hipStreamDestroy(st);
        }
        int both = 0;
        for (unsigned c : share[0]) both += (int)share[1].count(c);
        printf("   CUs in both chain 0 and chain 1: %d\n", both);
    }
    // the other natural layout: chain k gets whole mask words (words 2k, 2k+1 of 8 for n = 4)
    for (int k = 0; k < 4; ++k) {
        uint32_t words[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        words[2 * k] = words[2 * k + 1] = 0xffffffffu;
        hipStream_t st;
        if (hipExtStreamCreateWithCUMask(&st, 8, words) != hipSuccess) return 1;
        char what[64];
        snprintf(what, sizeof what, "whole words %d,%d", 2 * k, 2 * k + 1);
        show(what, run(st, d, G));
        hipStreamDestroy(st);
    }
    return 0;
}
