// In-kernel producer -> consumer hand-over across XCDs without a kernel boundary: what does it take, what does it cost?
// (The prerequisite of walking dependent layers inside one launch / with overlapping launches: DESIGN section 6a, last paragraph.)
// One launch of 2 * G workgroups (G = 256: one producer and one consumer per CU).  Producer g writes a 64 KB tile with write-through
// (sc0 sc1) stores, waits for them (s_waitcnt vmcnt(0)) and increments an agent-scope counter.  Consumer g (dispatched after every
// producer: blockIdx order) spins -- BOUNDED, 4 M polls -- on the counter with agent-scope loads until all G producers are in, then reads
// tile (g + 37) % G -- written on another CU, usually another XCD -- three ways: sc1 loads, sc0 sc1 loads, plain loads, and counts wrong
// pieces.  Every consumer first reads that tile with plain loads BEFORE the producers are done (the old pattern of the previous pass
// lands in its XCD's L2), so a load form that is served from a non-coherent L2 line returns the old pattern: three passes with
// three patterns, passes 1 and 2 are the ones that can show staleness.
//   hipcc -O3 --offload-arch=gfx950 flagwait.hip -o flagwait.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int G = 256, TILE16 = 4096;                     // 64 KB tiles of 16-byte pieces
__device__ __forceinline__ u32x4 ld_plain(const u32x4 *p) { return *p; }
__device__ __forceinline__ u32x4 ld_sc1(const u32x4 *p) {
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ u32x4 ld_sc01(const u32x4 *p) {
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__global__ __launch_bounds__(256) void handover(u32x4 *tiles, unsigned *cnt, unsigned target, unsigned pat, unsigned *res,
                                                unsigned long long *tm) {
    const int g = blockIdx.x % G, tid = threadIdx.x;
    if (blockIdx.x < G) {                                  // ---- producer
        u32x4 *t = tiles + (size_t)g * TILE16;
        for (int i = tid; i < TILE16; i += 256) {
            u32x4 v = {pat, (unsigned)g, (unsigned)i, pat ^ (unsigned)i};
            asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(t + i), "v"(v) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            tm[g] = __builtin_amdgcn_s_memtime();
        }
        return;
    }
    // ---- consumer: first pull the OLD contents of the tile it will read into this XCD's L2 (plain loads, before the producers are done),
    // so that a load form that may be served from a non-coherent L2 line shows up as stale data below
    {
        const u32x4 *t0 = tiles + (size_t)((g + 37) % G) * TILE16;
        unsigned acc = 0;
        for (int i = tid; i < TILE16; i += 256) acc += ld_plain(t0 + i)[3];
        if (acc == 0x12345678u) res[g * 8 + 5] = acc;          // (keep the loads)
    }
    __shared__ unsigned ok;
    if (tid == 0) {
        unsigned n = 0, seen = 0;
        for (; n < (4u << 20); ++n) {
            seen = __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (seen >= target) break;
            __builtin_amdgcn_s_sleep(4);
        }
        ok = seen >= target;
        tm[G + g] = __builtin_amdgcn_s_memtime();
        res[g * 8 + 7] = n;
    }
    __syncthreads();
    if (!ok) { if (tid == 0) res[g * 8 + 6] = 1; return; }   // gave up: reported, nothing hangs
    const u32x4 *t = tiles + (size_t)((g + 37) % G) * TILE16;
    const unsigned src = (unsigned)((g + 37) % G);
    unsigned bad[3] = {0, 0, 0};
    for (int i = tid; i < TILE16; i += 256) {
        const u32x4 a = ld_sc1(t + i), b = ld_sc01(t + i), c = ld_plain(t + i);        // coherent forms first
        bad[0] += !(c[0] == pat && c[1] == src && c[2] == (unsigned)i);
        bad[1] += !(a[0] == pat && a[1] == src && a[2] == (unsigned)i);
        bad[2] += !(b[0] == pat && b[1] == src && b[2] == (unsigned)i);
    }
    for (int k = 0; k < 3; ++k) atomicAdd(&res[g * 8 + k], bad[k]);
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
    u32x4 *tiles; unsigned *cnt, *res; unsigned long long *tm;
    CK(hipMalloc(&tiles, (size_t)G * TILE16 * 16)); CK(hipMalloc(&cnt, 4)); CK(hipMalloc(&res, G * 32)); CK(hipMalloc(&tm, 2 * G * 8));
    CK(hipMemset(cnt, 0, 4));
    for (int pass = 0; pass < 3; ++pass) {
        CK(hipMemset(res, 0, G * 32));
        hipLaunchKernelGGL(handover, dim3(2 * G), dim3(256), 0, 0, tiles, cnt, (unsigned)(G * (pass + 1)), 0xA0000000u + pass, res, tm);
        CK(hipDeviceSynchronize());
        std::vector<unsigned> h(G * 8); std::vector<unsigned long long> t(2 * G);
        CK(hipMemcpy(h.data(), res, G * 32, hipMemcpyDeviceToHost)); CK(hipMemcpy(t.data(), tm, 2 * G * 8, hipMemcpyDeviceToHost));
        unsigned long long bp = 0, bs = 0, bss = 0, gave = 0, polls = 0, lastp = 0, firstc = ~0ull, lastc = 0;
        for (int g = 0; g < G; ++g) {
            bp += h[g * 8]; bs += h[g * 8 + 1]; bss += h[g * 8 + 2]; gave += h[g * 8 + 6]; polls += h[g * 8 + 7];
            lastp = t[g] > lastp ? t[g] : lastp; firstc = t[G + g] < firstc ? t[G + g] : firstc; lastc = t[G + g] > lastc ? t[G + g] : lastc;
        }
        printf("pass %d: wrong pieces of %d -- plain loads %llu, sc1 loads %llu, sc0 sc1 loads %llu; consumers that gave up %llu; polls per consumer %.0f; "
               "last producer signal -> first / last consumer release: %lld / %lld ticks\n", pass, G * TILE16, bp, bs, bss, gave, (double)polls / G,
               (long long)(firstc - lastp), (long long)(lastc - lastp));
    }
    return 0;
}
