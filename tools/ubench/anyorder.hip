// Can two launches on ONE stream overlap on gfx950 when the second is launched with hipExtAnyOrderLaunch (no barrier bit)?
// (hip_ext.h says the flag is "not supported on AMD GFX9xx boards" for hipExtModuleLaunchKernel; nothing for hipExtLaunchKernel.)
// A "long" kernel spins ~200 us on 128 workgroups (half the CUs, one workgroup each); a "short" kernel stamps s_memtime.  Cases:
//   1. long, short on one stream, plain launches           -> short must start after long's end
//   2. long, short on one stream, short with AnyOrderLaunch -> overlap iff the flag works here
//   3. long, short on two streams                           -> overlap (reference)
// Prints for each case: start of short relative to start of long, and long's duration, in us (s_memtime at ~100 MHz is NOT assumed:
// the tick rate is calibrated against hipEvent time of the long kernel).
//   hipcc -O3 --offload-arch=gfx950 anyorder.hip -o anyorder.bin
// Why it matters: with launches that may overlap, a layer's workgroups could wait on per-sample counters of the layer before (all of
// whose workgroups are dispatched first, so no deadlock) -- kernel boundaries and phase-locked rounds would go (last paragraph).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void longk(unsigned long long *t, unsigned long long spin) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < spin) __builtin_amdgcn_s_sleep(8);
    if (blockIdx.x == 0 && threadIdx.x == 0) { t[0] = t0; t[1] = __builtin_amdgcn_s_memtime(); }
}
__global__ void shortk(unsigned long long *t) {
    if (blockIdx.x == 0 && threadIdx.x == 0) t[2] = __builtin_amdgcn_s_memtime();
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
    unsigned long long *d, h[3];
    CK(hipMalloc(&d, 24));
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // calibrate ticks per us
    unsigned long long spin = 20000;
    CK(hipEventRecord(e0, s1));
    hipLaunchKernelGGL(longk, dim3(128), dim3(64), 0, s1, d, spin);
    CK(hipEventRecord(e1, s1));
    CK(hipStreamSynchronize(s1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(h, d, 24, hipMemcpyDeviceToHost));
    const double tick_per_us = (double)(h[1] - h[0]) / (ms * 1e3);
    printf("s_memtime: %.1f ticks per us (kernel of %llu ticks took %.1f us incl. launch)\n", tick_per_us, spin, ms * 1e3);
    spin = (unsigned long long)(200.0 * tick_per_us);
    for (int c = 1; c <= 3; ++c) {
        CK(hipMemset(d, 0, 24));
        CK(hipDeviceSynchronize());
        hipLaunchKernelGGL(longk, dim3(128), dim3(64), 0, s1, d, spin);
        if (c == 2) {
            void *args[] = {&d};
            hipError_t e = hipExtLaunchKernel((const void *)shortk, dim3(1), dim3(64), args, 0, s1, nullptr, nullptr, hipExtAnyOrderLaunch);
            if (e != hipSuccess) { printf("case 2: hipExtLaunchKernel(AnyOrderLaunch): %s\n", hipGetErrorString(e)); (void)hipGetLastError(); continue; }
        } else {
            hipLaunchKernelGGL(shortk, dim3(1), dim3(64), 0, c == 3 ? s2 : s1, d);
        }
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h, d, 24, hipMemcpyDeviceToHost));
        printf("case %d (%s): short starts %+8.1f us after long's start; long ran %.1f us -> %s\n", c,
               c == 1 ? "one stream, plain" : c == 2 ? "one stream, AnyOrderLaunch" : "two streams",
               ((double)h[2] - (double)h[0]) / tick_per_us, (double)(h[1] - h[0]) / tick_per_us,
               h[2] < h[1] ? "OVERLAP" : "serial");
    }
    return 0;
}
