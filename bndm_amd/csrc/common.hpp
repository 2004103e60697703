// Shared helpers for libbndm_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdint>
#include "../../include/bndm_hip.h"

namespace bndm {

void set_error(const char *fmt, ...);

#define BNDM_CHECK_HIP(expr)                                                              \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess) {                                                           \
            ::bndm::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),      \
                              __FILE__, __LINE__);                                        \
            return (int)_e;                                                               \
        }                                                                                 \
    } while (0)

#define BNDM_REQUIRE(cond, ...)                                                           \
    do {                                                                                  \
        if (!(cond)) {                                                                    \
            ::bndm::set_error(__VA_ARGS__);                                               \
            return BNDM_E_ARG;                                                            \
        }                                                                                 \
    } while (0)

static inline int launch_status(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("launch of %s failed: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// 16-byte asynchronous global -> LDS copy (global_load_lds_dwordx4).  `lds_wave_base` must be
// wave-uniform; lane l's 16 bytes land at lds_wave_base + 16*l.  The global source is per lane.
__device__ __forceinline__ void glds16(const void *gsrc, void *lds_wave_base) {
    __builtin_amdgcn_global_load_lds(
        (const __attribute__((address_space(1))) void *)gsrc,
        (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}

// Buffer descriptor from values that ARE wave-uniform; the readfirstlanes make that provable to the compiler,
// which otherwise wraps every buffer instruction in a waterfall loop.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t uniform_rsrc(const void *p, int bytes) {
    const uint64_t u = (uint64_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void *)(((uint64_t)hi << 32) | lo), 0,
                                             __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
typedef __attribute__((address_space(3))) void *lds_ptr_t;

// Write-through store (sc0 sc1) for a tensor the NEXT kernel reads.  The L2s of the eight XCDs are not coherent, so the
// end of a kernel writes every dirty line back to memory before the dependent kernel's loads get through: a kernel that
// leaves up to 32 MB of freshly written output in the L2s makes its successor wait for that drain behind its first
// loads.  Written through, the output travels during the kernel and the write-back at its end finds nothing to do.
template <typename V> __device__ __forceinline__ void store_wt(V *p, const V &v) {
    static_assert(sizeof(V) == 16 || sizeof(V) == 8 || sizeof(V) == 4, "store_wt: 4, 8 or 16 bytes");
    // (the wait states behind the 16-byte form: a store of more than 64 bits reads its data registers late, and the hazard
    // recogniser does not look inside inline assembly -- without them the next VALU write of those registers corrupts it)
    if constexpr (sizeof(V) == 16)
        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    else if constexpr (sizeof(V) == 8) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}

__device__ __forceinline__ void wait_vmem_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

}  // namespace bndm
