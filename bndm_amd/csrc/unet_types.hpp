// 16-bit element traits shared by the UNet kernel files.
#pragma once
#include "common.hpp"

namespace bndm {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

template <typename T> struct TT;
template <> struct TT<_Float16> {
    using v8 = f16x8;
    using v4 = f16x4;
    using v2 = f16x2;
    static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
};
template <> struct TT<__bf16> {
    using v8 = bf16x8;
    using v4 = bf16x4;
    using v2 = bf16x2;
    static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
};

__device__ __forceinline__ float silu_f(float v) { return v / (1.0f + __expf(-v)); }

}  // namespace bndm
