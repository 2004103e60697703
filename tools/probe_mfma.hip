// Standalone probe (not part of the library): checks the operand/result lane maps assumed by the
// conv kernels for v_mfma_f32_32x32x16_f16 and v_mfma_f32_16x16x32_f16 on the actual device.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k32(const _Float16* A, const _Float16* B, float* D) {  // A[32][16], B[16][32] (k-major rows), D[32][32]
  int l = threadIdx.x;
  half8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = A[(l & 31) * 16 + 8 * (l >> 5) + j]; b[j] = B[(8 * (l >> 5) + j) * 32 + (l & 31)]; }
  f32x16 acc = {0};
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
  for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = acc[r];
}
__global__ void k16(const _Float16* A, const _Float16* B, float* D) {  // A[16][32], B[32][16], D[16][16]
  int l = threadIdx.x;
  half8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = A[(l & 15) * 32 + 8 * (l >> 4) + j]; b[j] = B[(8 * (l >> 4) + j) * 16 + (l & 15)]; }
  f32x4 acc = {0};
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[((l >> 4) * 4 + r) * 16 + (l & 15)] = acc[r];
}
int main() {
  std::vector<_Float16> A(512), B(512); std::vector<float> D(1024), R(1024);
  srand(1);
  for (auto& v : A) v = (_Float16)((rand() % 17 - 8) / 8.0f);
  for (auto& v : B) v = (_Float16)((rand() % 13 - 6) / 4.0f);
  _Float16 *dA, *dB; float* dD;
  hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dD, 4096);
  hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice);
  k32<<<1, 64>>>(dA, dB, dD); hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { float s = 0; for (int k = 0; k < 16; ++k) s += (float)A[i * 16 + k] * (float)B[k * 32 + j]; if (s != D[i * 32 + j]) ++bad; }
  printf("mfma_32x32x16_f16 layout: %s (%d mismatches)\n", bad ? "MISMATCH" : "OK", bad);
  k16<<<1, 64>>>(dA, dB, dD); hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
  int bad2 = 0;
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { float s = 0; for (int k = 0; k < 32; ++k) s += (float)A[i * 32 + k] * (float)B[k * 16 + j]; if (s != D[i * 16 + j]) ++bad2; }
  printf("mfma_16x16x32_f16 layout: %s (%d mismatches)\n", bad2 ? "MISMATCH" : "OK", bad2);
  return (bad || bad2) ? 1 : 0;
}
