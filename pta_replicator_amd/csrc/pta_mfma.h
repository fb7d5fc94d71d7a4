// v_mfma_f64_16x16x4_f64 wrapper and its lane <-> element maps (device only).
#pragma once
#include <hip/hip_runtime.h>

typedef double pta_f64x4 __attribute__((ext_vector_type(4)));

// D = A(16x4) * B(4x16) + C; lane l supplies A[l & 15][l >> 4] and B[l >> 4][l & 15]
__device__ __forceinline__ pta_f64x4 pta_mfma_f64(double a, double b, pta_f64x4 c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}
// accumulator register r of lane l is C[pta_mfma_row(l, r)][pta_mfma_col(l)]  (f64 differs from every other dtype)
__device__ __forceinline__ int pta_mfma_row(int l, int r) { return (l >> 4) + 4 * r; }
__device__ __forceinline__ int pta_mfma_col(int l) { return l & 15; }
