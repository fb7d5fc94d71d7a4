// Shared plumbing of libpta_replicator_amd.so: error channel, launch checks, small device helpers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/pta_replicator_amd.h"

void pta_set_error(const char *fmt, ...);

#define PTA_REQUIRE(cond, code, ...) \
  do {                               \
    if (!(cond)) {                   \
      pta_set_error(__VA_ARGS__);    \
      return (code);                 \
    }                                \
  } while (0)

#define PTA_HIP(call)                                                                         \
  do {                                                                                        \
    hipError_t e_ = (call);                                                                   \
    if (e_ != hipSuccess) {                                                                   \
      pta_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
      return PTA_E_HIP;                                                                       \
    }                                                                                         \
  } while (0)

#define PTA_LAUNCH_CHECK()                                                                          \
  do {                                                                                              \
    hipError_t e_ = hipGetLastError();                                                              \
    if (e_ != hipSuccess) {                                                                         \
      pta_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e_), __FILE__, __LINE__); \
      return PTA_E_HIP;                                                                             \
    }                                                                                               \
  } while (0)

static inline unsigned pta_cdiv(long long a, long long b) { return (unsigned)((a + b - 1) / b); }
static inline hipStream_t pta_stream(void *s) { return (hipStream_t)s; }

// internal launcher of the fp64 GEMM (pta_gemm.hip); element (m,k) of A is A[m*lda + k*ska]
int pta_dgemm_launch(int transB, int M, int N, int K, double alpha, const double *A, int64_t lda, int64_t ska,
                     const double *B, int64_t ldb, double beta, double *C, int64_t ldc, int lower_only, int batch,
                     int64_t sA, int64_t sB, int64_t sC, int algo, hipStream_t stream);

int pta_dgemm_tile_n(int M, int N, int K, int algo);
