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

// ---- ragged batches (pta_potrf_ragged): matrices of different orders in END-ALIGNED virtual coordinates -------------------------
// Every matrix of the batch is embedded in a virtual matrix of order E whose bottom-right corner it shares: virtual index i <-> real
// index i - front[b], front[b] = E - n[b].  In these coordinates every panel boundary, trailing size and tile grid of the blocked
// factorisation is THE SAME for all matrices that are active at a step - what differs per matrix is where its storage lies (off,
// ld) and where it begins (front): rows / columns / K indices below `front` do not exist and are masked.
//   off[b]   offset (doubles) of VIRTUAL element (0, 0) of matrix b from the batch's base pointer (may be negative; only indices
//            >= front are ever dereferenced);  ld[b] its leading dimension;  front[b] its first real row / column.
struct pta_rag {
  const int64_t *off;
  const int64_t *ld;
  const int64_t *front;
  int epi;  // host side: 1 = the tile products' EPI = 1 form (PTA_POTRF_EPI1, A/B)
};
// C[r0 + m, c0 + n] = alpha * sum_k A[r0 + m, k0 + k] * Bop[n, k] + beta * C[...] for every matrix b < batch of a ragged batch (m < M,
// n < N, k < K in virtual coordinates, masked below front[b]); Bop[n, k] = the matrix's own element [c0 + n, k0 + k] when Bws is NULL,
// else Bws[b * sB + n * ldb + k] (a uniformly strided workspace operand whose k index follows the same virtual columns).
// lower_only: only c0 + n <= r0 + m (the caller passes r0 == c0 for square updates).  Needs algo >= 1 kernels; all origins even.
int pta_dgemm_launch_rag(int M, int N, int K, double alpha, double *Abase, int r0, int c0, int k0, const double *Bws, int64_t ldb, int64_t sB,
                         double beta, int lower_only, int batch, pta_rag rg, hipStream_t stream);

// ---- assembly operands of the fused left-looking factorisation (pta_td_assemble_potrf; csrc/pta_td_fused.hip) ---------------------------
// Device pointers over the concatenated TOAs of a UNIFORM batch (matrix z's TOAs are [z toa_stride, (z + 1) toa_stride)):
struct pta_fuse {
  const double *Fr;       // [sum N, 64] row-major rows of the Fourier design matrix (red_noise.py:98-101), columns >= kf zero
  const double *Gr;       // [sum N, 64] = -phi_k Fr[i, k] (phi = the prior variances, red_noise.py:126)
  const double *sigma2;   // [sum N] white-noise variances (white_noise.py:105-109)
  const int32_t *epoch;   // [sum N] ECORR epoch of every TOA (white_noise.py:7-44), or NULL
  const double *ecorr2;   // [sum N] ecorr^2 of the TOA's epoch (white_noise.py:182), or NULL
  int64_t toa_stride;     // TOAs per matrix
  int32_t kf;             // red-noise columns in use (0 = none: the design-matrix phase is skipped)
};
int pta_td_fused_launch(int M, int N, int K, double *L, int64_t ld, int64_t sL, int r0, int batch, const pta_fuse &fz, hipStream_t stream);

// blocked substitution of a whole panel in one launch, a workgroup per 128-row tile walking the panel's blocks (csrc/pta_solve_rows.hip)
int pta_ws_solve_rows_launch(double *X, int64_t lda, int64_t sA, int B, int rows, int nb, int f128, const double *W, int64_t ldw, int64_t sW,
                             hipStream_t stream);
