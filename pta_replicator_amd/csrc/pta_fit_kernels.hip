// Timing-model projection of whole ensembles (the host sink of the reference refits / re-evaluates the timing
// model through PINT once per pulsar and call, simulate.py:40-69; for R realisations held on the device the linearised fit is ONE pass):
//     r  <-  r - M (M^T W M)^{-1} M^T W r          per (realisation, pulsar),
// M the [N_a x m] design matrix of the pulsar's (idealised) timing model, W = diag(1 / sigma^2).  With Q = (M^T W M)^{-1} M^T W
// precomputed on the host ([m x N_a], realisation independent) this is c = Q r (m column reductions over the pulsar's TOAs) followed
// by r -= M c.  One workgroup = one pulsar x PTA_FIT_RB realisations: the operand columns Q[:, i], M[:, i] are read once per TOA and
// reused for all realisations of the group; the m x RB partial sums of a thread are reduced across the wavefront with DPP row shifts /
// broadcasts (__shfl_xor below compiles to them) and across the four waves through LDS.
#include "pta_common.h"

#define PTA_FIT_RB 8     // realisations per workgroup
#define PTA_FIT_MMAX 12  // columns of the design matrix (spin 3, astrometric 9)

__device__ __forceinline__ double pta_wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <int M_>
__global__ __launch_bounds__(256) void k_tm_project(const double *__restrict__ Qt, const double *__restrict__ Mt, int64_t ld,
                                                    const int32_t *__restrict__ psr_off, double *__restrict__ rows, int64_t ld_rows, int R) {
  __shared__ double part[4][M_ * PTA_FIT_RB];
  __shared__ double coef[M_ * PTA_FIT_RB];
  const int a = blockIdx.x, rb = blockIdx.y * PTA_FIT_RB;
  const int i0 = psr_off[a], n = psr_off[a + 1] - i0;
  const int t = threadIdx.x, l = t & 63, w = t >> 6;
  int rq[PTA_FIT_RB];  // realisation rows of this group, clamped (a row past R recomputes row R - 1 and stores the identical value)
#pragma unroll
  for (int q = 0; q < PTA_FIT_RB; ++q) rq[q] = min(rb + q, R - 1);
  double acc[M_][PTA_FIT_RB];
#pragma unroll
  for (int k = 0; k < M_; ++k)
#pragma unroll
    for (int q = 0; q < PTA_FIT_RB; ++q) acc[k][q] = 0.0;
  for (int i = t; i < n; i += 256) {  // pass 1: c = Q r
    double r[PTA_FIT_RB];
#pragma unroll
    for (int q = 0; q < PTA_FIT_RB; ++q) r[q] = rows[(int64_t)rq[q] * ld_rows + i0 + i];
#pragma unroll
    for (int k = 0; k < M_; ++k) {
      const double qk = Qt[(int64_t)k * ld + i0 + i];
#pragma unroll
      for (int q = 0; q < PTA_FIT_RB; ++q) acc[k][q] = fma(qk, r[q], acc[k][q]);
    }
  }
#pragma unroll
  for (int k = 0; k < M_; ++k)
#pragma unroll
    for (int q = 0; q < PTA_FIT_RB; ++q) {
      const double s = pta_wave_sum(acc[k][q]);  // wavefront reduction: no LDS traffic, no barrier
      if (l == 0) part[w][k * PTA_FIT_RB + q] = s;
    }
  __syncthreads();
  if (t < M_ * PTA_FIT_RB) coef[t] = (part[0][t] + part[1][t]) + (part[2][t] + part[3][t]);
  __syncthreads();
  for (int i = t; i < n; i += 256) {  // pass 2: r -= M c  (the rows come back from L2)
    double d[PTA_FIT_RB];
#pragma unroll
    for (int q = 0; q < PTA_FIT_RB; ++q) d[q] = 0.0;
#pragma unroll
    for (int k = 0; k < M_; ++k) {
      const double mk = Mt[(int64_t)k * ld + i0 + i];
#pragma unroll
      for (int q = 0; q < PTA_FIT_RB; ++q) d[q] = fma(mk, coef[k * PTA_FIT_RB + q], d[q]);
    }
#pragma unroll
    for (int q = 0; q < PTA_FIT_RB; ++q) {
      const int64_t o = (int64_t)rq[q] * ld_rows + i0 + i;
      if (q == 0 || rq[q] != rq[q - 1]) rows[o] = rows[o] - d[q];  // a clamped duplicate row must not be subtracted twice
    }
  }
}

extern "C" int pta_tm_project(const double *Qt, const double *Mt, int64_t ld, int m, const int32_t *psr_off, int P, double *rows,
                              int64_t ld_rows, int R, void *stream) {
  PTA_REQUIRE(Qt && Mt && psr_off && rows, PTA_E_ARG, "pta_tm_project: NULL argument");
  PTA_REQUIRE(m >= 1 && m <= PTA_FIT_MMAX && P > 0 && P <= 65535 && R > 0, PTA_E_ARG, "pta_tm_project: m=%d (1..%d) P=%d R=%d", m, PTA_FIT_MMAX, P,
              R);
  const unsigned ng = pta_cdiv(R, PTA_FIT_RB);
  PTA_REQUIRE(ng <= 65535u, PTA_E_ARG, "pta_tm_project: R=%d exceeds one launch (<= %d)", R, 65535 * PTA_FIT_RB);
  dim3 g(P, ng), b(256);
  hipStream_t s = pta_stream(stream);
#define PTA_TM_CASE(MM) \
  case MM: hipLaunchKernelGGL(k_tm_project<MM>, g, b, 0, s, Qt, Mt, ld, psr_off, rows, ld_rows, R); break;
  switch (m) {
    PTA_TM_CASE(1) PTA_TM_CASE(2) PTA_TM_CASE(3) PTA_TM_CASE(4) PTA_TM_CASE(5) PTA_TM_CASE(6) PTA_TM_CASE(7) PTA_TM_CASE(8) PTA_TM_CASE(9)
    PTA_TM_CASE(10) PTA_TM_CASE(11) PTA_TM_CASE(12)
  }
#undef PTA_TM_CASE
  PTA_LAUNCH_CHECK();
  return PTA_OK;
}
