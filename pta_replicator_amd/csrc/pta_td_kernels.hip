// TD ("time-domain") mode: the dense path BASELINE.json's north_star describes - per-pulsar covariance
// assembly, blocked fp64 Cholesky (pta_potrf_batched, pta_orf_kernels.hip) and L . Z.  The reference has no
// such path; the covariance is the one implied by its RN/WN/ECORR synthesis (App. A.1):
//   C[i,j] = sum_c phi[c] F[i,c] F[j,c] + (i==j) sigma2[i] + (epoch_i == epoch_j) ecorr2[i]
// (red_noise.py:98-101,126-128; white_noise.py:105-109,182).
#include "pta_common.h"
#include "pta_mfma.h"

#define TBM 64
#define TBK 16
#define TLD 80

// Lower-triangular tiles only (the Cholesky reads nothing else).  Ft is already K-major, which is the LDS
// operand layout of the MFMA GEMM, so both slabs are loaded with fully coalesced rows; phi is applied on
// the way in.  HBM traffic = the C tiles written once (8 bytes per element of the lower triangle).
__global__ __launch_bounds__(256) void k_td_cov(const double *__restrict__ Ft, int64_t ldf, int N, int K,
                                                const double *__restrict__ phi, const double *__restrict__ sigma2,
                                                const int32_t *__restrict__ epoch_of, const double *__restrict__ ecorr2,
                                                double *__restrict__ C, int64_t ldc) {
  // launched over the nt (nt + 1) / 2 lower-triangular tiles only, row by row: no empty workgroups, XCDs evenly loaded
  const int tix = blockIdx.x;
  int bm = (int)((sqrt(8.0 * (double)tix + 1.0) - 1.0) * 0.5);
  while ((bm + 1) * (bm + 2) / 2 <= tix) ++bm;
  while (bm * (bm + 1) / 2 > tix) --bm;
  const int bn = tix - bm * (bm + 1) / 2;
  __shared__ double As[TBK][TLD];
  __shared__ double Bs[TBK][TLD];
  const int t = threadIdx.x, l = t & 63, w = t >> 6;
  const int wm = w >> 1, wn = w & 1;
  const int m0 = bm * TBM, n0 = bn * TBM;
  pta_f64x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = pta_f64x4{0.0, 0.0, 0.0, 0.0};
  for (int k0 = 0; k0 < K; k0 += TBK) {
    int kr = t >> 4, q = (t & 15) * 4;
    int gk = k0 + kr;
    double ph = (gk < K) ? phi[gk] : 0.0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int gi = m0 + q + j, gj = n0 + q + j;
      As[kr][q + j] = (gk < K && gi < N) ? ph * Ft[(int64_t)gk * ldf + gi] : 0.0;
      Bs[kr][q + j] = (gk < K && gj < N) ? Ft[(int64_t)gk * ldf + gj] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < TBK; kk += 4) {
      double a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = As[kk + (l >> 4)][wm * 32 + i * 16 + (l & 15)];
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = Bs[kk + (l >> 4)][wn * 32 + j * 16 + (l & 15)];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = pta_mfma_f64(a[i], b[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int row = m0 + wm * 32 + i * 16 + pta_mfma_row(l, r);
        int col = n0 + wn * 32 + j * 16 + pta_mfma_col(l);
        if (row < N && col < N && col <= row) {
          double v = acc[i][j][r];
          if (row == col) v = v + sigma2[row];
          if (epoch_of && epoch_of[row] == epoch_of[col]) v = v + ecorr2[row];
          C[(int64_t)row * ldc + col] = v;
        }
      }
}

extern "C" int pta_td_cov_assemble(const double *Ft, int64_t ldf, int N, int K, const double *phi, const double *sigma2,
                                   const int32_t *epoch_of, const double *ecorr2, double *C, int64_t ldc, void *stream) {
  PTA_REQUIRE(Ft && phi && sigma2 && C, PTA_E_ARG, "pta_td_cov_assemble: NULL argument");
  PTA_REQUIRE(!epoch_of || ecorr2, PTA_E_ARG, "pta_td_cov_assemble: ecorr2 missing");
  PTA_REQUIRE(N > 0 && K > 0 && ldf >= N && ldc >= N && pta_cdiv(N, TBM) <= 65535u, PTA_E_ARG, "pta_td_cov_assemble: N=%d K=%d", N, K);
  unsigned nt = pta_cdiv(N, TBM);
  hipLaunchKernelGGL(k_td_cov, dim3(nt * (nt + 1) / 2), dim3(256), 0, pta_stream(stream), Ft, ldf, N, K, phi, sigma2, epoch_of, ecorr2, C,
                     ldc);
  PTA_LAUNCH_CHECK();
  return PTA_OK;
}

int pta_get_gemm_algo();

// out[r, i] (+)= sum_j z[r, j] L[i, j]  =  (Z . L^T)[r, i]; L's strict upper triangle is zero.
extern "C" int pta_td_trmm(const double *L, int64_t ldl, int N, const double *z, int64_t ld_z, int R, double *out, int64_t ld_out,
                           int accumulate, void *stream) {
  PTA_REQUIRE(L && z && out, PTA_E_ARG, "pta_td_trmm: NULL argument");
  PTA_REQUIRE(N > 0 && R > 0 && ldl >= N && ld_z >= N && ld_out >= N, PTA_E_ARG, "pta_td_trmm: N=%d R=%d", N, R);
  return pta_dgemm_launch(1, R, N, N, 1.0, z, ld_z, 1, L, ldl, accumulate ? 1.0 : 0.0, out, ld_out, 0, 1, 0, 0, 0, pta_get_gemm_algo(),
                          pta_stream(stream));
}
