// ORF evaluation (spharmORFbasis.py): Hellings-Downs closed form, the anisotropic l <= 8 basis, the clm combination.
// The batched blocked Cholesky that factors the ORF (red_noise.py:235) lives in pta_potrf.hip.
#include "pta_common.h"
#include "pta_orf.h"
#include "pta_mfma.h"

// ---- ORF ---------------------------------------------------------------------------------------
__global__ void k_orf_hd(const double *__restrict__ locs, int P, double *__restrict__ orf) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  int a = blockIdx.y;
  if (b >= P) return;
  orf[(int64_t)a * P + b] = pta_orf_hd(locs[2 * a], locs[2 * b], locs[2 * a + 1], locs[2 * b + 1]);
}

extern "C" int pta_orf_hd(const double *locs, int P, double *orf, void *stream) {
  PTA_REQUIRE(locs && orf, PTA_E_ARG, "pta_orf_hd: NULL argument");
  PTA_REQUIRE(P > 0 && P <= 65535, PTA_E_ARG, "pta_orf_hd: P=%d", P);
  hipLaunchKernelGGL(k_orf_hd, dim3(pta_cdiv(P, 64), P), dim3(64), 0, pta_stream(stream), locs, P, orf);
  PTA_LAUNCH_CHECK();
  return PTA_OK;
}

// one thread per (a <= b, l): all 2l+1 real-form values, mirrored into both triangles
// (correlated_basis, spharmORFbasis.py:385-434)
__global__ void k_orf_basis(const double *__restrict__ locs, const double *__restrict__ zc, int P, double *__restrict__ basis) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  int a = blockIdx.y;
  int l = blockIdx.z;
  if (b >= P || b < a) return;
  double v[2 * PTA_ORF_LMAX + 1];
  pta_orf_pair_l(l, locs[2 * a], locs[2 * b], locs[2 * a + 1], locs[2 * b + 1], zc ? zc + 2 * ((int64_t)a * P + b) : nullptr, v);
  for (int mi = 0; mi <= 2 * l; ++mi) {
    int64_t k = (int64_t)l * l + mi;
    basis[(k * P + a) * P + b] = v[mi];
    basis[(k * P + b) * P + a] = v[mi];
  }
}

extern "C" int pta_orf_basis(const double *locs, const double *zeta_cos, int P, int lmax, double *basis, void *stream) {
  PTA_REQUIRE(locs && basis, PTA_E_ARG, "pta_orf_basis: NULL argument");
  PTA_REQUIRE(P > 0 && P <= 65535, PTA_E_ARG, "pta_orf_basis: P=%d", P);
  PTA_REQUIRE(lmax >= 0 && lmax <= PTA_ORF_LMAX, PTA_E_ARG, "pta_orf_basis: lmax=%d unsupported (0..%d)", lmax, PTA_ORF_LMAX);
  hipLaunchKernelGGL(k_orf_basis, dim3(pta_cdiv(P, 64), P, lmax + 1), dim3(64), 0, pta_stream(stream), locs, zeta_cos, P, basis);
  PTA_LAUNCH_CHECK();
  return PTA_OK;
}

__global__ void k_orf_combine(const double *__restrict__ basis, const double *__restrict__ clm, int nbasis, int64_t PP,
                              double *__restrict__ orf) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= PP) return;
  double s = 0.0;
  for (int k = 0; k < nbasis; ++k) s = s + clm[k] * basis[(int64_t)k * PP + i];  // sum(...) of red_noise.py:225
  orf[i] = s * 2.0;                                                              // :226
}

extern "C" int pta_orf_combine(const double *basis, const double *clm, int nbasis, int P, double *orf, void *stream) {
  PTA_REQUIRE(basis && clm && orf, PTA_E_ARG, "pta_orf_combine: NULL argument");
  PTA_REQUIRE(P > 0 && nbasis > 0, PTA_E_ARG, "pta_orf_combine: P=%d nbasis=%d", P, nbasis);
  int64_t PP = (int64_t)P * P;
  hipLaunchKernelGGL(k_orf_combine, dim3(pta_cdiv(PP, 256)), dim3(256), 0, pta_stream(stream), basis, clm, nbasis, PP, orf);
  PTA_LAUNCH_CHECK();
  return PTA_OK;
}
