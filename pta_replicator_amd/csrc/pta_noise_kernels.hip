// Per-signal kernels of the injection path: red-noise Fourier synthesis, EFAC/EQUAD, ECORR, CGW.
// These are the R-batched building blocks behind add_red_noise / add_measurement_noise /
// add_jitter / add_cgw; the fused throughput kernel lives in pta_engine_kernels.hip.
#include "pta_common.h"

#define PTA_TWO_PI 6.283185307179586  // float64(2*np.pi)

// ---- red noise -------------------------------------------------------------------------------
// red_noise.py:86-101.  One thread per (TOA, mode); TOA is the fastest index so both stores are
// coalesced.  The phase argument is formed with the reference's association so that it is the
// same float64 number NumPy builds: ((2 pi)*(t - t_ref))*f + phase   (t ~ 4.6e9 s: one ulp of the
// argument is ~2e-13 rad, so the association matters at the 1e-13 level, not at 1e-10).
__global__ void k_rn_basis(const double *__restrict__ t, int N, double t_ref, const double *__restrict__ f,
                           const double *__restrict__ phase, int nmodes, int cos_first, double *__restrict__ Ft,
                           int64_t ldf) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  int k = blockIdx.y;
  if (i >= N) return;
  double ph = phase ? phase[k] : 0.0;
  double arg = (PTA_TWO_PI * (t[i] - t_ref)) * f[k] + ph;
  double s, c;
  sincos(arg, &s, &c);
  Ft[(int64_t)(2 * k) * ldf + i] = cos_first ? c : s;
  Ft[(int64_t)(2 * k + 1) * ldf + i] = cos_first ? s : c;
}

extern "C" int pta_rn_basis(const double *t, int N, double t_ref, const double *f, const double *phase, int nmodes,
                            int cos_first, double *Ft, int64_t ldf, void *stream) {
  PTA_REQUIRE(t && f && Ft, PTA_E_ARG, "pta_rn_basis: NULL argument");
  PTA_REQUIRE(N > 0 && nmodes > 0 && nmodes <= 65535 && ldf >= N, PTA_E_ARG, "pta_rn_basis: N=%d nmodes=%d ldf=%lld", N,
              nmodes, (long long)ldf);
  hipLaunchKernelGGL(k_rn_basis, dim3(pta_cdiv(N, 256), nmodes), dim3(256), 0, pta_stream(stream), t, N, t_ref, f, phase,
                     nmodes, cos_first, Ft, ldf);
  PTA_LAUNCH_CHECK();
  return PTA_OK;
}

// red_noise.py:128 (dt = F @ y), batched over realisations.  One thread owns one TOA and RB
// realisations; the coefficient reads are wave-uniform (scalar loads), the F reads coalesced.
#define PTA_RN_RB 8
__global__ void k_rn_synth(const double *__restrict__ Ft, int64_t ldf, int N, int K, const double *__restrict__ coef,
                           int64_t ld_coef, int R, double *__restrict__ out, int64_t ld_out, int accumulate) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  int r0 = blockIdx.y * PTA_RN_RB;
  if (i >= N) return;
  double acc[PTA_RN_RB];
#pragma unroll
  for (int q = 0; q < PTA_RN_RB; ++q) acc[q] = 0.0;
  for (int c = 0; c < K; ++c) {
    double fv = Ft[(int64_t)c * ldf + i];
#pragma unroll
    for (int q = 0; q < PTA_RN_RB; ++q) {
      int r = min(r0 + q, R - 1);
      acc[q] = fma(fv, coef[(int64_t)r * ld_coef + c], acc[q]);
    }
  }
#pragma unroll
  for (int q = 0; q < PTA_RN_RB; ++q) {
    int r = r0 + q;
    if (r < R) {
      int64_t o = (int64_t)r * ld_out + i;
      out[o] = accumulate ? out[o] + acc[q] : acc[q];
    }
  }
}

extern "C" int pta_rn_synth(const double *Ft, int64_t ldf, int N, int K, const double *coef, int64_t ld_coef, int R,
                            double *out, int64_t ld_out, int accumulate, void *stream) {
  PTA_REQUIRE(Ft && coef && out, PTA_E_ARG, "pta_rn_synth: NULL argument");
  PTA_REQUIRE(N > 0 && K > 0 && R > 0 && ldf >= N && ld_coef >= K && ld_out >= N, PTA_E_ARG,
              "pta_rn_synth: N=%d K=%d R=%d ldf=%lld ld_coef=%lld ld_out=%lld", N, K, R, (long long)ldf, (long long)ld_coef,
              (long long)ld_out);
  PTA_REQUIRE(pta_cdiv(R, PTA_RN_RB) <= 65535u, PTA_E_ARG, "pta_rn_synth: R=%d too large for one launch", R);
  hipLaunchKernelGGL(k_rn_synth, dim3(pta_cdiv(N, 256), pta_cdiv(R, PTA_RN_RB)), dim3(256), 0, pta_stream(stream), Ft, ldf,
                     N, K, coef, ld_coef, R, out, ld_out, accumulate);
  PTA_LAUNCH_CHECK();
  return PTA_OK;
}

// ---- EFAC / EQUAD (white_noise.py:105-109) ------------------------------------------------------
__global__ void k_wn(const double *__restrict__ sigma, const double *__restrict__ efac, const double *__restrict__ equad,
                     int N, int tnequad, const double *__restrict__ z1, const double *__restrict__ z2, int64_t ld_z,
                     double *__restrict__ out, int64_t ld_out, int accumulate) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  int r = blockIdx.y;
  if (i >= N) return;
  double a = efac[i] * sigma[i];
  double b = tnequad ? equad[i] : efac[i] * equad[i];
  double v = a * z1[(int64_t)r * ld_z + i];
  v = v + b * z2[(int64_t)r * ld_z + i];
  int64_t o = (int64_t)r * ld_out + i;
  out[o] = accumulate ? out[o] + v : v;
}

extern "C" int pta_wn(const double *sigma, const double *efac, const double *equad, int N, int tnequad, const double *z1,
                      const double *z2, int64_t ld_z, int R, double *out, int64_t ld_out, int accumulate, void *stream) {
  PTA_REQUIRE(sigma && efac && equad && z1 && z2 && out, PTA_E_ARG, "pta_wn: NULL argument");
  PTA_REQUIRE(N > 0 && R > 0 && R <= 65535 && ld_z >= N && ld_out >= N, PTA_E_ARG, "pta_wn: N=%d R=%d", N, R);
  hipLaunchKernelGGL(k_wn, dim3(pta_cdiv(N, 256), R), dim3(256), 0, pta_stream(stream), sigma, efac, equad, N, tnequad, z1, z2,
                     ld_z, out, ld_out, accumulate);
  PTA_LAUNCH_CHECK();
  return PTA_OK;
}

// ---- ECORR (white_noise.py:182): a gather through the epoch map, not an N x E matvec ------------
__global__ void k_ecorr(const int32_t *__restrict__ epoch_of, const double *__restrict__ ecorr_epoch, int N,
                        const double *__restrict__ z, int64_t ld_z, double *__restrict__ out, int64_t ld_out,
                        int accumulate) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  int r = blockIdx.y;
  if (i >= N) return;
  int e = epoch_of[i];
  double v = ecorr_epoch[e] * z[(int64_t)r * ld_z + e];
  int64_t o = (int64_t)r * ld_out + i;
  out[o] = accumulate ? out[o] + v : v;
}

extern "C" int pta_ecorr(const int32_t *epoch_of, const double *ecorr_epoch, int N, int E, const double *z, int64_t ld_z,
                         int R, double *out, int64_t ld_out, int accumulate, void *stream) {
  PTA_REQUIRE(epoch_of && ecorr_epoch && z && out, PTA_E_ARG, "pta_ecorr: NULL argument");
  PTA_REQUIRE(N > 0 && E > 0 && R > 0 && R <= 65535 && ld_z >= E && ld_out >= N, PTA_E_ARG, "pta_ecorr: N=%d E=%d R=%d", N, E, R);
  hipLaunchKernelGGL(k_ecorr, dim3(pta_cdiv(N, 256), R), dim3(256), 0, pta_stream(stream), epoch_of, ecorr_epoch, N, z, ld_z,
                     out, ld_out, accumulate);
  PTA_LAUNCH_CHECK();
  return PTA_OK;
}

// ---- CGW (deterministic.py:97-163) --------------------------------------------------------------
struct pta_cgw_par {
  double v[PTA_CGW_NPAR];
};

__global__ void k_cgw(const double *__restrict__ mjd, int N, pta_cgw_par P, double *__restrict__ out, int accumulate) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const double tref = P.v[0], w0 = P.v[1], phase0 = P.v[2], w053 = P.v[3], fac1 = P.v[4], fac2 = P.v[5], fac3 = P.v[6];
  const double incfac1 = P.v[7], incfac2 = P.v[8], c2p = P.v[9], s2p = P.v[10], fplus = P.v[11], fcross = P.v[12];
  const double pdterm = P.v[13];
  const int mode = (int)P.v[14];
  const int psr_term = (int)P.v[15];
  double toas = mjd[i] * 86400.0 - tref;  // :98
  double tp = toas - pdterm;               // :108
  double omega, omega_p, phase, phase_p;
  if (mode == 0) {  // evolve :111-119
    omega = w0 * pow(1.0 - fac1 * toas, -3.0 / 8.0);
    omega_p = w0 * pow(1.0 - fac1 * tp, -3.0 / 8.0);
    phase = phase0 + fac2 * (w053 - pow(omega, -5.0 / 3.0));
    phase_p = phase0 + fac2 * (w053 - pow(omega_p, -5.0 / 3.0));
  } else if (mode == 1) {  // phase_approx :122-130
    omega = w0;
    omega_p = P.v[16];
    phase = phase0 + omega * toas;
    phase_p = P.v[17] + omega_p * toas;
  } else {  // monochromatic :133-141
    omega = w0;
    omega_p = w0;
    phase = phase0 + omega * toas;
    phase_p = phase0 + omega * tp;
  }
  double s, c;
  sincos(2.0 * phase, &s, &c);
  double At = s * incfac1, Bt = c * incfac2;
  sincos(2.0 * phase_p, &s, &c);
  double At_p = s * incfac1, Bt_p = c * incfac2;
  double alpha = fac3 / pow(omega, 1.0 / 3.0);
  double alpha_p = fac3 / pow(omega_p, 1.0 / 3.0);
  double rplus = alpha * (At * c2p + Bt * s2p);
  double rcross = alpha * (-At * s2p + Bt * c2p);
  double rplus_p = alpha_p * (At_p * c2p + Bt_p * s2p);
  double rcross_p = alpha_p * (-At_p * s2p + Bt_p * c2p);
  double res = psr_term ? fplus * (rplus_p - rplus) + fcross * (rcross_p - rcross) : -fplus * rplus - fcross * rcross;
  out[i] = accumulate ? out[i] + res : res;
}

extern "C" int pta_cgw(const double *mjd, int N, const double *par_host, double *out, int accumulate, void *stream) {
  PTA_REQUIRE(mjd && par_host && out, PTA_E_ARG, "pta_cgw: NULL argument");
  PTA_REQUIRE(N > 0, PTA_E_ARG, "pta_cgw: N=%d", N);
  pta_cgw_par P;
  for (int k = 0; k < PTA_CGW_NPAR; ++k) P.v[k] = par_host[k];
  hipLaunchKernelGGL(k_cgw, dim3(pta_cdiv(N, 256)), dim3(256), 0, pta_stream(stream), mjd, N, P, out, accumulate);
  PTA_LAUNCH_CHECK();
  return PTA_OK;
}
