// Catalogue of continuous-wave sources: sum over N_cw SMBHBs of the add_cgw waveform for one pulsar.
// Replaces add_catalog_of_cws and its two numba kernels loop_over_CWs_parallel / loop_over_CWs
// (deterministic.py:188-318, :321-440, :443-561): the reference's own "make it fast" code path.
//   (the per-source scalars of deterministic.py:331-383 - unit conversion, antenna patterns, chirp factors, pd (1 - cos mu) - come
//    from the host, pta_replicator_amd.deterministic.cw_source_params: one ulp of cos mu is up to 4e-11 rad of pulsar-term phase)
//   k_cw_catalog       : one thread per TOA, sources walked through the scalar unit; NaN contributions (binaries that
//                        already merged) are dropped as in :435,:556; per-chunk partial sums
//   k_cw_reduce        : fixed-order sum of the partials (deterministic, unlike atomics)
#include "pta_common.h"

#define CW_NPAR PTA_CW_NPAR

struct pta_cw_opts {
  double tref;
  int psr_term, mode;  // mode 0 evolve, 1 phase_approx, 2 monochromatic
};

#define CW_TILE 256
__global__ __launch_bounds__(CW_TILE) void k_cw_catalog(const double *__restrict__ mjd, int N, const double *__restrict__ par, int ncw,
                                                        int chunk, pta_cw_opts o, double *__restrict__ partial) {
  const int i = blockIdx.x * CW_TILE + threadIdx.x;
  const int c0 = blockIdx.y * chunk, c1 = min(ncw, c0 + chunk);
  const double toas = (i < N) ? mjd[i] * 86400.0 - o.tref : 0.0;
  double acc = 0.0;
  for (int c = c0; c < c1; ++c) {
    const double *__restrict__ p = par + (int64_t)c * CW_NPAR;  // uniform across the workgroup: scalar loads
    const double w0 = p[0], phase0 = p[1], w053 = p[2], fac1 = p[3], fac2 = p[4], fac3 = p[5];
    const double tp = toas - p[12];
    double omega, omega_p, phase, phase_p;
    if (o.mode == 0) {
      omega = w0 * pow(1.0 - fac1 * toas, -3.0 / 8.0);
      omega_p = w0 * pow(1.0 - fac1 * tp, -3.0 / 8.0);
      phase = phase0 + fac2 * (w053 - pow(omega, -5.0 / 3.0));
      phase_p = phase0 + fac2 * (w053 - pow(omega_p, -5.0 / 3.0));
    } else if (o.mode == 1) {
      omega = w0;
      omega_p = p[13];
      phase = phase0 + omega * toas;
      phase_p = p[14] + omega_p * toas;
    } else {
      omega = w0;
      omega_p = w0;
      phase = phase0 + omega * toas;
      phase_p = phase0 + omega * tp;
    }
    double s, cs;
    sincos(2.0 * phase, &s, &cs);
    const double At = s * p[6], Bt = cs * p[7];
    sincos(2.0 * phase_p, &s, &cs);
    const double At_p = s * p[6], Bt_p = cs * p[7];
    const double alpha = fac3 / pow(omega, 1.0 / 3.0), alpha_p = fac3 / pow(omega_p, 1.0 / 3.0);
    const double c2p = p[8], s2p = p[9];
    const double rplus = alpha * (At * c2p + Bt * s2p), rcross = alpha * (-At * s2p + Bt * c2p);
    const double rplus_p = alpha_p * (At_p * c2p + Bt_p * s2p), rcross_p = alpha_p * (-At_p * s2p + Bt_p * c2p);
    const double rrr = o.psr_term ? p[10] * (rplus_p - rplus) + p[11] * (rcross_p - rcross) : -p[10] * rplus - p[11] * rcross;
    acc += (rrr != rrr) ? 0.0 : rrr;  // np.where(np.isnan(rrr), 0.0, rrr)
  }
  if (i < N) partial[(int64_t)blockIdx.y * N + i] = acc;
}

__global__ void k_cw_reduce(const double *__restrict__ partial, int nchunk, int N, double *__restrict__ out, int accumulate) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  double s = 0.0;
  for (int c = 0; c < nchunk; ++c) s += partial[(int64_t)c * N + i];
  out[i] = accumulate ? out[i] + s : s;
}

extern "C" int pta_cw_catalog_workspace(int N, int ncw, int64_t *partial_doubles, int *nchunk_out) {
  PTA_REQUIRE(N > 0 && ncw > 0 && partial_doubles && nchunk_out, PTA_E_ARG, "pta_cw_catalog_workspace: bad argument");
  int tiles = (N + CW_TILE - 1) / CW_TILE;
  int want = (4096 + tiles - 1) / tiles;  // ~16 workgroups per CU in flight
  int nchunk = want < 1 ? 1 : (want > ncw ? ncw : want);
  if (nchunk > 65535) nchunk = 65535;
  *nchunk_out = nchunk;
  *partial_doubles = (int64_t)nchunk * N;
  return PTA_OK;
}

extern "C" int pta_cw_catalog(const double *mjd, int N, const double *par, int ncw, int psr_term, int mode, double tref,
                              double *partial_ws, double *out, int accumulate, void *stream) {
  PTA_REQUIRE(mjd && par && partial_ws && out, PTA_E_ARG, "pta_cw_catalog: NULL argument");
  PTA_REQUIRE(N > 0 && ncw > 0 && mode >= 0 && mode <= 2, PTA_E_ARG, "pta_cw_catalog: N=%d ncw=%d mode=%d", N, ncw, mode);
  pta_cw_opts o;
  o.tref = tref; o.psr_term = psr_term; o.mode = mode;
  int64_t b;
  int nchunk;
  pta_cw_catalog_workspace(N, ncw, &b, &nchunk);
  int chunk = (ncw + nchunk - 1) / nchunk;
  nchunk = (ncw + chunk - 1) / chunk;
  hipStream_t s = pta_stream(stream);
  hipLaunchKernelGGL(k_cw_catalog, dim3(pta_cdiv(N, CW_TILE), nchunk), dim3(CW_TILE), 0, s, mjd, N, par, ncw, chunk, o, partial_ws);
  PTA_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_cw_reduce, dim3(pta_cdiv(N, 256)), dim3(256), 0, s, partial_ws, nchunk, N, out, accumulate);
  PTA_LAUNCH_CHECK();
  return PTA_OK;
}
