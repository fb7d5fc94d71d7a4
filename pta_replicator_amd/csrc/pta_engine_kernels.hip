// Throughput mode: one fused pass writes R whole-array realisations,
//     out[r, i] = RN + GWB + EFAC/EQUAD + ECORR + deterministic,
// with every Gaussian deviate produced in registers (pta_rng.h).  This is the kernel the
// realisations/sec metric is quoted on; its algorithmic traffic is the 8 bytes per (realisation, TOA)
// it writes - everything it reads is realisation independent (Ft, noise vectors, epoch
// map) or tiny per realisation (60 RN coefficients and a 600-sample GWB row per pulsar).
#include "pta_common.h"
#include "pta_rng.h"

// coef[(r*P + a)*K + c] = amp[a*K + c] * z,  z = deviate c of stream (RN, a)   (red_noise.py:126-127)
__global__ void k_engine_rn_coef(uint64_t seed, uint64_t r0, int R, int P, int K, const double *__restrict__ amp,
                                 double *__restrict__ coef) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;  // (r, a, pair)
  int hp = K / 2;
  int total = R * P * hp;
  if (idx >= total) return;
  int p = idx % hp, ra = idx / hp;
  int a = ra % P, r = ra / P;
  double z0, z1;
  pta_normal_pair(seed, r0 + (uint64_t)r, pta_stream_id(PTA_STREAM_RN, (uint32_t)a), (uint32_t)p, z0, z1);
  int64_t o = (int64_t)ra * K + 2 * p;
  coef[o] = amp[(int64_t)a * K + 2 * p] * z0;
  coef[o + 1] = amp[(int64_t)a * K + 2 * p + 1] * z1;
}

extern "C" int pta_engine_rn_coef(uint64_t seed, uint64_t r0, int R, int P, int K, const double *amp, double *coef,
                                  void *stream) {
  PTA_REQUIRE(amp && coef, PTA_E_ARG, "pta_engine_rn_coef: NULL argument");
  PTA_REQUIRE(R > 0 && P > 0 && K > 0 && (K % 2) == 0, PTA_E_ARG, "pta_engine_rn_coef: R=%d P=%d K=%d (K must be even)", R, P, K);
  int64_t total = (int64_t)R * P * (K / 2);
  PTA_REQUIRE(total < (1LL << 31), PTA_E_ARG, "pta_engine_rn_coef: problem too large");
  hipLaunchKernelGGL(k_engine_rn_coef, dim3(pta_cdiv(total, 256)), dim3(256), 0, pta_stream(stream), seed, r0, R, P, K, amp, coef);
  PTA_LAUNCH_CHECK();
  return PTA_OK;
}

#define ENG_RB 4  // realisations per thread: amortises the realisation-independent loads (Ft row, noise vectors)

__global__ __launch_bounds__(256) void k_engine_synth(pta_engine_plan pl, uint64_t seed, uint64_t r0, int R,
                                                      double *__restrict__ out, int64_t ld_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int rb = blockIdx.y * ENG_RB;
  if (i >= pl.n_toa) return;
  const int a = pl.psr_of_toa[i];
  const int P = pl.n_psr;
  double v[ENG_RB];
  const double det = pl.det ? pl.det[i] : 0.0;
#pragma unroll
  for (int q = 0; q < ENG_RB; ++q) v[q] = 0.0;

  if (pl.rn_k > 0) {  // red noise: dt = F @ y (red_noise.py:128)
    const int K = pl.rn_k;
    for (int c = 0; c < K; ++c) {
      double fv = pl.Ft[(int64_t)c * pl.ldf + i];
#pragma unroll
      for (int q = 0; q < ENG_RB; ++q) {
        int r = min(rb + q, R - 1);
        v[q] = fma(fv, pl.rn_coef[((int64_t)r * P + a) * K + c], v[q]);
      }
    }
  }
  if (pl.gw_npts > 0) {  // GWB: interpolate the mixed grid series (red_noise.py:286-287)
    const int j = pl.gw_jlo[i];
    const double x = pl.toa_s[i];
    const double x0 = pl.gw_ut[j], dx = pl.gw_ut[j + 1] - x0;
#pragma unroll
    for (int q = 0; q < ENG_RB; ++q) {
      int r = min(rb + q, R - 1);
      const double *g = pl.gw_G + ((int64_t)r * P + a) * pl.gw_npts;
      double slope = (g[j + 1] - g[j]) / dx;
      v[q] = v[q] + (slope * (x - x0) + g[j]);
    }
  }
  if (pl.wn_a) {  // EFAC/EQUAD: (efac sigma) z1 + (efac equad | equad) z2 (white_noise.py:105-109)
    const double wa = pl.wn_a[i], wb = pl.wn_b[i];
    const uint32_t strm = pta_stream_id(PTA_STREAM_WN, (uint32_t)a);
    const uint32_t pair = (uint32_t)pl.idx_in_psr[i];
#pragma unroll
    for (int q = 0; q < ENG_RB; ++q) {
      double z1, z2;
      pta_normal_pair(seed, r0 + (uint64_t)(rb + q), strm, pair, z1, z2);
      v[q] = v[q] + (wa * z1 + wb * z2);
    }
  }
  if (pl.ecorr_toa) {  // ECORR: ecorr[e(i)] z[e(i)] (white_noise.py:182)
    const double ec = pl.ecorr_toa[i];
    if (ec != 0.0) {
      const uint32_t strm = pta_stream_id(PTA_STREAM_ECORR, (uint32_t)a);
      const uint32_t e = (uint32_t)pl.epoch_of[i];
#pragma unroll
      for (int q = 0; q < ENG_RB; ++q) v[q] = v[q] + ec * pta_normal_single(seed, r0 + (uint64_t)(rb + q), strm, e);
    }
  }
#pragma unroll
  for (int q = 0; q < ENG_RB; ++q)
    if (rb + q < R) out[(int64_t)(rb + q) * ld_out + i] = v[q] + det;
}

extern "C" int pta_engine_synth(const pta_engine_plan *plan_host, uint64_t seed, uint64_t r0, int R, double *out, int64_t ld_out,
                                void *stream) {
  PTA_REQUIRE(plan_host && out, PTA_E_ARG, "pta_engine_synth: NULL argument");
  const pta_engine_plan &p = *plan_host;
  PTA_REQUIRE(p.n_toa > 0 && p.n_psr > 0 && R > 0 && ld_out >= p.n_toa, PTA_E_ARG, "pta_engine_synth: n_toa=%d n_psr=%d R=%d", p.n_toa,
              p.n_psr, R);
  PTA_REQUIRE(p.psr_of_toa && p.idx_in_psr, PTA_E_ARG, "pta_engine_synth: psr_of_toa / idx_in_psr missing");
  PTA_REQUIRE(p.rn_k == 0 || (p.Ft && p.rn_coef && p.ldf >= p.n_toa), PTA_E_ARG, "pta_engine_synth: red-noise inputs missing");
  PTA_REQUIRE(p.gw_npts == 0 || (p.gw_G && p.gw_ut && p.gw_jlo && p.toa_s && p.gw_npts >= 2), PTA_E_ARG,
              "pta_engine_synth: GWB inputs missing");
  PTA_REQUIRE(!p.wn_a || p.wn_b, PTA_E_ARG, "pta_engine_synth: wn_b missing");
  PTA_REQUIRE(!p.ecorr_toa || p.epoch_of, PTA_E_ARG, "pta_engine_synth: epoch_of missing");
  PTA_REQUIRE(pta_cdiv(R, ENG_RB) <= 65535u, PTA_E_ARG, "pta_engine_synth: R=%d too large for one launch", R);
  hipLaunchKernelGGL(k_engine_synth, dim3(pta_cdiv(p.n_toa, 256), pta_cdiv(R, ENG_RB)), dim3(256), 0, pta_stream(stream), p, seed, r0, R,
                     out, ld_out);
  PTA_LAUNCH_CHECK();
  return PTA_OK;
}
