// TEST HARNESS (not product code): compiles the __host__ __device__ math headers of
// pta_replicator_amd/csrc with g++ so that the exact device formulas can be checked against the
// oracle / golden vectors on a machine without a GPU.  Loaded by tests/test_hostcheck.py via ctypes.
#include <stdint.h>
#include "../../pta_replicator_amd/csrc/pta_rng.h"
#include "../../pta_replicator_amd/csrc/pta_orf.h"
#include "../../pta_replicator_amd/csrc/pta_fft.h"

// FFT passes with the 512 "threads" of a workgroup emulated.  The kernel only puts a workgroup barrier between the
// column pair (s = 512, 64) and the block pair (s = 8, 1); inside a pair a wave runs both passes back to back.  The
// emulation therefore executes WAVE by WAVE (all lanes of pass 1, then all lanes of pass 2, then the next wave): if a
// pass needed data of another wave, this order would expose it.
template <bool INV, int L1, int L2>
static void hc_pair(double *re, double *im, const double *tw) {
  for (int w = 0; w < PTA_FFT_THREADS / 64; ++w) {
    for (int l = 0; l < 64; ++l) pta_fft_pass<INV, L1>(re, im, tw, 64 * w + l);
    for (int l = 0; l < 64; ++l) pta_fft_pass<INV, L2>(re, im, tw, 64 * w + l);
  }
}

extern "C" {

void hc_philox(const uint32_t *ctr, const uint32_t *key, uint32_t *out) {
  pta_u32x4 c = {ctr[0], ctr[1], ctr[2], ctr[3]};
  pta_u32x4 v = pta_philox4x32_10(c, key[0], key[1]);
  out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
}

void hc_normal_pairs(uint64_t seed, uint64_t realisation, uint32_t stream, int npairs, double *out) {
  for (int p = 0; p < npairs; ++p) pta_normal_pair(seed, realisation, stream, (uint32_t)p, out[2 * p], out[2 * p + 1]);
}

void hc_uniform_pairs(uint64_t seed, uint64_t realisation, uint32_t stream, int npairs, double *out) {
  for (int p = 0; p < npairs; ++p) pta_uniform_pair(pta_philox_draw(seed, realisation, stream, (uint32_t)p), out[2 * p], out[2 * p + 1]);
}

// the two transcendentals of the Box-Muller transform: table-driven (product path) and polynomial (cross-check)
void hc_neg2log(const double *u, int n, int poly, double *out) {
  for (int i = 0; i < n; ++i) out[i] = poly ? pta_neg2log_poly(u[i]) : pta_neg2log(u[i]);
}
void hc_sincos_2pi(const double *u, int n, int poly, double *sn, double *cs) {
  for (int i = 0; i < n; ++i) {
    if (poly) pta_sincos_2pi_poly(u[i], sn[i], cs[i]);
    else pta_sincos_2pi(u[i], sn[i], cs[i]);
  }
}

uint32_t hc_stream_id(uint32_t kind, uint32_t pulsar) { return pta_stream_id(kind, pulsar); }

void hc_orf_hd(const double *locs, int P, double *orf) {
  for (int a = 0; a < P; ++a)
    for (int b = 0; b < P; ++b) orf[a * P + b] = pta_orf_hd(locs[2 * a], locs[2 * b], locs[2 * a + 1], locs[2 * b + 1]);
}

void hc_orf_basis(const double *locs, int P, int lmax, double *basis) {
  for (int l = 0; l <= lmax; ++l)
    for (int a = 0; a < P; ++a)
      for (int b = a; b < P; ++b) {
        double v[2 * PTA_ORF_LMAX + 1];
        pta_orf_pair_l(l, locs[2 * a], locs[2 * b], locs[2 * a + 1], locs[2 * b + 1], nullptr, v);
        for (int mi = 0; mi <= 2 * l; ++mi) {
          int k = l * l + mi;
          basis[((int64_t)k * P + a) * P + b] = v[mi];
          basis[((int64_t)k * P + b) * P + a] = v[mi];
        }
      }
}

void hc_fft_forward(double *re, double *im, const double *tw) {
  hc_pair<false, 9, 6>(re, im, tw);
  hc_pair<false, 3, 0>(re, im, tw);
}
void hc_fft_inverse(double *re, double *im, const double *tw) {
  hc_pair<true, 0, 3>(re, im, tw);
  hc_pair<true, 6, 9>(re, im, tw);
}
int hc_fft_phys(int i) { return PTA_FFT_PHYS(i); }
int hc_fft_plane() { return PTA_FFT_PLANE; }
}
