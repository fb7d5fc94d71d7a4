// In-LDS 4096-point complex fp64 FFT (radix 8 x 4 passes) for the chirp-z form of the GWB inverse DFT.
//
// The reference's n = 2Nf-2 = 5998 = 2 x 2999 (prime) inverse FFT (red_noise.py:275-279) of which only npts = 600
// samples are used (:285) is evaluated as Bluestein's convolution: with W = e^{2 pi i / n},
//     x_j = (2 / (n dt)) Re( W^{j^2/2} sum_k (sqrtC_k w_k W^{k^2/2}) W^{-(j-k)^2/2} ),   k = 1..Nf-2,
// and j - k spans 3597 <= 4096 values, so the sum is ONE circular convolution of length L = 4096:
//     forward FFT of the chirped draws  ->  pointwise product with the (precomputed) FFT of the chirp  ->  inverse FFT.
// ~0.5 MFLOP per row on the VALU instead of 3.6 MFLOP of dense DFT on the matrix cores.
//
// Forward = decimation in frequency (natural in, digit-reversed out), inverse = decimation in time (digit-reversed
// in, natural out): the pointwise product happens in digit-reversed order, so no reordering pass exists.  All code
// is per-"thread" work between barriers, __host__ __device__, so tests/hostcheck can run it on the CPU.
#pragma once
#include "pta_rng.h"  // PTA_HD

#define PTA_FFT_N 4096
#define PTA_FFT_THREADS 512  // one radix-8 butterfly per thread per pass: 16 waves per CU with two rows resident
// physical LDS index of logical element i: one pad double per 8 keeps every pass's 8-strided accesses conflict free
#define PTA_FFT_PHYS(i) ((i) + ((i) >> 3))
#define PTA_FFT_PLANE (PTA_FFT_N + PTA_FFT_N / 8)  // doubles per plane (re or im)

struct pta_cplx {
  double re, im;
};

PTA_HD pta_cplx pta_cmul(pta_cplx a, pta_cplx b) { return {fma(a.re, b.re, -(a.im * b.im)), fma(a.re, b.im, a.im * b.re)}; }
PTA_HD pta_cplx pta_cadd(pta_cplx a, pta_cplx b) { return {a.re + b.re, a.im + b.im}; }
PTA_HD pta_cplx pta_csub(pta_cplx a, pta_cplx b) { return {a.re - b.re, a.im - b.im}; }
// multiply by -i (forward) or +i (inverse)
template <bool INV>
PTA_HD pta_cplx pta_cmuli(pta_cplx a) {
  return INV ? pta_cplx{-a.im, a.re} : pta_cplx{a.im, -a.re};
}

// 8-point DFT, X_q = sum_p x_p e^{-+ 2 pi i p q / 8}  (INV: + sign), in place on v[0..7]
template <bool INV>
PTA_HD void pta_dft8(pta_cplx *v) {
  const double h = 0.70710678118654752440;
  pta_cplx a0 = pta_cadd(v[0], v[4]), a4 = pta_csub(v[0], v[4]);
  pta_cplx a1 = pta_cadd(v[1], v[5]), a5 = pta_csub(v[1], v[5]);
  pta_cplx a2 = pta_cadd(v[2], v[6]), a6 = pta_csub(v[2], v[6]);
  pta_cplx a3 = pta_cadd(v[3], v[7]), a7 = pta_csub(v[3], v[7]);
  // odd branch twiddles w^1, w^2, w^3 with w = e^{-+ i pi/4}
  pta_cplx t5 = INV ? pta_cplx{h * (a5.re - a5.im), h * (a5.re + a5.im)} : pta_cplx{h * (a5.re + a5.im), h * (a5.im - a5.re)};
  pta_cplx t6 = pta_cmuli<INV>(a6);
  pta_cplx t7 = INV ? pta_cplx{-h * (a7.re + a7.im), h * (a7.re - a7.im)} : pta_cplx{h * (a7.im - a7.re), -h * (a7.re + a7.im)};
  // DFT4 of (a0,a1,a2,a3) -> X0,X2,X4,X6 ; DFT4 of (a4,t5,t6,t7) -> X1,X3,X5,X7
  pta_cplx c0 = pta_cadd(a0, a2), c2 = pta_csub(a0, a2), c1 = pta_cadd(a1, a3), c3 = pta_cmuli<INV>(pta_csub(a1, a3));
  v[0] = pta_cadd(c0, c1);
  v[4] = pta_csub(c0, c1);
  v[2] = pta_cadd(c2, c3);
  v[6] = pta_csub(c2, c3);
  pta_cplx d0 = pta_cadd(a4, t6), d2 = pta_csub(a4, t6), d1 = pta_cadd(t5, t7), d3 = pta_cmuli<INV>(pta_csub(t5, t7));
  v[1] = pta_cadd(d0, d1);
  v[5] = pta_csub(d0, d1);
  v[3] = pta_cadd(d2, d3);
  v[7] = pta_csub(d2, d3);
}

// One radix-8 pass for one thread (512 butterflies per pass over NT threads).  LOG2S in {9, 6, 3, 0} (stride s = 512,
// 64, 8, 1).  tw[m] = e^{-2 pi i m / 4096} (re, im interleaved).
//   forward (DIF): butterfly, then output q times W^{o q}          passes in order s = 512, 64, 8, 1
//   inverse (DIT): input q times conj(W)^{o q}, then butterfly     passes in order s = 1, 8, 64, 512
// The eight elements of a butterfly sit at constant physical distances (s + s/8 for s >= 8 because q*s is a multiple
// of 8; 1 for s = 1 where base = 8 beta -> 9 beta), so every LDS access is base register + immediate offset.
template <bool INV, int LOG2S, int NT = PTA_FFT_THREADS>
PTA_HD void pta_fft_pass(double *re, double *im, const double *tw, int tid) {
  constexpr int S = 1 << LOG2S;
  constexpr int QSTEP = (LOG2S >= 3) ? S + (S >> 3) : 1;
  constexpr int TSTEP = 512 >> LOG2S;  // 4096 / (8 s)
  const pta_cplx *tw2 = reinterpret_cast<const pta_cplx *>(tw);
#pragma unroll
  for (int h = 0; h < 512 / NT; ++h) {
    const int beta = tid + NT * h;
    const int b = beta >> LOG2S, o = beta & (S - 1);
    const int p0 = PTA_FFT_PHYS((b << (LOG2S + 3)) + o);
    pta_cplx v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = {re[p0 + q * QSTEP], im[p0 + q * QSTEP]};
    const int m1 = o * TSTEP;
    if (INV && LOG2S > 0) {
#pragma unroll
      for (int q = 1; q < 8; ++q) {
        pta_cplx w = tw2[q * m1];
        v[q] = pta_cmul(v[q], pta_cplx{w.re, -w.im});
      }
    }
    pta_dft8<INV>(v);
    if (!INV && LOG2S > 0) {
#pragma unroll
      for (int q = 1; q < 8; ++q) v[q] = pta_cmul(v[q], tw2[q * m1]);
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      re[p0 + q * QSTEP] = v[q].re;
      im[p0 + q * QSTEP] = v[q].im;
    }
  }
}
