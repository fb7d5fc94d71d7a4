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
// in, natural out): the pointwise product happens in digit-reversed order, so no reordering pass exists.
//
// Work distribution (512 threads = 8 waves, one radix-8 butterfly per thread per pass).  4096 = 64 x 64: the two
// long-stride passes (s = 512, 64) only mix the 64 elements of a "column" {o + 64 m}, the two short-stride passes
// (s = 8, 1) only the 64 elements of a contiguous block.  Wave w owns columns 8w..8w+7 in the first pair and blocks
// 8w..8w+7 in the second; with the padding below every pass reads and writes LDS without bank conflicts.
// All code is per-"thread" work, __host__ __device__, so tests/hostcheck can run it on the CPU.
#pragma once
#include "pta_rng.h"  // PTA_HD

#define PTA_FFT_N 4096
#define PTA_FFT_THREADS 512
// physical LDS index of logical element i: one pad double per 8 elements and eight more per 512 keep the 8-strided
// accesses of every pass, under the wave <-> column/block mapping above, on distinct banks
#define PTA_FFT_PHYS(i) ((i) + ((i) >> 3) + (((i) >> 9) << 3))
#define PTA_FFT_PLANE (PTA_FFT_N + PTA_FFT_N / 8 + 64)  // doubles per plane (re or im)

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

// Stride of a pass: LOG2S in {9, 6, 3, 0} (s = 512, 64, 8, 1).  Butterfly (b, o), o < s, touches logical elements
// b*8s + o + q*s, q = 0..7, which sit at constant physical distances QSTEP (every q*s is a multiple of 8 for s >= 8
// and never crosses a 512 boundary for s < 512; for s = 1 the eight elements are contiguous).
template <int LOG2S>
struct pta_fft_stride {
  static constexpr int S = 1 << LOG2S;
  static constexpr int QSTEP = (LOG2S == 9) ? 512 + 64 + 8 : (LOG2S >= 3 ? S + (S >> 3) : 1);
  static constexpr int TSTEP = 512 >> LOG2S;  // 4096 / (8 s)
};

template <int LOG2S>
PTA_HD int pta_fft_base(int b, int o) {  // logical index of element q = 0
  return (b << (LOG2S + 3)) + o;
}

// (b, o) of the butterfly thread `tid` executes in each pass (wave w = tid >> 6, lane l = tid & 63)
template <int LOG2S>
PTA_HD void pta_fft_map(int tid, int &b, int &o) {
  const int w = tid >> 6, l = tid & 63;
  if (LOG2S == 9) {         // s = 512: column 8w + (l & 7), row group l >> 3
    b = 0;
    o = 8 * w + (l & 7) + 64 * (l >> 3);
  } else if (LOG2S == 6) {  // s = 64: 512-block l >> 3, column 8w + (l & 7)
    b = l >> 3;
    o = 8 * w + (l & 7);
  } else if (LOG2S == 3) {  // s = 8: 64-block 8w + (l >> 3), offset l & 7
    b = 8 * w + (l >> 3);
    o = l & 7;
  } else {                  // s = 1: butterfly = tid
    b = tid;
    o = 0;
  }
}

template <int LOG2S>
PTA_HD void pta_fft_load(const double *re, const double *im, int b, int o, pta_cplx *v) {
  const int p0 = PTA_FFT_PHYS(pta_fft_base<LOG2S>(b, o));
#pragma unroll
  for (int q = 0; q < 8; ++q) v[q] = {re[p0 + q * pta_fft_stride<LOG2S>::QSTEP], im[p0 + q * pta_fft_stride<LOG2S>::QSTEP]};
}

template <int LOG2S>
PTA_HD void pta_fft_store(double *re, double *im, int b, int o, const pta_cplx *v) {
  const int p0 = PTA_FFT_PHYS(pta_fft_base<LOG2S>(b, o));
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    re[p0 + q * pta_fft_stride<LOG2S>::QSTEP] = v[q].re;
    im[p0 + q * pta_fft_stride<LOG2S>::QSTEP] = v[q].im;
  }
}

// the twiddles of one butterfly: w[q] = W^(q m1), W = e^{-2 pi i / 4096}, m1 = o * TSTEP.  Only W^m1, W^2m1 and W^4m1 are
// loaded (for s = 512 the q-strided gathers of all seven touch up to 56 cache lines per wave and q; these three touch 14);
// the other four are products - each one rounding (~1e-16) away from the table value.
template <int LOG2S, int TW = 3>
PTA_HD void pta_fft_twiddles(const double *tw, int o, pta_cplx *w) {
  const pta_cplx *tw2 = reinterpret_cast<const pta_cplx *>(tw);
  const int m1 = o * pta_fft_stride<LOG2S>::TSTEP;
  if (LOG2S > 0) {
    w[1] = tw2[m1];
    if (TW == 1) {  // one load, squarings: two more roundings on w[4..7]
      w[2] = pta_cmul(w[1], w[1]);
      w[4] = pta_cmul(w[2], w[2]);
    } else {
      w[2] = tw2[2 * m1];
      w[4] = tw2[4 * m1];
    }
    w[3] = pta_cmul(w[1], w[2]);
    w[5] = pta_cmul(w[1], w[4]);
    w[6] = pta_cmul(w[2], w[4]);
    w[7] = pta_cmul(w[3], w[4]);
  } else {
#pragma unroll
    for (int q = 1; q < 8; ++q) w[q] = pta_cplx{1.0, 0.0};
  }
}

// the arithmetic of one butterfly, in registers.  w[q] = e^{-2 pi i o q / (8 s)} from pta_fft_twiddles.
//   forward (DIF): butterfly, then output q times W^{o q}          passes in order s = 512, 64, 8, 1
//   inverse (DIT): input q times conj(W)^{o q}, then butterfly     passes in order s = 1, 8, 64, 512
template <bool INV, int LOG2S>
PTA_HD void pta_fft_core(pta_cplx *v, const pta_cplx *w) {
  if (INV && LOG2S > 0) {
#pragma unroll
    for (int q = 1; q < 8; ++q) v[q] = pta_cmul(v[q], pta_cplx{w[q].re, -w[q].im});
  }
  pta_dft8<INV>(v);
  if (!INV && LOG2S > 0) {
#pragma unroll
    for (int q = 1; q < 8; ++q) v[q] = pta_cmul(v[q], w[q]);
  }
}

// load - butterfly - store of one pass for one thread
template <bool INV, int LOG2S, int TW = 3>
PTA_HD void pta_fft_pass(double *re, double *im, const double *tw, int tid) {
  int b, o;
  pta_fft_map<LOG2S>(tid, b, o);
  pta_cplx v[8], w[8];
  pta_fft_twiddles<LOG2S, TW>(tw, o, w);
  pta_fft_load<LOG2S>(re, im, b, o, v);
  pta_fft_core<INV, LOG2S>(v, w);
  pta_fft_store<LOG2S>(re, im, b, o, v);
}
