// Radix-16 counterparts of the radix-8 building blocks of pta_fft.h, for the ISA-level comparison VERDICT r4 #5 asks for
// (scripts/isa_radix16_table.py): would three radix-16 passes per 4096-point FFT (4096 = 16^3) issue fewer instructions than the four
// radix-8 passes of the production chirp-z kernel?  __host__ __device__: tests/test_hostcheck-style check against numpy.fft in the script.
#pragma once
#include "pta_fft.h"

// 16-point DFT, X_q = sum_p x_p e^{-+ 2 pi i p q / 16} (INV: + sign), in place: 4 x 4 decomposition (p = 4 p1 + p0, q = q0 + 4 q1 ... )
//   1. DFT4 over p1 for each p0:  y[p0][q0] = sum_p1 x[4 p1 + p0] W4^{p1 q0}
//   2. twiddle y[p0][q0] *= W16^{p0 q0}
//   3. DFT4 over p0 for each q0:  X[q0 + 4 q1] = sum_p0 y[p0][q0] W4^{p0 q1}
template <bool INV>
PTA_HD void pta_dft4(pta_cplx &a, pta_cplx &b, pta_cplx &c, pta_cplx &d) {  // in: x0..x3, out: X0..X3
  const pta_cplx s0 = pta_cadd(a, c), s1 = pta_csub(a, c), s2 = pta_cadd(b, d), s3 = pta_cmuli<INV>(pta_csub(b, d));
  a = pta_cadd(s0, s2);
  c = pta_csub(s0, s2);
  b = pta_cadd(s1, s3);
  d = pta_csub(s1, s3);
}

template <bool INV>
PTA_HD pta_cplx pta_w16(pta_cplx v, int e) {  // v * W16^e, W16 = e^{-+ 2 pi i / 16}; e compile-time after unrolling
  const double c1 = 0.92387953251128675613, s1 = 0.38268343236508977173, h = 0.70710678118654752440;
  switch (e & 15) {
    case 0: return v;
    case 4: return pta_cmuli<INV>(v);
    case 8: return pta_cplx{-v.re, -v.im};
    case 12: return pta_cmuli<!INV>(v);
    case 2: return INV ? pta_cplx{h * (v.re - v.im), h * (v.re + v.im)} : pta_cplx{h * (v.re + v.im), h * (v.im - v.re)};
    case 6: return INV ? pta_cplx{-h * (v.re + v.im), h * (v.re - v.im)} : pta_cplx{h * (v.im - v.re), -h * (v.re + v.im)};
    default: {
      // generic: W16^e = cos(pi e / 8) -+ i sin(pi e / 8)
      const double cc[16] = {1, c1, h, s1, 0, -s1, -h, -c1, -1, -c1, -h, -s1, 0, s1, h, c1};
      const double ss[16] = {0, s1, h, c1, 1, c1, h, s1, 0, -s1, -h, -c1, -1, -c1, -h, -s1};
      const pta_cplx w = {cc[e & 15], INV ? ss[e & 15] : -ss[e & 15]};
      return pta_cmul(v, w);
    }
  }
}

template <bool INV>
PTA_HD void pta_dft16(pta_cplx *v) {
  pta_cplx y[4][4];
#pragma unroll
  for (int p0 = 0; p0 < 4; ++p0) {
    pta_cplx a = v[p0], b = v[4 + p0], c = v[8 + p0], d = v[12 + p0];
    pta_dft4<INV>(a, b, c, d);
    y[p0][0] = a;
    y[p0][1] = pta_w16<INV>(b, p0 * 1);
    y[p0][2] = pta_w16<INV>(c, p0 * 2);
    y[p0][3] = pta_w16<INV>(d, p0 * 3);
  }
#pragma unroll
  for (int q0 = 0; q0 < 4; ++q0) {
    pta_cplx a = y[0][q0], b = y[1][q0], c = y[2][q0], d = y[3][q0];
    pta_dft4<INV>(a, b, c, d);
    v[q0] = a;
    v[q0 + 4] = b;
    v[q0 + 8] = c;
    v[q0 + 12] = d;
  }
}

// twiddles of one radix-16 butterfly from ONE table load (the production kernel's TW = 1 rule): w[q] = W^(q m1), q = 1 .. 15
PTA_HD void pta_fft16_twiddles(const double *tw, int m1, pta_cplx *w) {
  const pta_cplx *tw2 = reinterpret_cast<const pta_cplx *>(tw);
  w[1] = tw2[m1];
  w[2] = pta_cmul(w[1], w[1]);
  w[4] = pta_cmul(w[2], w[2]);
  w[8] = pta_cmul(w[4], w[4]);
  w[3] = pta_cmul(w[1], w[2]);
  w[5] = pta_cmul(w[1], w[4]);
  w[6] = pta_cmul(w[2], w[4]);
  w[7] = pta_cmul(w[3], w[4]);
#pragma unroll
  for (int q = 9; q < 16; ++q) w[q] = pta_cmul(w[q - 8], w[8]);
}

template <bool INV, bool TWIDDLED>
PTA_HD void pta_fft16_core(pta_cplx *v, const pta_cplx *w) {
  if (INV && TWIDDLED) {
#pragma unroll
    for (int q = 1; q < 16; ++q) v[q] = pta_cmul(v[q], pta_cplx{w[q].re, -w[q].im});
  }
  pta_dft16<INV>(v);
  if (!INV && TWIDDLED) {
#pragma unroll
    for (int q = 1; q < 16; ++q) v[q] = pta_cmul(v[q], w[q]);
  }
}
