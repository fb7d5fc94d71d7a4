// Overlap-reduction-function basis: Hellings-Downs closed form and the Gair et al. (2014) /
// Mingarelli et al. (2013) anisotropic basis, evaluated per pulsar pair.
//
// Replaces spharmORFbasis.correlated_basis (spharmORFbasis.py:385-434) and its callees
// (calczeta :14-35, Fminus00/Fminus01/Fplus01/Fplus00 :43-161, arbORF :164-248, dlmk :251-268,
// Dlmk :271-279, gamma :282-306, arbCompFrame_ORF :309-344, rotated_Gamma_ml :347-359,
// real_rotated_Gammas :362-382).  The reference walks pairs x (l,m) in pure Python (0.27 ms per
// pair at lmax 0, ~40 ms at lmax 4); here one GPU thread owns one (pair, l) and produces all 2l+1
// real-form values.  __host__ __device__ so tests/hostcheck can run the same code on the CPU.
#pragma once
#include <math.h>
#include "pta_rng.h"  // PTA_HD

#define PTA_ORF_LMAX 8  // (2*lmax)! and every intermediate stay exactly representable in fp64

#ifndef PTA_PI
#define PTA_PI 3.14159265358979323846
#endif

PTA_HD double pta_fact(int n) {
  // n! for n <= 18 is exact in fp64; table avoids tgamma round-off
  const double t[19] = {1.0, 1.0, 2.0, 6.0, 24.0, 120.0, 720.0, 5040.0, 40320.0, 362880.0, 3628800.0, 39916800.0,
                        479001600.0, 6227020800.0, 87178291200.0, 1307674368000.0, 20922789888000.0,
                        355687428096000.0, 6402373705728000.0};
  return t[n];
}

// x**p, integer p >= 0, rounded ONCE: the product chain is carried in double-double (error-free products through fma), so
// the result is the correctly rounded power - what NumPy's float64 ** int (libm pow, < 0.52 ulp) returns in all but a
// vanishing fraction of cases.  The finite sums below cancel heavily for l >= 3 (condition ~1e5-1e6): the device library's
// pow() (1-2 ulp) left a 7e-10 parity error at l = 4 against 1e-11 of rounding noise in the reference itself.
PTA_HD double pta_ipow(double x, int p) {
  double hi = 1.0, lo = 0.0;
  for (int i = 0; i < p; ++i) {
    const double ph = hi * x;
    const double pl = fma(hi, x, -ph) + lo * x;  // exact low part of hi * x, plus the carried low part
    const double s = ph + pl;
    lo = pl - (s - ph);
    hi = s;
  }
  return hi;
}

// x**(n2 / 2) for integer n2 >= 0 and x >= 0, rounded once: integer part as above, an odd n2 adds a double-double sqrt(x).
PTA_HD double pta_pow_half(double x, int n2) {
  double hi = 1.0, lo = 0.0;
  for (int i = 0; i < (n2 >> 1); ++i) {
    const double ph = hi * x;
    const double pl = fma(hi, x, -ph) + lo * x;
    const double s = ph + pl;
    lo = pl - (s - ph);
    hi = s;
  }
  if (n2 & 1) {
    const double r = sqrt(x);                                   // correctly rounded (IEEE)
    const double rl = (r > 0.0) ? fma(-r, r, x) / (2.0 * r) : 0.0;  // sqrt(x) = r + rl to ~2^-100
    const double ph = hi * r;
    const double pl = fma(hi, r, -ph) + (hi * rl + lo * r);
    return ph + pl;
  }
  return hi;
}

PTA_HD double pta_pow2i(int e) { return ldexp(1.0, e); }
PTA_HD double pta_sgn(int n) { return (n & 1) ? -1.0 : 1.0; }

// angular separation with the reference's exact-equality and clamping rules (spharmORFbasis.py:14-35)
PTA_HD double pta_calczeta(double phi1, double phi2, double th1, double th2) {
  if (phi1 == phi2 && th1 == th2) return 0.0;
  double arg = sin(th1) * sin(th2) * cos(phi1 - phi2) + cos(th1) * cos(th2);
  if (arg < -1.0) return PTA_PI;
  if (arg > 1.0) return 0.0;
  return acos(arg);
}

// sum_{i=0..imax} sum_{j=m..l} (-1)^s 2^(i-j) q!(l+j)! (2^p - base^p) / (i!(q-i)! j!(l-j)!(j-m)! p)
//   p = q-i+j-m+extra_p ;  s = q-i+j+m ("minus" family) or l+q-i+j ("plus" family)
PTA_HD double pta_fsum(int qq, int mm, int ll, double base, int extra_p, bool plus_family, int imax) {
  double tot = 0.0;
  for (int ii = 0; ii <= imax; ++ii) {
    for (int jj = mm; jj <= ll; ++jj) {
      int p = qq - ii + jj - mm + extra_p;
      int s = plus_family ? (ll + qq - ii + jj) : (qq - ii + jj + mm);
      double num = pta_fact(qq) * pta_fact(ll + jj) * (pta_pow2i(p) - pta_ipow(base, p));
      double den = pta_fact(ii) * pta_fact(qq - ii) * pta_fact(jj) * pta_fact(ll - jj) * pta_fact(jj - mm) * (double)p;
      tot += (pta_pow2i(ii - jj) * pta_sgn(s)) * num / den;
    }
  }
  return tot;
}

PTA_HD double pta_Fminus00(int qq, int mm, int ll, double c) { return pta_fsum(qq, mm, ll, 1.0 + c, 1, false, qq); }
PTA_HD double pta_Fminus01(int qq, int mm, int ll, double c) { return pta_fsum(qq, mm, ll, 1.0 + c, 2, false, qq); }
PTA_HD double pta_Fplus00(int qq, int mm, int ll, double c) { return pta_fsum(qq, mm, ll, 1.0 - c, 1, true, qq); }
PTA_HD double pta_Fplus01(int qq, int mm, int ll, double c) {  // spharmORFbasis.py:97-134
  double omc = 1.0 - c;
  double tot = pta_fsum(qq, mm, ll, omc, 0, true, qq - 1);
  for (int jj = mm + 1; jj <= ll; ++jj) {
    tot += (pta_pow2i(qq - jj) * pta_sgn(ll + jj)) * (pta_fact(ll + jj) * (pta_pow2i(jj - mm) - pta_ipow(omc, jj - mm))) /
           (pta_fact(jj) * pta_fact(ll - jj) * pta_fact(jj - mm) * (double)(jj - mm));
  }
  tot += (pta_sgn(ll + mm) * pta_pow2i(qq - mm) * pta_fact(ll + mm) * log(2.0 / omc)) / (pta_fact(mm) * pta_fact(ll - mm));
  return tot;
}

// computational-frame Gamma_lm, zeta in (0, pi]   (spharmORFbasis.py:164-248).  c = cos(zeta) is an INPUT: the finite sums
// below cancel heavily for l >= 3 (condition ~1e5-1e6 in c), so the caller may supply the very c the reference computes
// (host libm) instead of the device's cos(acos(.)) - the 1-2 ulp between them was the whole l >= 3 parity error.
PTA_HD double pta_arbORF(int mm, int ll, double zeta, double c) {
  const double NORM = 3.0 / (8.0 * PTA_PI);
  double pre = sqrt((2.0 * ll + 1.0) * PTA_PI);
  if (mm == 0) {
    double body = -(1.0 + c) * pta_Fminus00(0, 0, ll, c);
    if (ll <= 2) {
      double delta = (ll == 0) ? 1.0 + c / 3.0 : (ll == 1 ? -(1.0 + c) / 3.0 : 2.0 * c / 15.0);
      body = delta - (1.0 + c) * pta_Fminus00(0, 0, ll, c);
    }
    if (zeta != 0.0) body = body - (1.0 - c) * pta_Fplus01(1, 0, ll, c);
    return NORM * 0.5 * pre * body;
  }
  if (mm == 1) {
    double a = pta_pow_half(1.0 + c, 3) / pta_pow_half(1.0 - c, 1);
    double b = pta_pow_half(1.0 - c, 3) / pta_pow_half(1.0 + c, 1);
    double body = -a * pta_Fminus00(1, 1, ll, c) - b * pta_Fplus01(2, 1, ll, c);
    if (ll == 1 || ll == 2) {
      double delta = (ll == 1) ? 2.0 * sin(zeta) / 3.0 : -2.0 * sin(zeta) / 5.0;
      body = delta - a * pta_Fminus00(1, 1, ll, c) - b * pta_Fplus01(2, 1, ll, c);
    }
    return NORM * 0.25 * pre * sqrt(pta_fact(ll - 1) / pta_fact(ll + 1)) * body;
  }
  // exponents h + 1, h, h - 1 with h = m / 2: half-integers, evaluated as x^(n2/2) with n2 = m + 2, m, m - 2
  double body = (pta_pow_half(1.0 + c, mm + 2) / pta_pow_half(1.0 - c, mm)) * pta_Fminus00(mm, mm, ll, c) -
                (pta_pow_half(1.0 + c, mm) / pta_pow_half(1.0 - c, mm - 2)) * pta_Fminus01(mm - 1, mm, ll, c) +
                (pta_pow_half(1.0 - c, mm + 2) / pta_pow_half(1.0 + c, mm)) * pta_Fplus01(mm + 1, mm, ll, c) -
                (pta_pow_half(1.0 - c, mm) / pta_pow_half(1.0 + c, mm - 2)) * pta_Fplus00(mm, mm, ll, c);
  return -NORM * 0.25 * pre * sqrt(pta_fact(ll - mm) / pta_fact(ll + mm)) * body;
}

// zeta == 0 closed forms (pulsar-term doubling) and zeta == pi special cases (spharmORFbasis.py:309-344)
PTA_HD double pta_compframe_orf(int mm, int ll, double zeta, double c) {
  const double NORM = 3.0 / (8.0 * PTA_PI);
  if (zeta == 0.0) {
    if (ll == 0) return 2.0 * NORM * 0.25 * sqrt(PTA_PI * 4.0) * (1.0 + (c / 3.0));
    if (ll == 1 && mm == 0) return -2.0 * 0.5 * NORM * sqrt(PTA_PI / 3.0) * (1.0 + c);
    if (ll == 2 && mm == 0) return 2.0 * 0.25 * NORM * (4.0 / 3.0) * sqrt(PTA_PI / 5.0) * c;
    return 0.0;
  }
  if (zeta == PTA_PI) {
    if (ll > 2 || ((ll == 1 || ll == 2) && mm != 0)) return 0.0;
    return pta_arbORF(mm, ll, zeta, c);
  }
  return pta_arbORF(mm, ll, zeta, c);
}

// terminating Gauss series 2F1(a,b;c;z), a = m-l <= 0 and b = -k-l <= 0 integers (what
// scipy.special.hyp2f1 evaluates at spharmORFbasis.py:262)
PTA_HD double pta_hyp2f1_terminating(int a, int b, int c, double z) {
  int nterms = (-a < -b) ? -a : -b;
  double term = 1.0, tot = 1.0;
  for (int n = 0; n < nterms; ++n) {
    term *= ((double)(a + n) * (double)(b + n)) / ((double)(c + n) * (n + 1.0)) * z;
    tot += term;
  }
  return tot;
}

// Wigner small-d as the reference defines it (spharmORFbasis.py:251-268)
PTA_HD double pta_dlmk(int l, int m, int k, double theta) {
  double sign = 1.0;
  if (m < k) {
    sign = pta_sgn(m - k);
    int t = m;
    m = k;
    k = t;
  }
  double factor = sqrt(pta_fact(l - k) * pta_fact(l + m) / pta_fact(l + k) / pta_fact(l - m));
  double ch = cos(theta / 2.0), sh = sin(theta / 2.0), th = tan(theta / 2.0);
  double part2 = pta_ipow(ch, 2 * l + k - m) * pta_ipow(-sh, m - k) / pta_fact(m - k);
  double part3 = pta_hyp2f1_terminating(m - l, -k - l, m - k + 1, -(th * th));
  return sign * factor * part2 * part3;
}

// third Euler angle (spharmORFbasis.py:282-306)
PTA_HD double pta_third_euler(double phi1, double phi2, double th1, double th2) {
  double g = 0.0;
  if (!(phi1 == phi2 && th1 == th2)) {
    g = atan(sin(th2) * sin(phi2 - phi1) / (cos(th1) * sin(th2) * cos(phi1 - phi2) - sin(th1) * cos(th2)));
  }
  double dummy = cos(g) * cos(th1) * sin(th2) * cos(phi1 - phi2) + sin(g) * sin(th2) * sin(phi2 - phi1) -
                 cos(g) * sin(th1) * cos(th2);
  return (dummy >= 0.0) ? g : PTA_PI + g;
}

// All 2l+1 real-form cosmic-frame values for one pair and one l; out[m+l], m = -l..l
// (correlated_basis inner body, spharmORFbasis.py:400-432).
// zc = {zeta, cos(zeta)} of the pair as the reference computes them (host), or NULL to derive both here.
PTA_HD void pta_orf_pair_l(int l, double phi1, double phi2, double th1, double th2, const double *zc, double *out) {
  double zeta = zc ? zc[0] : pta_calczeta(phi1, phi2, th1, th2);
  double cz = zc ? zc[1] : cos(zeta);
  double gam[2 * PTA_ORF_LMAX + 1];
  for (int mm = 0; mm <= l; ++mm) {
    double v = pta_compframe_orf(mm, l, zeta, cz);
    gam[l + mm] = v;
    gam[l - mm] = pta_sgn(mm) * v;  // Gamma_{l,-m} = (-1)^m Gamma_{lm} in the computational frame
  }
  double g3 = pta_third_euler(phi1, phi2, th1, th2);
  // R(m) = e^{+i m phi1} sum_k d^l_{mk}(theta1) e^{+i k gamma} Gamma_k   ( = sum_k conj(D^l_mk) Gamma_k )
  double Rre[2 * PTA_ORF_LMAX + 1], Rim[2 * PTA_ORF_LMAX + 1];
  for (int m = -l; m <= l; ++m) {
    double sre = 0.0, sim = 0.0;
    for (int k = -l; k <= l; ++k) {
      double d = pta_dlmk(l, m, k, th1) * gam[k + l];
      sre += d * cos(k * g3);
      sim += d * sin(k * g3);
    }
    double cr = cos(m * phi1), ci = sin(m * phi1);
    Rre[m + l] = cr * sre - ci * sim;
    Rim[m + l] = cr * sim + ci * sre;
  }
  const double is2 = 1.0 / sqrt(2.0);
  for (int m = -l; m <= l; ++m) {
    double v;
    if (m > 0) {
      v = is2 * (Rre[m + l] + pta_sgn(m) * Rre[-m + l]);
    } else if (m == 0) {
      v = Rre[l];
    } else {  // Re( (R(-m) - (-1)^m R(m)) / (sqrt(2) i) ) = Im(...)/sqrt(2)
      v = is2 * (Rim[-m + l] - pta_sgn(m) * Rim[m + l]);
    }
    out[m + l] = v;
  }
}

// lmax = 0, clm = [sqrt(4 pi)] fast path:  ORF_ab = 2 sqrt(4 pi) Gamma_00 = 1 + c/3 ... = 2*HD(zeta)
//   zeta == 0 (same position incl. the diagonal, or a separation that rounds to 0): 2 (pulsar-term
//   doubling, spharmORFbasis.py:311,327-329)
PTA_HD double pta_orf_hd(double phi1, double phi2, double th1, double th2) {
  double zeta = pta_calczeta(phi1, phi2, th1, th2);
  if (zeta == 0.0) return 2.0;
  double x = (1.0 - cos(zeta)) / 2.0;
  return 2.0 * (0.5 - x / 4.0 + 1.5 * x * log(x));
}
