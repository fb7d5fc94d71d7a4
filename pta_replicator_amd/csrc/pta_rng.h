// Counter-based Gaussian draws for "throughput mode" (SURVEY.md §7).
//
// The reference consumes NumPy's process-global legacy MT19937 stream
// (red_noise.py:119,127,176,238-240; white_noise.py:80,105-109,155,182), which is
// inherently serial.  On the device every normal deviate is instead a pure function of
//     (seed, realisation r, stream s, pair index p)
// through Philox-4x32-10 (Salmon et al. 2011) + one Box-Muller transform, so any lane of any
// kernel can produce exactly the deviate it needs, results do not depend on the launch
// geometry or the number of GPUs, and a dumped draw buffer (pta_rng_fill_normal) can be
// replayed through the reference algebra on the host (tests/).
//
// Everything here is __host__ __device__ so that tests/hostcheck can compile the very same
// code with g++ and compare it with NumPy on a machine without a GPU.
#pragma once
#include <stdint.h>
#include <math.h>
#include "pta_rng_tables.h"

#ifndef PTA_HD
#if defined(__HIPCC__)
#define PTA_HD __host__ __device__ __forceinline__
#else
#define PTA_HD inline
#endif
#endif

// stream ids (counter word 1): which random object of realisation r a deviate belongs to
#define PTA_STREAM_GWB 1u   // + pulsar index: pair p = frequency bin k  -> (re, im) of w[a,k]   (red_noise.py:240)
#define PTA_STREAM_RN 2u    // + pulsar index: pair p -> coefficients (2p, 2p+1)                 (red_noise.py:127)
#define PTA_STREAM_WN 3u    // + pulsar index: pair p = TOA i -> (z1[i], z2[i])                  (white_noise.py:105-109)
#define PTA_STREAM_ECORR 4u // + pulsar index: pair p = epoch e>>1, branch e&1                   (white_noise.py:182)
#define PTA_STREAM_TD 5u    // + pulsar index: TD-mode z[i] of the pulsar's N_a x N_a factor, pair p = i>>1, branch i&1
#define PTA_STREAM_TDGW 6u  // + pulsar index: TD-mode z[j] of the npts x npts GWB grid factor, pair p = j>>1, branch j&1

PTA_HD uint32_t pta_stream_id(uint32_t kind, uint32_t pulsar) { return (kind << 24) | (pulsar & 0xFFFFFFu); }

struct pta_u32x4 {
  uint32_t x, y, z, w;
};

PTA_HD void pta_mulhilo32(uint32_t a, uint32_t b, uint32_t &hi, uint32_t &lo) {
  uint64_t p = (uint64_t)a * (uint64_t)b;
  hi = (uint32_t)(p >> 32);
  lo = (uint32_t)p;
}

// a ^ b ^ c in ONE instruction: gfx950 has no v_xor3_b32, but it has v_bitop3_b32 (any 3-input boolean function; truth table
// 0x96 = parity), which the compiler does not select for a chain of two xors by itself.  Two of these per Philox round instead
// of four v_xor_b32: 20 of the 56 instructions of a Philox-4x32-10 block.
PTA_HD uint32_t pta_xor3(uint32_t a, uint32_t b, uint32_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
#else
  return a ^ b ^ c;
#endif
}

// Philox-4x32-10, Random123 reference constants.
PTA_HD pta_u32x4 pta_philox4x32_10(pta_u32x4 c, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint32_t hi0, lo0, hi1, lo1;
    pta_mulhilo32(0xD2511F53u, c.x, hi0, lo0);
    pta_mulhilo32(0xCD9E8D57u, c.z, hi1, lo1);
    pta_u32x4 n;
    n.x = pta_xor3(hi1, c.y, k0);
    n.y = lo1;
    n.z = pta_xor3(hi0, c.w, k1);
    n.w = lo0;
    c = n;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return c;
}

// counter layout: (pair index, stream id, realisation lo, realisation hi); key = 64-bit seed
PTA_HD pta_u32x4 pta_philox_draw(uint64_t seed, uint64_t realisation, uint32_t stream, uint32_t pair) {
  pta_u32x4 c;
  c.x = pair;
  c.y = stream;
  c.z = (uint32_t)realisation;
  c.w = (uint32_t)(realisation >> 32);
  return pta_philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
}

// Two 52-bit uniforms from one Philox block: u1 in (0,1], u2 in [0,1).  The 52 random bits are dropped
// straight into the mantissa of a double in [1,2) (three integer ops + one exact subtraction per uniform).
PTA_HD double pta_bits_to_double(uint64_t b) {
  double d;
  __builtin_memcpy(&d, &b, sizeof(d));
  return d;
}
PTA_HD uint64_t pta_double_to_bits(double d) {
  uint64_t b;
  __builtin_memcpy(&b, &d, sizeof(b));
  return b;
}

PTA_HD void pta_uniform_pair(pta_u32x4 v, double &u1, double &u2) {
  uint64_t a = ((((uint64_t)v.x << 32) | v.y) >> 12) | 0x3FF0000000000000ull;
  uint64_t b = ((((uint64_t)v.z << 32) | v.w) >> 12) | 0x3FF0000000000000ull;
  u1 = 2.0 - pta_bits_to_double(a);
  u2 = pta_bits_to_double(b) - 1.0;
}

// x/y and sqrt(x) without the IEEE division / square-root expansions: those lean on VCC (v_div_scale ->
// v_div_fmas) and on long fix-up tails, which serialises the eight independent Box-Muller chains a thread of the
// fused kernel keeps in flight.  Hardware reciprocal / reciprocal-sqrt seeds (~2^-23) + one Newton step + one exact-
// residual correction: < 1 ulp for the well-scaled arguments used here.  The host twins are the plain operators.
// fma(x, p, c) with a compile-time constant addend.  Left to itself the compiler turns a Horner step into v_mov_b64 (constant ->
// accumulator) + v_fmac_f64; the three-source form reads the constant from a scalar register pair instead: one VALU
// instruction per step (11 fewer per Box-Muller pair).  The host twin is plain fma - identical rounding.
PTA_HD double pta_fma_k(double x, double p, double c) {
#if defined(__HIP_DEVICE_COMPILE__)
  double d;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(x), "v"(p), "s"(c));
  return d;
#else
  return fma(x, p, c);
#endif
}

PTA_HD double pta_div(double x, double y) {
#if defined(__HIP_DEVICE_COMPILE__)
  double r = __builtin_amdgcn_rcp(y);      // relative error ~2^-23
  r = fma(fma(-y, r, 1.0), r, r);          // ~2^-46
  double q = x * r;
  return fma(fma(-y, q, x), r, q);         // exact residual x - y q, corrected with the 2^-46 reciprocal: error ~2^-92 + rounding
#else
  return x / y;
#endif
}
PTA_HD double pta_sqrt_pos(double x) {  // x >= 0; exact zero is nudged to 1e-300 (sqrt -> 1e-150) to keep rsq finite
#if defined(__HIP_DEVICE_COMPILE__)
  x = fmax(x, 1e-300);
  double r = __builtin_amdgcn_rsq(x);
  double g = x * r, h = 0.5 * r;
  double d = fma(-h, g, 0.5);              // one coupled Newton step: g ~ sqrt(x), h ~ 1/(2 sqrt(x)) to ~2^-46
  g = fma(g, d, g);
  h = fma(h, d, h);
  return fma(fma(-g, g, x), h, g);         // exact residual x - g^2, corrected: error ~2^-92 + rounding
#else
  return sqrt(fmax(x, 1e-300));
#endif
}

// (kept as the cross-check of the table-driven transform below and for the microbenchmark's A/B; not on the product path)
// -2 ln(u) for u in (0,1].  Classic argument reduction u = 2^e m, m in [sqrt(1/2), sqrt(2)), then
// ln(1+f) = f - (f^2/2 - s (f^2/2 + R(s^2))), s = f/(2+f), with the degree-14 minimax R of Sun's fdlibm
// (public domain, e_log.c; < 1 ulp).  About a third of the instructions of the generic library log(),
// which matters because Gaussian generation is the VALU-bound part of the whole pipeline.
PTA_HD double pta_neg2log_poly(double u) {
  const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
  const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
               Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
               Lg7 = 1.479819860511658591e-01;
  uint64_t b = pta_double_to_bits(u);
  int e = (int)(b >> 52) - 1023;
  double m = pta_bits_to_double((b & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull);  // [1,2)
  if (m > 1.4142135623730951) {
    m *= 0.5;
    e += 1;
  }
  double f = m - 1.0;
  double s = pta_div(f, 2.0 + f);
  double z = s * s, w = z * z;
  double t1 = w * pta_fma_k(w, fma(w, Lg6, Lg4), Lg2);
  double t2 = z * pta_fma_k(w, pta_fma_k(w, fma(w, Lg7, Lg5), Lg3), Lg1);
  double R = t2 + t1;
  double hfsq = 0.5 * f * f;
  double dk = (double)e;
  double lnu = dk * ln2_hi - ((hfsq - fma(s, hfsq + R, dk * ln2_lo)) - f);
  return -2.0 * lnu;
}

// sin(2 pi u), cos(2 pi u) for u in [0,1): exact reduction to the nearest quarter turn, then the fdlibm
// kernels (k_sin.c / k_cos.c, |x| <= pi/4, < 1 ulp) and a quadrant rotation.
PTA_HD void pta_sincos_2pi_poly(double u, double &sn, double &cs) {
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
               S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
               C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  double q = rint(4.0 * u);        // 0..4
  double r = u - 0.25 * q;         // [-1/8, 1/8], exact
  double x = 6.283185307179586 * r;
  double z = x * x;
  double ps = pta_fma_k(z, pta_fma_k(z, pta_fma_k(z, pta_fma_k(z, fma(z, S6, S5), S4), S3), S2), S1);
  double s = fma(x * z, ps, x);
  double pc = z * pta_fma_k(z, pta_fma_k(z, pta_fma_k(z, pta_fma_k(z, fma(z, C6, C5), C4), C3), C2), C1);
  double c = 1.0 - (0.5 * z - z * pc);
  int k = (int)q & 3;
  double s1 = (k & 1) ? c : s;
  double c1 = (k & 1) ? s : c;
  sn = (k == 2 || k == 3) ? -s1 : s1;   // k: 0 (s,c) 1 (c,-s) 2 (-s,-c) 3 (-c,s)
  cs = (k == 1 || k == 2) ? -c1 : c1;
}


// ---- table-driven transform (default) --------------------------------------------------------------------------------------
// Gaussian generation is the VALU-bound part of the whole pipeline (~2/3 of the fused kernel's instructions), so the two
// transcendentals trade polynomial length for one 16-byte table read each.  The table (pta_rng_tables.h, 2.5 KB) is staged into
// LDS once per workgroup - pta_rng_stage_tables() + a barrier at the top of every kernel that draws - because 64 lanes reading
// 64 random 16-byte entries cost the CU's single texture-addresser as many cycles from global memory as the instructions saved.
//
//  -2 ln(u), u = 2^e m: the top 7 mantissa bits j select (c_j, T_j = 2 ln c_j) with c_j ~ 1 / (centre of the interval), the
//    mantissa is re-centred to m' in [0.707, 1.414) by integer arithmetic on the exponent field (k = e or e + 1), and
//        -2 ln u = -k 2 ln2 + T_j - 2 ln(1 + r),   r = m' c_j - 1  (one fma, |r| <= 2^-8),
//    with -2 ln(1 + r) = -2 r + r^2 (1 - 2/3 r + 1/2 r^2 - 2/5 r^3 + 1/3 r^4 - 2/7 r^5), truncation 2e-18 relative.  Near u = 1
//    the table entry is (1, 0), so the result keeps its accuracy RELATIVE to itself where the Box-Muller radius goes to zero.
//    ~25 instructions against 38 for the fdlibm-style evaluation above (no division, no degree-14 polynomial).
//  sin / cos(2 pi u): nearest 32nd of a turn from the table (exact zeros at the quarter turns), remainder |x| <= pi / 32 through
//    the Taylor series to x^9 / x^8 (truncation < 3e-17), one rotation: ~23 instructions against 37 (no quadrant selects).
// Accuracy (tests/test_hostcheck.py, scripts/gpu_rng_accuracy.py): -2 ln u within 1.5 ulp, sin / cos within 4.5 ulp (the rotation sums
// two rounded products) and 2.3e-16 absolute, every deviate within 5 ulp of an 80-bit evaluation of the same uniforms (99.9 %: 2.7).
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ double *pta_rng_lds() {
  __shared__ double __attribute__((aligned(16))) tab[PTA_RNG_TAB_DOUBLES];
  return tab;
}
// every thread of the workgroup calls this once, before its first draw; the caller then issues __syncthreads()
__device__ __forceinline__ void pta_rng_stage_tables() {
  double *t = pta_rng_lds();
  for (int i = threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z); i < PTA_RNG_TAB_DOUBLES; i += blockDim.x * blockDim.y * blockDim.z)
    t[i] = pta_rng_tab[i];
}
#define PTA_RNG_TABLE pta_rng_lds()
#else
PTA_HD void pta_rng_stage_tables() {}  // host pass of hipcc / host twin: the table is read in place
#define PTA_RNG_TABLE pta_rng_tab
#endif

PTA_HD double pta_neg2log(double u) {
  const double L2H = 1.38629436073824763298e+00, L2L = 3.81642985854117540004e-10;  // 2 ln2 = L2H + L2L, L2H with 21 trailing zero bits
  const uint64_t b = pta_double_to_bits(u);
  const uint32_t hi = (uint32_t)(b >> 32);
  const uint32_t j = (hi >> (20 - PTA_RNG_LOG_BITS)) & ((1u << PTA_RNG_LOG_BITS) - 1u);
  // -k: the add carries into the exponent field iff j >= PTA_RNG_LOG_CENTRED
  const int nk = 1023 - (int)((hi + (((1u << PTA_RNG_LOG_BITS) - PTA_RNG_LOG_CENTRED) << (20 - PTA_RNG_LOG_BITS))) >> 20);
  const uint32_t hi2 = hi + ((uint32_t)nk << 20);                                      // exponent field of m' = u 2^-k
  const double m = pta_bits_to_double(((uint64_t)hi2 << 32) | (b & 0xFFFFFFFFull));
  const double *e = PTA_RNG_TABLE + 2 * j;
  const double c = e[0], T = e[1];
  const double r = fma(m, c, -1.0);
  const double W = pta_fma_k(r, pta_fma_k(r, pta_fma_k(r, pta_fma_k(r, fma(r, -2.85714285714285698425e-01, 3.33333333333333314830e-01),
                                                   -4.00000000000000022204e-01), 0.5), -6.66666666666666629659e-01), 1.0);
  const double t = fma(r * r, W, -2.0 * r);
  const double nkd = (double)nk;
  return fma(nkd, L2H, T + fma(nkd, L2L, t));
}

PTA_HD void pta_sincos_2pi(double u, double &sn, double &cs) {
  const double kd = rint((double)PTA_RNG_SC_N * u);            // 0..32
  const double r = fma(kd, -1.0 / PTA_RNG_SC_N, u);            // [-1/64, 1/64], exact
  const int k = (int)kd & (PTA_RNG_SC_N - 1);
  const double *e = PTA_RNG_TABLE + PTA_RNG_SC_OFF + 2 * k;
  const double sa = e[0], ca = e[1];
  const double x = 6.283185307179586 * r;
  const double z = x * x;
  const double ps = pta_fma_k(z, pta_fma_k(z, pta_fma_k(z, fma(z, 2.75573192239858925110e-06, -1.98412698412698412526e-04), 8.33333333333333321769e-03),
                                       -1.66666666666666657415e-01), 1.0);
  const double cx = pta_fma_k(z, pta_fma_k(z, pta_fma_k(z, fma(z, 2.48015873015873015658e-05, -1.38888888888888894189e-03), 4.16666666666666643537e-02),
                                       -0.5), 1.0);
  const double sx = x * ps;
  sn = fma(sa, cx, ca * sx);
  cs = fma(ca, cx, -(sa * sx));
}

// Box-Muller: (z0, z1) iid N(0,1).
// fast = 0 (default): fp64 table-driven transform; the deviates sit within a few ulp of an 80-bit evaluation of the same
//           uniforms (scripts/gpu_rng_accuracy.py, 2^21 deviates) and are reproducible on the host to ~1e-16.
// fast = 1 ("fast RNG math", opt-in): the SAME uniforms through the hardware fp32 transcendentals (v_log_f32, v_sqrt_f32,
//           v_sin_f32 / v_cos_f32, which take their argument in turns) - deviates accurate to ~1e-6, a statistically
//           irrelevant perturbation of a random number, at 40 % of the instructions.  Signal arithmetic stays fp64.
PTA_HD void pta_normal_pair(uint64_t seed, uint64_t realisation, uint32_t stream, uint32_t pair, double &z0, double &z1, int fast = 0) {
  double u1, u2, s, c;
  pta_uniform_pair(pta_philox_draw(seed, realisation, stream, pair), u1, u2);
  if (fast) {
#if defined(__HIP_DEVICE_COMPILE__)
    float rad = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf((float)u1));  // sqrt(-2 ln2 log2 u1)
    float x = (float)u2;
    z0 = (double)(rad * __builtin_amdgcn_cosf(x));
    z1 = (double)(rad * __builtin_amdgcn_sinf(x));
#else
    float rad = sqrtf(-1.3862943611198906f * log2f((float)u1));
    float x = 6.2831853071795864769f * (float)u2;
    z0 = (double)(rad * cosf(x));
    z1 = (double)(rad * sinf(x));
#endif
    return;
  }
  double rad = pta_sqrt_pos(pta_neg2log(u1));
  pta_sincos_2pi(u2, s, c);
  z0 = rad * c;
  z1 = rad * s;
}

// single deviate with index e of a stream: pair e>>1, branch e&1
PTA_HD double pta_normal_single(uint64_t seed, uint64_t realisation, uint32_t stream, uint32_t e, int fast = 0) {
  double z0, z1;
  pta_normal_pair(seed, realisation, stream, e >> 1, z0, z1, fast);
  return (e & 1u) ? z1 : z0;
}
