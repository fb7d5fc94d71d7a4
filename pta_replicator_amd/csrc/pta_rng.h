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
#define PTA_STREAM_TD 5u    // + pulsar index: TD-mode z[i], pair p = i>>1, branch i&1

PTA_HD uint32_t pta_stream_id(uint32_t kind, uint32_t pulsar) { return (kind << 24) | (pulsar & 0xFFFFFFu); }

struct pta_u32x4 {
  uint32_t x, y, z, w;
};

PTA_HD void pta_mulhilo32(uint32_t a, uint32_t b, uint32_t &hi, uint32_t &lo) {
  uint64_t p = (uint64_t)a * (uint64_t)b;
  hi = (uint32_t)(p >> 32);
  lo = (uint32_t)p;
}

// Philox-4x32-10, Random123 reference constants.
PTA_HD pta_u32x4 pta_philox4x32_10(pta_u32x4 c, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint32_t hi0, lo0, hi1, lo1;
    pta_mulhilo32(0xD2511F53u, c.x, hi0, lo0);
    pta_mulhilo32(0xCD9E8D57u, c.z, hi1, lo1);
    pta_u32x4 n;
    n.x = hi1 ^ c.y ^ k0;
    n.y = lo1;
    n.z = hi0 ^ c.w ^ k1;
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

// two 53-bit uniforms from one Philox block: u1 in (0,1], u2 in [0,1)
PTA_HD void pta_uniform_pair(pta_u32x4 v, double &u1, double &u2) {
  uint64_t a = (((uint64_t)v.x << 32) | v.y) >> 11;
  uint64_t b = (((uint64_t)v.z << 32) | v.w) >> 11;
  u1 = ((double)a + 1.0) * 0x1.0p-53;
  u2 = (double)b * 0x1.0p-53;
}

PTA_HD void pta_sincos_2pi(double u, double &s, double &c) {
#if defined(__HIP_DEVICE_COMPILE__)
  sincospi(2.0 * u, &s, &c);
#else
  // host twin (tests only): exact quadrant reduction, then libm
  double x = 2.0 * u;            // [0, 2)
  double q = floor(2.0 * x + 0.5);  // nearest multiple of 1/2
  double rem = x - 0.5 * q;      // [-1/4, 1/4]
  double sr = sin(M_PI * rem), cr = cos(M_PI * rem);
  switch (((int)q) & 3) {
    case 0: s = sr; c = cr; break;
    case 1: s = cr; c = -sr; break;
    case 2: s = -sr; c = -cr; break;
    default: s = -cr; c = sr; break;
  }
#endif
}

// Box-Muller: (z0, z1) iid N(0,1)
PTA_HD void pta_normal_pair(uint64_t seed, uint64_t realisation, uint32_t stream, uint32_t pair, double &z0, double &z1) {
  double u1, u2, s, c;
  pta_uniform_pair(pta_philox_draw(seed, realisation, stream, pair), u1, u2);
  double rad = sqrt(-2.0 * log(u1));
  pta_sincos_2pi(u2, s, c);
  z0 = rad * c;
  z1 = rad * s;
}

// single deviate with index e of a stream: pair e>>1, branch e&1
PTA_HD double pta_normal_single(uint64_t seed, uint64_t realisation, uint32_t stream, uint32_t e) {
  double z0, z1;
  pta_normal_pair(seed, realisation, stream, e >> 1, z0, z1);
  return (e & 1u) ? z1 : z0;
}
