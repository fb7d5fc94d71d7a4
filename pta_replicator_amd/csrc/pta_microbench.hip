// Roofline denominators measured on the device itself (bench.py / DESIGN.md): the local guide gives HBM
// and fp32/bf16 peaks but no fp64 figures.
#include "pta_common.h"
#include "pta_mfma.h"
#include "pta_rng.h"

__global__ __launch_bounds__(256) void k_mb_mfma(double *out, int iters) {
  pta_f64x4 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = pta_f64x4{0.0, 0.0, 0.0, 0.0};
  double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = pta_mfma_f64(a, b, acc[i]);
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 123.456) out[0] = s;
}

// how many INDEPENDENT accumulators does a wave need to keep the fp64 matrix pipe busy?  NACC accumulators, each product with its own
// B operand and a shared A operand (the pattern of a wave that walks rows against resident columns), back to back (kind 8; option = NACC,
// bytes = workgroups per CU = waves per SIMD)
template <int NACC>
__global__ __launch_bounds__(256, 2) void k_mb_mfma_nacc(double *out, int iters) {
  pta_f64x4 acc[NACC];
  double b[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) {
    acc[i] = pta_f64x4{0.0, 0.0, 0.0, 0.0};
    b[i] = 1.0 - (threadIdx.x + 32 * i) * 1e-9;
  }
  double a = 1.0 + threadIdx.x * 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = pta_mfma_f64(a, b[i], acc[i]);
    a = a + 1e-12;
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 123.456) out[0] = s;
}

// the register-tile pattern of the GEMM kernels: 4 A fragments x 4 B fragments -> 16 accumulators, acc[i][j] += a[i] b[j]
// (16 independent MFMAs between two uses of an accumulator, operands change from one instruction to the next)
__global__ __launch_bounds__(256, 2) void k_mb_mfma_tile(double *out, int iters) {
  pta_f64x4 acc[4][4];
  double a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a[i] = 1.0 + (threadIdx.x + 64 * i) * 1e-9;
    b[i] = 1.0 - (threadIdx.x + 32 * i) * 1e-9;
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = pta_f64x4{0.0, 0.0, 0.0, 0.0};
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = pta_mfma_f64(a[i], b[j], acc[i][j]);
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  if (s == 123.456) out[0] = s;
}

// do fp64 MFMA and fp64 VALU FMA share the double-precision ALUs?  Even waves run the MFMA register-tile loop, odd waves the FMA
// loop, on the same SIMDs; the host reports the SUM of both rates (two separate resources would give up to 78 + 68 TFLOP/s)
__global__ __launch_bounds__(256, 2) void k_mb_mix(double *out, int iters) {
  double s = 0.0;
  if (((threadIdx.x >> 6) + blockIdx.x) & 1) {
    double x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = 1.0 + i * 1e-3 + threadIdx.x * 1e-9;
    const double a = 1.0000001, b = 1e-9;
    for (int it = 0; it < 4 * iters; ++it) {   // 64 FMAs x 64 lanes x 2 flop = 8192 flop per 4 inner iterations = half an MFMA batch
#pragma unroll
      for (int i = 0; i < 16; ++i) x[i] = fma(x[i], a, b);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i];
  } else {
    pta_f64x4 acc[4][4];
    double a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      a[i] = 1.0 + (threadIdx.x + 64 * i) * 1e-9;
      b[i] = 1.0 - (threadIdx.x + 32 * i) * 1e-9;
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = pta_f64x4{0.0, 0.0, 0.0, 0.0};
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = pta_mfma_f64(a[i], b[j], acc[i][j]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  }
  if (s == 123.456) out[0] = s;
}

__global__ __launch_bounds__(256) void k_mb_fma(double *out, int iters) {
  double x[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) x[i] = 1.0 + i * 1e-3 + threadIdx.x * 1e-9;
  double a = 1.0000001, b = 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = fma(x[i], a, b);
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += x[i];
  if (s == 123.456) out[0] = s;
}

__global__ void k_mb_write(double2 *out, int64_t n2) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  double2 v = make_double2(1.0, 2.0);
  for (; i < n2; i += stride) out[i] = v;
}

// the STORE PATTERN of the walking assembly kernel without its arithmetic (kind 7): a wave owns a strip of `strip2` double2 columns (16 bytes
// each: 32 -> 512-byte row segments) of a [rows, ld2]-double2 matrix and walks down `seg_rows` rows, every store instruction writing
// 64 / strip2 whole row segments; the four waves of a workgroup own four neighbouring strips of the same rows.  What HBM takes when 2048
// waves write row segments 40 KB apart - the ceiling k_td_cov_walk's stores can reach, as opposed to the contiguous stream of k_mb_write.
__global__ __launch_bounds__(256) void k_mb_write_strips(double2 *__restrict__ out, int64_t ld2, int rows, int strip2, int nstrips, int seg_rows, int64_t blk2, int64_t nblk) {
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t gw = (int64_t)blockIdx.x * 4 + w;
  const int strip = (int)(gw % nstrips);
  const int nseg = (rows + seg_rows - 1) / seg_rows;
  const int seg = (int)((gw / nstrips) % nseg);
  const int64_t b = gw / ((int64_t)nstrips * nseg);
  if (b >= nblk) return;
  double2 *__restrict__ M = out + b * blk2;
  const double2 v = make_double2(1.0 + l, 2.0);
  const int r0 = seg * seg_rows, r1 = min(rows, r0 + seg_rows);
  if (strip2 <= 64) {
    const int rpi = 64 / strip2, lr = l / strip2, lc = l % strip2;
    for (int r = r0 + lr; r < r1; r += rpi) M[(int64_t)r * ld2 + (int64_t)strip * strip2 + lc] = v;
  } else {
    for (int r = r0; r < r1; ++r)
      for (int c = l; c < strip2; c += 64) M[(int64_t)r * ld2 + (int64_t)strip * strip2 + c] = v;
  }
}

__global__ void k_mb_copy(const double2 *__restrict__ in, double2 *__restrict__ out, int64_t n2) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n2; i += stride) out[i] = in[i];
}

__global__ __launch_bounds__(256) void k_mb_rng(double *out, int iters, int fast) {
  pta_rng_stage_tables();  // Box-Muller tables -> LDS (pta_rng.h)
  __syncthreads();
  double s = 0.0;
  uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  for (int it = 0; it < iters; ++it) {
    double z0, z1;
    if (fast == 2) {  // the polynomial (fdlibm-style) transform the table-driven default replaced: A/B of the two
      double u1, u2, sn, cs;
      pta_uniform_pair(pta_philox_draw(42, (uint64_t)it, pta_stream_id(PTA_STREAM_WN, 0), p), u1, u2);
      const double rad = pta_sqrt_pos(pta_neg2log_poly(u1));
      pta_sincos_2pi_poly(u2, sn, cs);
      z0 = rad * cs;
      z1 = rad * sn;
    } else {
      pta_normal_pair(42, (uint64_t)it, pta_stream_id(PTA_STREAM_WN, 0), p, z0, z1, fast);
    }
    s += z0 * z1;
  }
  if (s == 123.456) out[0] = s;
}

extern "C" int pta_microbench(int kind, int64_t bytes, int iters, int option, double *result_host) {
  PTA_REQUIRE(result_host && iters > 0, PTA_E_ARG, "pta_microbench: bad argument");
  int dev = 0;
  PTA_HIP(hipGetDevice(&dev));
  hipDeviceProp_t prop;
  PTA_HIP(hipGetDeviceProperties(&prop, dev));
  const int cus = prop.multiProcessorCount;
  // kinds 0/1/4: `bytes` in 1..32 selects the number of 256-thread blocks per CU (= waves per SIMD); default 8
  const int bpc = ((kind == 0 || kind == 1 || kind == 4 || kind == 5 || kind == 6 || kind == 8) && bytes >= 1 && bytes <= 32) ? (int)bytes : 8;
  // kind 7: `option` = bytes per row segment (256 .. 8192), `bytes` = total bytes written per launch (blocks of 5000 x 5008 doubles)
  const int s7_rows = 5000, s7_ld2 = 5008 / 2;
  const int s7_seg = option & 0xFFFF, s7_lds = (option >> 16) * 1024;  // option = bytes per row segment | (KB of dynamic LDS per workgroup << 16): 70 KB -> two workgroups per CU, as k_td_cov_walk
  const int s7_strip2 = kind == 7 ? (s7_seg >= 16 ? s7_seg / 16 : 32) : 32;
  const int s7_nstrips = s7_ld2 / s7_strip2;
  const int64_t s7_blk2 = (int64_t)s7_rows * s7_ld2;
  const int64_t s7_nblk = kind == 7 ? (bytes / (s7_blk2 * 16) > 0 ? bytes / (s7_blk2 * 16) : 1) : 0;
  hipEvent_t e0, e1;
  PTA_HIP(hipEventCreate(&e0));
  PTA_HIP(hipEventCreate(&e1));
  double *buf = nullptr, *buf2 = nullptr;
  int64_t nbytes = (kind == 2 || kind == 3) ? bytes : (kind == 7 ? s7_nblk * s7_blk2 * 16 : 4096);
  PTA_REQUIRE(nbytes >= 4096, PTA_E_ARG, "pta_microbench: bytes too small");
  PTA_HIP(hipMalloc(&buf, nbytes));
  if (kind == 3) PTA_HIP(hipMalloc(&buf2, nbytes));
  float ms = 0.f;
  double work = 0.0;
  const int reps = (kind == 2 || kind == 3 || kind == 7) ? iters : 3;
  for (int pass = 0; pass < 2; ++pass) {  // pass 0 = warm-up
    PTA_HIP(hipEventRecord(e0, 0));
    for (int rep = 0; rep < reps; ++rep) {
      switch (kind) {
        case 0:
          hipLaunchKernelGGL(k_mb_mfma, dim3(cus * bpc), dim3(256), 0, 0, buf, iters);
          work = (double)cus * bpc * 4 * iters * 8.0 * 2048.0 * reps;  // waves * mfma * flop
          break;
        case 5:
          hipLaunchKernelGGL(k_mb_mfma_tile, dim3(cus * bpc), dim3(256), 0, 0, buf, iters);
          work = (double)cus * bpc * 4 * iters * 16.0 * 2048.0 * reps;
          break;
        case 6:  // half the waves: 16 MFMAs x 2048 flop per iteration; the other half: 4 x 16 FMAs x 64 lanes x 2 flop per iteration
          hipLaunchKernelGGL(k_mb_mix, dim3(cus * bpc), dim3(256), 0, 0, buf, iters);
          work = (double)cus * bpc * 2 * iters * (16.0 * 2048.0 + 4 * 16.0 * 64.0 * 2.0) * reps;
          break;
        case 1:
          hipLaunchKernelGGL(k_mb_fma, dim3(cus * bpc), dim3(256), 0, 0, buf, iters);
          work = (double)cus * bpc * 256 * iters * 16.0 * 2.0 * reps;
          break;
        case 2:
          hipLaunchKernelGGL(k_mb_write, dim3(cus * 8), dim3(256), 0, 0, (double2 *)buf, nbytes / 16);
          work = (double)nbytes * reps;
          break;
        case 3:
          hipLaunchKernelGGL(k_mb_copy, dim3(cus * 8), dim3(256), 0, 0, (const double2 *)buf2, (double2 *)buf, nbytes / 16);
          work = 2.0 * (double)nbytes * reps;
          break;
        case 7: {
          const int nseg = (s7_rows + 511) / 512;
          const int64_t waves = s7_nblk * nseg * s7_nstrips;
          hipLaunchKernelGGL(k_mb_write_strips, dim3((unsigned)((waves + 3) / 4)), dim3(256), s7_lds, 0, (double2 *)buf, (int64_t)s7_ld2, s7_rows, s7_strip2,
                             s7_nstrips, 512, s7_blk2, s7_nblk);
          work = (double)s7_nblk * s7_rows * s7_nstrips * s7_strip2 * 16.0 * reps;
          break;
        }
        case 8: {
          const int na = option;
          if (na == 1) hipLaunchKernelGGL(k_mb_mfma_nacc<1>, dim3(cus * bpc), dim3(256), 0, 0, buf, iters);
          else if (na == 2) hipLaunchKernelGGL(k_mb_mfma_nacc<2>, dim3(cus * bpc), dim3(256), 0, 0, buf, iters);
          else if (na == 4) hipLaunchKernelGGL(k_mb_mfma_nacc<4>, dim3(cus * bpc), dim3(256), 0, 0, buf, iters);
          else if (na == 8) hipLaunchKernelGGL(k_mb_mfma_nacc<8>, dim3(cus * bpc), dim3(256), 0, 0, buf, iters);
          else if (na == 12) hipLaunchKernelGGL(k_mb_mfma_nacc<12>, dim3(cus * bpc), dim3(256), 0, 0, buf, iters);
          else hipLaunchKernelGGL(k_mb_mfma_nacc<16>, dim3(cus * bpc), dim3(256), 0, 0, buf, iters);
          work = (double)cus * bpc * 4 * iters * (double)(na == 1 || na == 2 || na == 4 || na == 8 || na == 12 ? na : 16) * 2048.0 * reps;
          break;
        }
        case 4:
          hipLaunchKernelGGL(k_mb_rng, dim3(cus * bpc), dim3(256), 0, 0, buf, iters, option);
          work = (double)cus * bpc * 256 * iters * 2.0 * reps;  // normals
          break;
        default:
          pta_set_error("pta_microbench: unknown kind %d", kind);
          return PTA_E_ARG;
      }
    }
    PTA_HIP(hipEventRecord(e1, 0));
    PTA_HIP(hipEventSynchronize(e1));
    PTA_HIP(hipEventElapsedTime(&ms, e0, e1));
  }
  PTA_LAUNCH_CHECK();
  (void)hipFree(buf);
  if (buf2) (void)hipFree(buf2);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *result_host = work / (ms * 1e-3) / 1e12;  // T(flop|byte|normal)/s
  return PTA_OK;
}

// ---- engine-clock probe ------------------------------------------------------------------------------------------
// ONE wave spins for `us` microseconds of constant-rate time and writes a pair (s_memrealtime, s_memtime) every `period_us`: s_memtime
// ticks once per shader cycle, s_memrealtime at the constant 100 MHz reference (MI355X_MICROARCH.md), so the slope between two samples
// IS the engine clock over that interval - measured on the chip, beside whatever the caller's other streams run.  (The per-dispatch
// quotient GRBM_GUI_ACTIVE / duration of round 3 folded launch gaps into the clock: it read 1.8-2.1 GHz for 3-ms dispatches that this
// probe shows running at 2.28-2.30.)  Asynchronous on `stream`; unused sample slots are zeroed.
__global__ __launch_bounds__(64) void k_clock_probe(uint64_t *__restrict__ out, int max_samples, int us, int period_us) {
  if (threadIdx.x != 0) return;
  const uint64_t t0 = wall_clock64();
  const uint64_t t_end = t0 + (uint64_t)us * 100u;
  uint64_t next = t0;
  int n = 0;
  while (n < max_samples) {
    const uint64_t rt = wall_clock64();
    if (rt >= next) {
      out[2 * n] = rt;
      out[2 * n + 1] = clock64();
      ++n;
      next += (uint64_t)period_us * 100u;
    }
    if (rt >= t_end) break;
    __builtin_amdgcn_s_sleep(8);
  }
  for (int i = n; i < max_samples; ++i) out[2 * i] = 0, out[2 * i + 1] = 0;
}

extern "C" int pta_clock_probe(uint64_t *samples, int max_samples, int us, int period_us, void *stream) {
  PTA_REQUIRE(samples && max_samples >= 2 && us > 0 && period_us > 0, PTA_E_ARG, "pta_clock_probe: bad argument");
  hipLaunchKernelGGL(k_clock_probe, dim3(1), dim3(64), 0, pta_stream(stream), samples, max_samples, us, period_us);
  PTA_LAUNCH_CHECK();
  return PTA_OK;
}
