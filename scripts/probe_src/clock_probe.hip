// Engine-clock probe ("why do the VALU-bound kernels run at 1.8-2.1 GHz?").
// One wave spins for `us` microseconds of constant-rate time and writes pairs (s_memrealtime, s_memtime) every `period_us`:
// s_memtime ticks once per shader cycle, s_memrealtime at the constant 100 MHz reference (MI355X_MICROARCH.md), so the slope
// between two samples IS the engine clock over that interval - measured on the chip, beside whatever kernel the other streams
// run, at 10-us resolution (rocm-smi samples at ~4 Hz and per-dispatch counters fold launch gaps into the quotient).
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o scripts/probe_src/libclock_probe.so scripts/probe_src/clock_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ __launch_bounds__(64) void k_clock_probe(uint64_t *__restrict__ out, int max_samples, int us, int period_us) {
  if (threadIdx.x != 0) return;
  const uint64_t t0 = wall_clock64();
  const uint64_t t_end = t0 + (uint64_t)us * 100u;
  uint64_t next = t0;
  int n = 0;
  while (n < max_samples) {
    const uint64_t rt = wall_clock64();
    if (rt >= next) {
      const uint64_t sc = clock64();
      out[2 * n] = rt;
      out[2 * n + 1] = sc;
      ++n;
      next += (uint64_t)period_us * 100u;
    }
    if (rt >= t_end) break;
    __builtin_amdgcn_s_sleep(8);
  }
  for (int i = n; i < max_samples; ++i) out[2 * i] = 0, out[2 * i + 1] = 0;
}

extern "C" int clock_probe_launch(void *out_dev, int max_samples, int us, int period_us, void *stream) {
  hipLaunchKernelGGL(k_clock_probe, dim3(1), dim3(64), 0, (hipStream_t)stream, (uint64_t *)out_dev, max_samples, us, period_us);
  return (int)hipGetLastError();
}

// ---- control loads, launched on the caller's stream (pta_microbench frees its buffer = a device-wide synchronisation that would wait
// for the probe): 0 = 16 independent v_fma_f64 chains per lane, 1 = v_mad_u64_u32 chains (Philox's multiplier), 2 = fp64 MFMA 4 x 4
// register tile, 3 = streaming 16-byte stores over `n16` double2
typedef double f64x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_load_fma(double *out, int iters) {
  double a[16];
  for (int i = 0; i < 16; ++i) a[i] = 1.0 + 1e-3 * (threadIdx.x + i);
  const double m = 1.0000001, c = 1e-9;
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = fma(a[i], m, c);
  double s = 0;
  for (int i = 0; i < 16; ++i) s += a[i];
  if (s == 123.456) out[0] = s;
}
__global__ __launch_bounds__(256) void k_load_imad(double *out, int iters) {
  uint64_t a[8];
  for (int i = 0; i < 8; ++i) a[i] = 0x9E3779B97F4A7C15ull * (threadIdx.x + i + 1);
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = (uint64_t)(uint32_t)a[i] * 0xD2511F53u + (a[i] >> 32);   // v_mad_u64_u32
  uint64_t s = 0;
  for (int i = 0; i < 8; ++i) s ^= a[i];
  if (s == 42) out[0] = 1.0;
}
__global__ __launch_bounds__(256) void k_load_mfma(double *out, int iters) {
  f64x4 acc[4][4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) acc[i][j] = f64x4{0.0, 0.0, 0.0, 0.0};
  double a[4], b[4];
  for (int i = 0; i < 4; ++i) a[i] = 1e-3 * (threadIdx.x + i), b[i] = 1e-3 * (threadIdx.x - i);
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
  double s = 0;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][3];
  if (s == 123.456) out[0] = s;
}
__global__ __launch_bounds__(256) void k_load_write(double2 *out, int64_t n16) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) out[i] = make_double2(1.0, 2.0);
}
extern "C" int clock_probe_load(int kind, void *buf, int64_t n16, int iters, int blocks, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  if (kind == 0) hipLaunchKernelGGL(k_load_fma, dim3(blocks), dim3(256), 0, s, (double *)buf, iters);
  else if (kind == 1) hipLaunchKernelGGL(k_load_imad, dim3(blocks), dim3(256), 0, s, (double *)buf, iters);
  else if (kind == 2) hipLaunchKernelGGL(k_load_mfma, dim3(blocks), dim3(256), 0, s, (double *)buf, iters);
  else hipLaunchKernelGGL(k_load_write, dim3(blocks), dim3(256), 0, s, (double2 *)buf, n16);
  return (int)hipGetLastError();
}
