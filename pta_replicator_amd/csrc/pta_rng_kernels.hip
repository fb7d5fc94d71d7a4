// Throughput-mode draws: raw Philox access (known-answer tests) and the buffer fill that dumps the
// exact normals the fused kernels generate in-register.
#include "pta_common.h"
#include "pta_rng.h"

__global__ void k_philox_raw(const uint32_t *__restrict__ ctr, const uint32_t *__restrict__ key, int n,
                             uint32_t *__restrict__ out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  pta_u32x4 c = {ctr[4 * i + 0], ctr[4 * i + 1], ctr[4 * i + 2], ctr[4 * i + 3]};
  pta_u32x4 v = pta_philox4x32_10(c, key[0], key[1]);
  out[4 * i + 0] = v.x;
  out[4 * i + 1] = v.y;
  out[4 * i + 2] = v.z;
  out[4 * i + 3] = v.w;
}

extern "C" int pta_rng_philox_raw(const uint32_t *ctr, const uint32_t *key, int n, uint32_t *out, void *stream) {
  PTA_REQUIRE(ctr && key && out && n > 0, PTA_E_ARG, "pta_rng_philox_raw: bad argument");
  hipLaunchKernelGGL(k_philox_raw, dim3(pta_cdiv(n, 256)), dim3(256), 0, pta_stream(stream), ctr, key, n, out);
  PTA_LAUNCH_CHECK();
  return PTA_OK;
}

__global__ void k_fill_normal(uint64_t seed, uint64_t r0, uint32_t stream_id, int npairs, int interleave,
                              double *__restrict__ z0, double *__restrict__ z1, int64_t ld, int fast) {
  pta_rng_stage_tables();  // Box-Muller tables -> LDS (pta_rng.h)
  __syncthreads();
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  int r = blockIdx.y;
  if (p >= npairs) return;
  double a, b;
  pta_normal_pair(seed, r0 + (uint64_t)r, stream_id, (uint32_t)p, a, b, fast);
  if (interleave) {
    double2 v = make_double2(a, b);
    *reinterpret_cast<double2 *>(z0 + (int64_t)r * ld + 2 * (int64_t)p) = v;
  } else {
    z0[(int64_t)r * ld + p] = a;
    z1[(int64_t)r * ld + p] = b;
  }
}

extern "C" int pta_rng_fill_normal(uint64_t seed, uint64_t r0, int R, uint32_t stream_id, int npairs, int interleave,
                                   double *z0, double *z1, int64_t ld, int rng_fast, void *stream) {
  PTA_REQUIRE(z0 && (interleave || z1), PTA_E_ARG, "pta_rng_fill_normal: NULL output");
  PTA_REQUIRE(R > 0 && npairs > 0, PTA_E_ARG, "pta_rng_fill_normal: R=%d npairs=%d", R, npairs);
  PTA_REQUIRE(ld >= (interleave ? 2 * (int64_t)npairs : (int64_t)npairs), PTA_E_ARG, "pta_rng_fill_normal: ld too small");
  PTA_REQUIRE(!interleave || (ld % 2 == 0 && ((uintptr_t)z0 % 16) == 0), PTA_E_ARG,
              "pta_rng_fill_normal: interleaved output needs even ld and 16-byte alignment");
  hipLaunchKernelGGL(k_fill_normal, dim3(pta_cdiv(npairs, 256), R), dim3(256), 0, pta_stream(stream), seed, r0, stream_id,
                     npairs, interleave, z0, z1, ld, rng_fast ? 1 : 0);
  PTA_LAUNCH_CHECK();
  return PTA_OK;
}

// The same fill for ALL blocks of a TD plan in one launch: block b's deviates (stream (stream_kind, b), n = blk_n[b], pairs written
// whole) go to z[r * ld + blk_zoff[b] ..] - what k_td_trmm_rng / k_td_trmm_z128 read through pta_td_plan.z.  68 launches of the
// per-block form cost 1.1 ms per 1024 realisations of the 68 x 5000 array, this one 0.6 alone (1.6 beside the GWB grid stage; see FILL_ROWS).
// A workgroup draws the same 256 pairs of FILL_ROWS consecutive rows: the 2.5 KB table staging (a global read + a barrier) and the
// block's layout loads are paid once per 2048 pairs instead of once per 256 - with one row per workgroup the launch was 696 k workgroups
// of ~0.4 us of arithmetic behind ~1.5 us of load latency each: 1.59 ms per 1024 realisations of the 68 x 5000 array = 0.22 T normals/s
// against the 0.59 T/s of the in-library RNG microbenchmark, and it sits in front of every L.z product (round 6).
#define FILL_ROWS 8
__global__ void k_fill_normal_blocks(uint64_t seed, uint64_t r0, uint32_t stream_kind, const int32_t *__restrict__ blk_n,
                                     const int32_t *__restrict__ blk_zoff, double *__restrict__ z, int64_t ld, int fast, int R) {
  pta_rng_stage_tables();
  __syncthreads();
  const int b = blockIdx.z;
  const int npairs = (blk_n[b] + 1) >> 1;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= npairs) return;
  const uint32_t sid = pta_stream_id(stream_kind, (uint32_t)b);
  double *__restrict__ zp = z + blk_zoff[b] + 2 * (int64_t)p;
  const int rbeg = blockIdx.y * FILL_ROWS, rend = min(R, rbeg + FILL_ROWS);
  for (int r = rbeg; r < rend; ++r) {
    double a, c;
    pta_normal_pair(seed, r0 + (uint64_t)r, sid, (uint32_t)p, a, c, fast);
    *reinterpret_cast<double2 *>(zp + (int64_t)r * ld) = make_double2(a, c);
  }
}

extern "C" int pta_rng_fill_normal_blocks(uint64_t seed, uint64_t r0, int R, uint32_t stream_kind, int n_blocks, const int32_t *blk_n,
                                          const int32_t *blk_zoff, int max_n, double *z, int64_t ld, int rng_fast, void *stream) {
  PTA_REQUIRE(blk_n && blk_zoff && z, PTA_E_ARG, "pta_rng_fill_normal_blocks: NULL argument");
  PTA_REQUIRE(R > 0 && R <= 65535 && n_blocks > 0 && n_blocks <= 65535 && max_n > 0, PTA_E_ARG,
              "pta_rng_fill_normal_blocks: R=%d n_blocks=%d max_n=%d", R, n_blocks, max_n);
  PTA_REQUIRE(ld % 2 == 0 && ((uintptr_t)z % 16) == 0, PTA_E_ARG, "pta_rng_fill_normal_blocks: even ld and a 16-byte aligned z needed");
  hipLaunchKernelGGL(k_fill_normal_blocks, dim3(pta_cdiv((max_n + 1) / 2, 256), pta_cdiv(R, FILL_ROWS), n_blocks), dim3(256), 0, pta_stream(stream),
                     seed, r0, stream_kind, blk_n, blk_zoff, z, ld, rng_fast ? 1 : 0, R);
  PTA_LAUNCH_CHECK();
  return PTA_OK;
}
