// Reproducer of round 4's "NaN until a device-wide synchronisation" (VERDICT r4, weak #1): the column-walking assembly kernel of
// commit ee73ef5 (withdrawn in 18da960), VERBATIM, behind a one-function C ABI.  scripts/gpu_r5_nan_repro.py runs it on a design
// matrix that is a VIEW into a larger slab whose tail is NaN (or zero): with K = 58 (components = 29, the failing test case) the kernel's
// A-fragment loads of the cut k-step read rows k = 58, 59 - behind the [K, N] matrix - and multiply them by b = 0: 0 x NaN = NaN.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -shared -fPIC -I pta_replicator_amd/csrc \
//         -o scripts/probe_src/libnan_repro_r4_walk.so scripts/probe_src/nan_repro_r4_walk.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "pta_mfma.h"

#define TCW_SEG 512
template <int NKS>
__global__ __launch_bounds__(256, 2) void k_td_cov_walk_r4(const double *__restrict__ Ft, int64_t ldf, int K, const double *__restrict__ phi,
                                                        const double *__restrict__ sigma2, const int32_t *__restrict__ epoch_of,
                                                        const double *__restrict__ ecorr2, double *__restrict__ Cbase,
                                                        const int64_t *__restrict__ blk_pos, const int32_t *__restrict__ blk_ld,
                                                        const int32_t *__restrict__ blk_n, const int32_t *__restrict__ blk_off) {
  const int blk = blockIdx.y;
  const int N = blk_n[blk];
  int item = blockIdx.x, cs = 0;
  const int nstrip = (N + 127) >> 7;
  for (; cs < nstrip; ++cs) {  // item -> (strip, segment); workgroup-uniform
    const int ns = (N - 128 * cs + TCW_SEG - 1) / TCW_SEG;
    if (item < ns) break;
    item -= ns;
  }
  if (cs >= nstrip) return;
  const int t = threadIdx.x, l = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6), li = l & 15, lq = l >> 4;
  const int c0 = 128 * cs + 32 * w;  // this wave's 32 columns
  if (c0 >= N) return;
  const int rbeg = 128 * cs + TCW_SEG * item, rend = min(N, rbeg + TCW_SEG);
  const int64_t off = blk_off[blk];
  const int64_t ldc = blk_ld[blk];
  double *__restrict__ C = Cbase + blk_pos[blk];
  const double *__restrict__ F = Ft + off;
  const double *__restrict__ ph = phi + (int64_t)blk * K;
  // resident B operand: lane holds B[k = 4 ks + lq][j = li] = phi_k F[k, c0 + 16 jt + li]; k >= K enters as zero
  double b[2][NKS];
  int ccol[2], ecol[2];
#pragma unroll
  for (int jt = 0; jt < 2; ++jt) {
    const int col = c0 + 16 * jt + li;
    ccol[jt] = min(col, N - 1);
    ecol[jt] = (epoch_of && col < N) ? epoch_of[off + col] : -1;
  }
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    const int k = 4 * ks + lq, kc = min(k, K - 1);
    const double p = k < K ? ph[kc] : 0.0;
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) b[jt][ks] = p * F[(int64_t)kc * ldf + ccol[jt]];
  }
  const int64_t kbase = (int64_t)lq * ldf;  // this lane's k row inside a k-step
  for (int r0 = max(rbeg, c0); r0 < rend; r0 += 32) {  // wave-uniform; rows above the wave's own columns are not in the lower triangle
    int rowc[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) rowc[it] = min(r0 + 16 * it + li, N - 1);
    double a[2][NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const double *__restrict__ Fk = F + kbase + (int64_t)min(4 * ks, K - 1 - lq < 0 ? 0 : 4 * ks) * ldf;  // (k clamped below; a masked k meets b = 0)
#pragma unroll
      for (int it = 0; it < 2; ++it) a[it][ks] = Fk[rowc[it]];
    }
    // the epilogue's per-row operands, requested beside the fragments: epoch, ECORR variance and (diagonal steps) white variance
    const bool diag_step = r0 < c0 + 32;  // wave-uniform
    int erow[2][4];
    double e2[2][4], s2[2][4];
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = min(r0 + 16 * it + pta_mfma_row(l, r), N - 1);
        erow[it][r] = epoch_of ? epoch_of[off + row] : -2;
        e2[it][r] = epoch_of ? ecorr2[off + row] : 0.0;
        s2[it][r] = diag_step ? sigma2[off + row] : 0.0;
      }
    pta_f64x4 acc[2][2];
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
      for (int jt = 0; jt < 2; ++jt) acc[it][jt] = pta_f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
      for (int it = 0; it < 2; ++it)
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) acc[it][jt] = pta_mfma_f64(a[it][ks], b[jt][ks], acc[it][jt]);
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
      for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = r0 + 16 * it + pta_mfma_row(l, r), col = c0 + 16 * jt + li;
          double v = acc[it][jt][r];
          if (diag_step && row == col) v = v + s2[it][r];
          if (erow[it][r] == ecol[jt] && col <= row) v = v + e2[it][r];
          if (row < N && col <= row) C[(int64_t)row * ldc + col] = v;
        }
  }
}

extern "C" int nan_repro_r4_walk(const double *Ft, int64_t ldf, int K, const double *phi, const double *sigma2, const int32_t *epoch_of,
                                 const double *ecorr2, double *Cbase, const int64_t *blk_pos, const int32_t *blk_ld, const int32_t *blk_n,
                                 const int32_t *blk_off, int n_blocks, int max_n, void *stream) {
  int64_t items = 0;
  for (int cs = 0; 128 * cs < max_n; ++cs) items += (max_n - 128 * cs + TCW_SEG - 1) / TCW_SEG;
  if (K <= 60)
    hipLaunchKernelGGL(k_td_cov_walk_r4<15>, dim3((unsigned)items, n_blocks), dim3(256), 0, (hipStream_t)stream, Ft, ldf, K, phi, sigma2, epoch_of,
                       ecorr2, Cbase, blk_pos, blk_ld, blk_n, blk_off);
  else
    hipLaunchKernelGGL(k_td_cov_walk_r4<16>, dim3((unsigned)items, n_blocks), dim3(256), 0, (hipStream_t)stream, Ft, ldf, K, phi, sigma2, epoch_of,
                       ecorr2, Cbase, blk_pos, blk_ld, blk_n, blk_off);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
