// TD mode, uniform batches, left-looking factorisation: the covariance ASSEMBLY fused into the first (and only) touch of every tile.
//
// In the left-looking panel order (pta_potrf.hip: pta_potrf_chain_ws_left) a tile of the lower triangle is written exactly once before
// its panel is factored: by the update of its block column, C <- C - L[rows, 0:k0] L[cols, 0:k0]^T.  C itself is the covariance the
// reference's synthesis implies, C = F diag(phi) F^T + diag(sigma^2) + sum_e ecorr_e^2 1_e 1_e^T (red_noise.py:98-101,126-128,
// white_noise.py:105-109,182) - a K = 60 product and two index terms.  So instead of writing C (6.8 GB at 68 x 5000^2,
// k_td_cov_walk) and reading it back in the update, the update computes it: the tile product's K loop starts with four slabs of the
// design-matrix rows (Fr x Gr^T, Gr = -phi F: the same MFMA stream, 64 more columns of K on a K >= 1032 product) and its epilogue adds
// the diagonal / ECORR terms to -acc and STORES - no C read, no assembly launch, no 13.6 GB round trip.
// Block column 0 (nothing to its left) is the K = 0 case: the four F slabs only.
//
// Kernel = k_dgemm_glds128 (pta_gemm.hip) without its ragged / C-prefetch forms: 128 x 128 tile per workgroup, 4 waves as 2 x 2 of 64 x 64,
// operand slabs of 16 k by LDS DMA (global_load_lds_dwordx4) into XOR-swizzled unpadded rows, the k slots of an MFMA step permuted so that a
// lane's four values of a slab are two conflict-free ds_read_b128; trapezoid tile grid dealt evenly to the XCDs.
#include "pta_common.h"
#include "pta_mfma.h"

#define FZ_T 128
#define FZ_K 16
#define FZ_ROWB 128  // bytes per LDS row (16 doubles)
#define FZ_FW 64     // columns of the row-major design-matrix operands (K <= 64, zero padded)

__device__ __forceinline__ int pta_fz_f(int row) {
  const int e = (row >> 1) & 7;
  return (e & 1) | (((e >> 2) & 1) * 6);
}

// block column [r0, r0 + N) x rows [r0, r0 + M) of every matrix of the batch (M >= N: the N x N diagonal block on top of a rectangle):
//   C[m, c] = sum_{k < kf} Fr[m, k] phi_k Fr[c, k] - sum_{k < K} L[m, k] L[c, k] + [m == c] sigma2[m] + [epoch[m] == epoch[c]] ecorr2[m],  c <= m
// (indices relative to r0; K = r0 = the columns to the left, all final).
__global__ __launch_bounds__(256, 2) void k_td_fused_update(int M, int N, int K, double *__restrict__ Lb, int64_t ld, int64_t sL, int r0, pta_fuse fz) {
  int bm, bn;
  {
    const int tix = blockIdx.x;
    const int nt = (N + FZ_T - 1) / FZ_T, tri = nt * (nt + 1) / 2;
    if (tix < tri) {
      bm = (int)((sqrt(8.0 * (double)tix + 1.0) - 1.0) * 0.5);
      while ((bm + 1) * (bm + 2) / 2 <= tix) ++bm;
      while (bm * (bm + 1) / 2 > tix) --bm;
      bn = tix - bm * (bm + 1) / 2;
    } else {
      bm = nt + (tix - tri) / nt;
      bn = (tix - tri) % nt;
    }
  }
  double *__restrict__ Lm = Lb + (int64_t)blockIdx.z * sL;
  const double *__restrict__ A = Lm + (int64_t)r0 * ld;  // row m of the block column: A + m ld (its columns [0, K) are final)
  double *__restrict__ C = Lm + (int64_t)r0 * ld + r0;
  const int64_t toa0 = (int64_t)blockIdx.z * fz.toa_stride + r0;  // global TOA index of the block column's first row / column
  const double *__restrict__ Fa = fz.Fr + toa0 * FZ_FW;
  const double *__restrict__ Ga = fz.Gr + toa0 * FZ_FW;
  __shared__ double __attribute__((aligned(256))) slab[2][2][FZ_T * FZ_K];
  const int t = threadIdx.x, l = t & 63, w = t >> 6;
  const int wm = w >> 1, wn = w & 1;
  const int m0 = bm * FZ_T, n0 = bn * FZ_T;
  typedef double pta_f64x2 __attribute__((ext_vector_type(2)));
  // ---- DMA side: wave w stages rows [32 w, 32 w + 32) of both operands, 8 rows per instruction
  int ra[4], rb[4], kc[4];
  const double *__restrict__ srcA[4];
  const double *__restrict__ srcB[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = 32 * w + 8 * j + (l >> 3);
    kc[j] = 2 * ((l & 7) ^ pta_fz_f(row));
    ra[j] = min(m0 + row, M - 1);
    rb[j] = min(n0 + row, N - 1);
    srcA[j] = A + (int64_t)ra[j] * ld;
    srcB[j] = A + (int64_t)rb[j] * ld;  // the column operand is rows [r0 + n0 ..) of the same matrix
  }
  auto stageF = [&](int v, int st) {  // slab v < 4 of the design-matrix phase: columns 16 v .. 16 v + 15 of Fr / Gr
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      char *dA = reinterpret_cast<char *>(&slab[st][0][0]) + (32 * w + 8 * j) * FZ_ROWB;
      char *dB = reinterpret_cast<char *>(&slab[st][1][0]) + (32 * w + 8 * j) * FZ_ROWB;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(Fa + (int64_t)ra[j] * FZ_FW + 16 * v + kc[j]),
                                       (__attribute__((address_space(3))) void *)dA, 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(Ga + (int64_t)rb[j] * FZ_FW + 16 * v + kc[j]),
                                       (__attribute__((address_space(3))) void *)dB, 16, 0, 0);
    }
  };
  auto stageL = [&](int k0, int st) {  // slab of the factor phase: columns k0 .. k0 + 15 of L (clamped into [0, K - 2]; a tail is masked below)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = min(k0 + kc[j], K - 2);
      char *dA = reinterpret_cast<char *>(&slab[st][0][0]) + (32 * w + 8 * j) * FZ_ROWB;
      char *dB = reinterpret_cast<char *>(&slab[st][1][0]) + (32 * w + 8 * j) * FZ_ROWB;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(srcA[j] + k),
                                       (__attribute__((address_space(3))) void *)dA, 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(srcB[j] + k),
                                       (__attribute__((address_space(3))) void *)dB, 16, 0, 0);
    }
  };
  // ---- fragment side (as k_dgemm_glds128)
  const int fi = l & 15, fq = l >> 4;
  const int fsw = pta_fz_f(fi);
  const int offA = (wm * 64 + fi) * FZ_ROWB, offB = (wn * 64 + fi) * FZ_ROWB;
  const int c0 = ((2 * fq) ^ fsw) * 16, c1 = ((2 * fq + 1) ^ fsw) * 16;
  pta_f64x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = pta_f64x4{0.0, 0.0, 0.0, 0.0};
  // one slab: 16 fragment reads, 64 MFMAs.  Both phases add into the SAME accumulators: the design-matrix operand Gr carries -phi, so
  // acc = L L^T - F phi F^T and the epilogue stores -acc.  `next` issues the DMA of the following slab behind the first 16 products.
  auto slab_product = [&](int cur, int kv, auto next) {
    const char *pa = reinterpret_cast<const char *>(&slab[cur][0][0]) + offA;
    const char *pb = reinterpret_cast<const char *>(&slab[cur][1][0]) + offB;
    pta_f64x2 a0[4], b0[4], a1[4], b1[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      a0[i] = *reinterpret_cast<const pta_f64x2 *>(pa + i * 16 * FZ_ROWB + c0);
      b0[i] = *reinterpret_cast<const pta_f64x2 *>(pb + i * 16 * FZ_ROWB + c0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      a1[i] = *reinterpret_cast<const pta_f64x2 *>(pa + i * 16 * FZ_ROWB + c1);
      b1[i] = *reinterpret_cast<const pta_f64x2 *>(pb + i * 16 * FZ_ROWB + c1);
    }
    if (kv < FZ_K) {  // K tail of the factor phase: slots past K were clamped duplicates
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (4 * fq + 0 >= kv) a0[i].x = 0.0, b0[i].x = 0.0;
        if (4 * fq + 1 >= kv) a0[i].y = 0.0, b0[i].y = 0.0;
        if (4 * fq + 2 >= kv) a1[i].x = 0.0, b1[i].x = 0.0;
        if (4 * fq + 3 >= kv) a1[i].y = 0.0, b1[i].y = 0.0;
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = pta_mfma_f64(a0[i].x, b0[j].x, acc[i][j]);
    next();
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = pta_mfma_f64(a0[i].y, b0[j].y, acc[i][j]);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = pta_mfma_f64(a1[i].x, b1[j].x, acc[i][j]);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = pta_mfma_f64(a1[i].y, b1[j].y, acc[i][j]);
    __builtin_amdgcn_sched_barrier(0);  // the closing barrier's vmcnt(0) stays behind the 64 MFMAs
  };
  const int nf = fz.kf > 0 ? FZ_FW / FZ_K : 0;  // slabs of the design-matrix phase (0: no red noise)
  const int nfull = K / FZ_K, ktail = K - nfull * FZ_K, nslab = nfull + (ktail ? 1 : 0);
  // virtual slab v: v < nf = design-matrix slab v (stage v & 1), else factor slab v - nf (stage v & 1)
  if (nf > 0) stageF(0, 0);
  else if (nslab > 0) stageL(0, 0);
  __syncthreads();
  for (int v = 0; v < nf; ++v) {
    slab_product(v & 1, FZ_K, [&] {
      if (v + 1 < nf) stageF(v + 1, (v + 1) & 1);
      else if (nslab > 0) stageL(0, (v + 1) & 1);
    });
    __syncthreads();
  }
  for (int sidx = 0; sidx < nfull; ++sidx) {
    const int v = nf + sidx;
    slab_product(v & 1, FZ_K, [&] {
      if (sidx + 1 < nslab) stageL((sidx + 1) * FZ_K, (v + 1) & 1);
    });
    __syncthreads();
  }
  if (ktail) slab_product((nf + nfull) & 1, ktail, [] {});
  // ---- epilogue: C = -acc + [row == col] sigma2[row] + [epoch[row] == epoch[col]] ecorr2[row]; no C read
  const int colb = n0 + wn * 64 + pta_mfma_col(l);
  const bool ep = fz.epoch != nullptr;
  int ecol[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) ecol[j] = ep ? fz.epoch[toa0 + min(colb + 16 * j, N - 1)] : -1;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int rowb = m0 + wm * 64 + i * 16 + (l >> 4);
    int erow[4];
    double e2[4], s2[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t g = toa0 + min(rowb + 4 * r, M - 1);
      erow[r] = ep ? fz.epoch[g] : -2;
      e2[r] = ep ? fz.ecorr2[g] : 0.0;
      s2[r] = fz.sigma2[g];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = rowb + 4 * r, col = colb + 16 * j;
        double v = -acc[i][j][r];
        if (erow[r] == ecol[j]) v += e2[r];
        if (row == col) v += s2[r];
        if (row < M && col < N && col <= row) C[(int64_t)row * ld + col] = v;
      }
  }
}

// host launcher (internal: pta_potrf.hip's left-looking chain calls it in place of the C -= L L^T product when assembly operands are given)
int pta_td_fused_launch(int M, int N, int K, double *L, int64_t ld, int64_t sL, int r0, int batch, const pta_fuse &fz, hipStream_t stream) {
  PTA_REQUIRE(L && fz.Fr && fz.Gr && fz.sigma2 && (!fz.epoch || fz.ecorr2), PTA_E_ARG, "pta_td_fused_launch: NULL argument");
  PTA_REQUIRE(M >= N && N > 0 && K >= 0 && batch > 0 && batch <= 65535 && !(K & 1) && !(ld & 1) && !(sL & 1) && !(r0 & 1) && ((uintptr_t)L % 16) == 0 &&
                  ((uintptr_t)fz.Fr % 16) == 0 && ((uintptr_t)fz.Gr % 16) == 0 && K == r0,
              PTA_E_ARG, "pta_td_fused_launch: M=%d N=%d K=%d r0=%d (even sizes / pitches, 16-byte aligned operands, K == r0)", M, N, K, r0);
  const unsigned nt = pta_cdiv(N, FZ_T), mt = pta_cdiv(M, FZ_T);
  hipLaunchKernelGGL(k_td_fused_update, dim3(nt * (nt + 1) / 2 + (mt - nt) * nt, 1, batch), dim3(256), 0, stream, M, N, K, L, ld, sL, r0, fz);
  PTA_LAUNCH_CHECK();
  return PTA_OK;
}
