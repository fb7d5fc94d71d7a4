// Blocked substitution of the workspace scheme (pta_potrf.hip: pta_ws_solve_phase) as ONE launch per panel: a workgroup keeps ONE 128-row
// tile of the rows below the panel and walks the panel's 128-column blocks left to right, X_j = [X_{<j} | B_j] S_j^T with K = 128 (j + 1)
// (S_j = [-W_jj L11[j, <j] | W_jj] from the workspace), storing X_j over B_j and reading it back as part of the next block's operand.
//
// Why: the per-block form is eight launches per panel of (row tiles x matrices) workgroups - 31 x 34 = 1054 on the chip's 512 slots, 2.06
// rounds that run as 3, each block paying the tile product's prologue / epilogue (worth 84 columns of K on a mean K of 576): 49 TFLOP/s
// where the same kernel does 67-69 on the block-column updates (profiles/r06_potrf_left_looking.txt).  Row tiles do not depend on each other,
// so nothing forces a device-wide barrier between the blocks: here a row tile's eight products run back to back in one workgroup.
//
// The one hazard is the workgroup reading back what it has just written: stores are write-through to the XCD's L2, but the CU's vector L1
// may still hold the lines of B_j it loaded as operand one block earlier.  So (i) the operand slabs are fetched with the sc1 cache policy
// (global_load_lds ... sc1: served from L2, never from L1 - MI355X_MICROARCH.md, "loads bypass L1 only"), and (ii) a block's stores are
// waited for (s_waitcnt vmcnt(0): stores count in vmcnt on gfx950) in front of the barrier that precedes the next block's first fetch.
// A workgroup only ever reads and writes its own rows (and the read-only workspace), so no other ordering is needed.
//
// Tile product = k_dgemm_glds128 (pta_gemm.hip): 128 x 128 tile, 4 waves as 2 x 2 of 64 x 64, operand slabs of 16 k by LDS DMA into
// XOR-swizzled unpadded rows, k slots permuted so that a lane's four values of a slab are two conflict-free ds_read_b128.
#include "pta_common.h"
#include "pta_mfma.h"

#define SR_T 128
#define SR_K 16
#define SR_ROWB 128
#define SR_SC1 16  // cache-policy bit of global_load_lds on gfx940+: sc1 (agent scope: the load is served from L2)

__device__ __forceinline__ int pta_sr_f(int row) {
  const int e = (row >> 1) & 7;
  return (e & 1) | (((e >> 2) & 1) * 6);
}

// X: rows below the panel at the panel's first column (row pitch lda, matrix stride sA), `rows` of them; W: the panel's strips (row pitch
// ldw, matrix stride sW), block j = rows [128 j, 128 j + wj) of W, K_j = oj + wj columns; blocks: j = 0 has f128 columns, the others 128.
__global__ __launch_bounds__(256, 2) void k_ws_solve_rows(double *__restrict__ X, int64_t lda, int64_t sA, int rows, int nb, int f128,
                                                          const double *__restrict__ W, int64_t ldw, int64_t sW) {
  X += (int64_t)blockIdx.z * sA;
  W += (int64_t)blockIdx.z * sW;
  __shared__ double __attribute__((aligned(256))) slab[2][2][SR_T * SR_K];
  const int t = threadIdx.x, l = t & 63, w = t >> 6;
  const int wm = w >> 1, wn = w & 1;
  const int m0 = blockIdx.x * SR_T;
  typedef double pta_f64x2 __attribute__((ext_vector_type(2)));
  int kc[4];
  const double *__restrict__ srcA[4];
  int rb[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = 32 * w + 8 * j + (l >> 3);
    kc[j] = 2 * ((l & 7) ^ pta_sr_f(row));
    srcA[j] = X + (int64_t)min(m0 + row, rows - 1) * lda;
    rb[j] = row;  // row of the strip (output column of the block); clamped per block below
  }
  const int fi = l & 15, fq = l >> 4;
  const int fsw = pta_sr_f(fi);
  const int offA = (wm * 64 + fi) * SR_ROWB, offB = (wn * 64 + fi) * SR_ROWB;
  const int c0 = ((2 * fq) ^ fsw) * 16, c1 = ((2 * fq + 1) ^ fsw) * 16;
  for (int jb = 0; jb < nb; ++jb) {
    const int oj = jb == 0 ? 0 : f128 + 128 * (jb - 1), wj = jb == 0 ? f128 : 128;
    const int K = oj + wj;
    const double *__restrict__ S = W + (int64_t)jb * 128 * ldw;
    auto stage = [&](int k0, int st) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = min(k0 + kc[j], K - 2);
        char *dA = reinterpret_cast<char *>(&slab[st][0][0]) + (32 * w + 8 * j) * SR_ROWB;
        char *dB = reinterpret_cast<char *>(&slab[st][1][0]) + (32 * w + 8 * j) * SR_ROWB;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(srcA[j] + k),
                                         (__attribute__((address_space(3))) void *)dA, 16, 0, SR_SC1);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(S + (int64_t)min(rb[j], wj - 1) * ldw + k),
                                         (__attribute__((address_space(3))) void *)dB, 16, 0, 0);
      }
    };
    pta_f64x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = pta_f64x4{0.0, 0.0, 0.0, 0.0};
    auto slab_product = [&](int cur, int kv, int next_k0) {
      const char *pa = reinterpret_cast<const char *>(&slab[cur][0][0]) + offA;
      const char *pb = reinterpret_cast<const char *>(&slab[cur][1][0]) + offB;
      pta_f64x2 a0[4], b0[4], a1[4], b1[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        a0[i] = *reinterpret_cast<const pta_f64x2 *>(pa + i * 16 * SR_ROWB + c0);
        b0[i] = *reinterpret_cast<const pta_f64x2 *>(pb + i * 16 * SR_ROWB + c0);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        a1[i] = *reinterpret_cast<const pta_f64x2 *>(pa + i * 16 * SR_ROWB + c1);
        b1[i] = *reinterpret_cast<const pta_f64x2 *>(pb + i * 16 * SR_ROWB + c1);
      }
      if (kv < SR_K) {  // K tail (the first panel's n mod 128 extra columns): slots past K were clamped duplicates
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
      if (next_k0 >= 0) stage(next_k0, cur ^ 1);
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
      __builtin_amdgcn_sched_barrier(0);
    };
    const int nfull = K / SR_K, ktail = K - nfull * SR_K, nslab = nfull + (ktail ? 1 : 0);
    stage(0, 0);
    __syncthreads();
    for (int sidx = 0; sidx < nfull; ++sidx) {
      slab_product(sidx & 1, SR_K, sidx + 1 < nslab ? (sidx + 1) * SR_K : -1);
      __syncthreads();
    }
    if (ktail) slab_product(nfull & 1, ktail, -1);
    // every wave has read the block's last slab (the loop's closing barrier, or - K tail - the products above use registers only): store X_j
    // over B_j, wait for the stores to reach L2, then let every wave pass before the next block's first fetch reads them back
    double *__restrict__ C = X + oj;
    const int colb = wn * 64 + pta_mfma_col(l);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rowb = m0 + wm * 64 + i * 16 + (l >> 4);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = rowb + 4 * r, col = colb + j * 16;
          if (row < rows && col < wj) C[(int64_t)row * lda + col] = acc[i][j][r];
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
}

int pta_ws_solve_rows_launch(double *X, int64_t lda, int64_t sA, int B, int rows, int nb, int f128, const double *W, int64_t ldw, int64_t sW,
                             hipStream_t stream) {
  PTA_REQUIRE(X && W, PTA_E_ARG, "pta_ws_solve_rows: NULL argument");
  PTA_REQUIRE(rows > 0 && nb > 0 && f128 >= 2 && f128 <= 128 && !(f128 & 1) && !(lda & 1) && !(sA & 1) && !(ldw & 1) && !(sW & 1) && ((uintptr_t)X % 16) == 0 &&
                  ((uintptr_t)W % 16) == 0 && B > 0 && B <= 65535,
              PTA_E_ARG, "pta_ws_solve_rows: rows=%d nb=%d f128=%d (even sizes / pitches, 16-byte aligned operands)", rows, nb, f128);
  hipLaunchKernelGGL(k_ws_solve_rows, dim3(pta_cdiv(rows, SR_T), 1, B), dim3(256), 0, stream, X, lda, sA, rows, nb, f128, W, ldw, sW);
  PTA_LAUNCH_CHECK();
  return PTA_OK;
}
