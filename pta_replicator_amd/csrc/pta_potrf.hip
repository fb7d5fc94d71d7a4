// Batched blocked fp64 Cholesky, lower, row-major, in place: np.linalg.cholesky of the ORF (red_noise.py:235) and the N_toa x N_toa
// covariances of TD mode.  Entry points: pta_potrf_batched / _ex / _ws (uniform batches), pta_potrf_ragged (+ _plan; matrices of
// different orders as one end-aligned schedule), pta_potrf_workspace_doubles, pta_potrf_warmup.  The tile products are pta_gemm.hip's.
// (Moved out of pta_orf_kernels.hip in round 6, unchanged.)
#include <stdlib.h>
#include "pta_common.h"
#include "pta_mfma.h"

// ---- blocked Cholesky ---------------------------------------------------------------------------
// Right-looking, block size 64, row-major, lower.  Per block column:
//   k_potf2 : the 64x64 diagonal block is factored in LDS by one workgroup per matrix
//   k_trsm  : the panel below it is solved against L11^T, 64 rows per workgroup, L11 and the tile in LDS
//   SYRK    : the trailing submatrix update A22 -= L21 L21^T runs on the fp64 MFMA GEMM (pta_gemm.hip)
#define CH_NB 64
#define CH_LD 65

// Factor the diagonal block AND invert the factor in the same 64-step sweep, the 64 x 64 tile held in REGISTERS (16 elements
// per thread: rows ti + 16 a, columns tc + 16 b).  Tile layout: L below the diagonal, the pivot d_j on it (sqrt(d_j) goes to
// D), and X^T above it, X = L^{-1} (forward substitution in its right-looking form: once row j of X is final,
// X[i][:] -= L[i][j] X[j][:] for i > j).  With col[x] = r * tile(x, j), r = 1 / sqrt(d_j), step j is, for every position
// (p, q) with q > j:
//     p >= q  (Schur complement of L)  or  p < j  (X[q][p])  :  tile(p, q) -= col[p] * col[q]
//     p == j                           (X[q][j] = -L[q][j] / L[j][j]) :  tile(j, q)  = -col[q] * r
// and column j itself is scaled by r.  The owners of column j publish it (unscaled) through a double-buffered LDS vector:
// ONE barrier per step, no serial section.  The tile leaves as the MFMA panel solve wants it (L below, X^T above; X's
// diagonal 1 / L[j][j] is recomputed by the consumer).
// The 64-step sweep of k_potf2 on a tile held in the registers of 256 threads (v[a][b] = element (ti + 16 a, tc + 16 b); lower
// triangle + diagonal loaded, zeros above); colbuf = the workgroup's double-buffered pivot column.  Shared with k_diag128.
__device__ __forceinline__ void pta_potf2_sweep(double (&v)[4][4], double (*colbuf)[CH_NB], int nb, int ti, int tc, int32_t *info_b, int col0) {
  int bad = 0;  // first pivot that is not positive (LAPACK's info), the same value in every thread; stored once, after the sweep
  const bool diag_ge = ti >= tc;  // p >= q inside a diagonal 16 x 16 sub-block (a == b)
#pragma unroll
  for (int jq = 0; jq < 4; ++jq) {
    for (int jr = 0; jr < 16; ++jr) {
      const int j = 16 * jq + jr;
      if (j >= nb) break;  // uniform
      double *cb = colbuf[j & 1];
      if (tc == jr) {
#pragma unroll
        for (int a = 0; a < 4; ++a) cb[ti + 16 * a] = v[a][jq];
      }
      __syncthreads();
      const double d = cb[j];
      double r = __builtin_amdgcn_rsq(d);  // 1/sqrt(d): hardware seed + two Newton steps
      r = r * fma(-0.5 * d * r, r, 1.5);
      r = r * fma(-0.5 * d * r, r, 1.5);
      double cp[4], cq[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) cp[a] = cb[ti + 16 * a] * r;
#pragma unroll
      for (int b = 0; b < 4; ++b) cq[b] = cb[tc + 16 * b] * r;
      bad = (!(d > 0.0) && bad == 0) ? col0 + j + 1 : bad;
      // the step as selects (no branch, no store): with a, b, jq compile-time most of the conditions fold away - what is left per step
      // are four comparisons of (ti, tc) with jr and the selects of the sub-blocks in row / column jq
      const bool c_gt = tc > jr, c_eq = tc == jr, r_lt = ti < jr, r_eq = ti == jr;
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          if (b < jq) continue;                                           // q < j: finished columns
          const bool qg = b > jq || c_gt;                                 // q > j
          const bool qe = b == jq && c_eq;                                // q == j
          const bool pl = a < jq || (a == jq && r_lt);                    // p < j
          const bool pe = a == jq && r_eq;                                // p == j
          const bool pgq = a > b || (a == b && diag_ge);                  // p >= q
          double nv = (qg && (pgq || pl)) ? fma(-cp[a], cq[b], v[a][b]) : v[a][b];   // Schur complement of L / X[q][p]
          nv = (qg && pe) ? -cq[b] * r : nv;                              // X[q][j] = -L[q][j] / L[j][j]
          nv = qe ? (pe ? d * r : cp[a]) : nv;                            // column j scaled; L[j][j] = sqrt(d)
          v[a][b] = nv;
        }
    }
  }
  if (bad && ti == 0 && tc == 0 && *info_b == 0) *info_b = bad;
}

__global__ __launch_bounds__(256) void k_potf2(double *__restrict__ A, int64_t n, int64_t sA, int k0, int nb, int32_t *__restrict__ info) {
  __shared__ double colbuf[2][CH_NB];
  double *M = A + (int64_t)blockIdx.x * sA + (int64_t)k0 * n + k0;  // n = row pitch (lda)
  const int t = threadIdx.x, ti = t >> 4, tc = t & 15;
  double v[4][4];
  // all 16 loads first, unconditional, from clamped addresses; the selects follow (predicated, or consumed one by one, each load
  // is waited for on its own: 16 serial round trips at the head of a kernel that runs one workgroup per matrix)
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int pc = min(ti + 16 * a, nb - 1);
      v[a][b] = M[(int64_t)pc * n + min(tc + 16 * b, pc)];
    }
  asm volatile("" ::: "memory");  // keep the loads together: nothing below may be scheduled between them
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int p = ti + 16 * a, q = tc + 16 * b;
      v[a][b] = (p < nb && q <= p) ? v[a][b] : 0.0;
    }
  pta_potf2_sweep(v, colbuf, nb, ti, tc, info + blockIdx.x, k0);
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int p = ti + 16 * a, q = tc + 16 * b;
      if (p < nb && q < nb) M[(int64_t)p * n + q] = v[a][b];
    }
}

__global__ __launch_bounds__(256) void k_trsm(double *__restrict__ A, int nrow, int64_t n, int64_t sA, int k0, int nb) {
  __shared__ double L[CH_NB][CH_LD];
  __shared__ double X[CH_NB][CH_LD];
  double *M = A + (int64_t)blockIdx.y * sA;  // n = row pitch (lda), nrow = order of the matrix
  const int t = threadIdx.x;
  const int r0 = k0 + nb + blockIdx.x * CH_NB;
  const int rows = min(CH_NB, nrow - r0);
  for (int i = t >> 6; i < nb; i += 4)
    for (int c = t & 63; c < nb; c += 64) L[i][c] = M[(int64_t)(k0 + i) * n + (k0 + c)];
  for (int i = t >> 6; i < rows; i += 4)
    for (int c = t & 63; c < nb; c += 64) X[i][c] = M[(int64_t)(r0 + i) * n + (k0 + c)];
  // X <- X L^{-T}, column sweep.  Four threads per row: lane quartet q takes columns c = j+1+q, j+5+q, ...
  const int row = t >> 2, q = t & 3;
  for (int j = 0; j < nb; ++j) {
    __syncthreads();
    if (q == 0 && row < rows) X[row][j] = X[row][j] / L[j][j];
    __syncthreads();
    if (row < rows) {
      const double xj = X[row][j];
      for (int c = j + 1 + q; c < nb; c += 4) X[row][c] = fma(-xj, L[c][j], X[row][c]);
    }
  }
  __syncthreads();
  for (int i = t >> 6; i < rows; i += 4)
    for (int c = t & 63; c < nb; c += 64) M[(int64_t)(r0 + i) * n + (k0 + c)] = X[i][c];
}

// Panel solve on the matrix cores: X <- X L11^{-T} = X . Linv^T, K = 64, one 16-row block of X per wave and step.
// The sum over k is order independent, so the four k-slots of a v_mfma_f64_16x16x4 step t are given the columns k = 16 q + t
// (q = lane >> 4) instead of 4 t + q: a lane's sixteen A elements are then 128 CONTIGUOUS bytes of its own row - X goes from
// global memory straight into registers (eight 16-byte loads), needs no LDS and no barrier, and the wave writes its 16 x 64
// result back in place before it moves on.  The inverted diagonal block is staged once per workgroup (transposed into LDS as
// k_potf2 parked it) and its B fragments - Linv[16 j + n][16 q + t] - are kept in registers for every block the wave walks.
__global__ __launch_bounds__(256) void k_trsm_mfma(double *__restrict__ A, int nrow, int64_t n, int64_t sA, int k0, int nb) {
  __shared__ double Li[CH_NB][CH_LD];  // Li[c][t] = (L11^{-1})[c][t], zero for t > c
  double *M = A + (int64_t)blockIdx.y * sA;
  const int t = threadIdx.x, l = t & 63, w = t >> 6;
  for (int i = t >> 6; i < CH_NB; i += 4)
    for (int c = t & 63; c < CH_NB; c += 64) {
      // tile element (i, c) of the factored diagonal block, read along its row (coalesced): above the diagonal it is
      // (L11^{-1})[c][i] (parked transposed by k_potf2) and goes to Li[c][i]; on it, 1 / L[i][i]; below, Li[c][i] = 0
      double v = 0.0;
      if (i < nb && c < nb) {
        const double m = M[(int64_t)(k0 + i) * n + (k0 + c)];
        if (c > i) v = m;
        else if (c == i) v = 1.0 / m;
      }
      Li[c][i] = v;
    }
  __syncthreads();
  const int q = l >> 4, c = l & 15;
  double bq[4][16];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int s = 0; s < 16; ++s) bq[j][s] = Li[j * 16 + c][16 * q + s];
  typedef double f64x2_a8 __attribute__((ext_vector_type(2), aligned(8)));  // k0 may be odd: 8-byte alignment only
  const int r0 = k0 + nb;
  const int nblk = (nrow - r0 + 15) >> 4;
  // every 16-byte load of a row stays inside the block's nb columns (nb < 64 only for the first, narrow block)
  const int kq = 16 * q;
  for (int blk = blockIdx.x * 4 + w; blk < nblk; blk += gridDim.x * 4) {
    const int row = min(r0 + blk * 16 + c, nrow - 1);  // rows past the end recompute the last one; their results are not stored
    const double *__restrict__ xr = M + (int64_t)row * n + k0;
    double a[16];
    if (nb >= 2) {  // workgroup-uniform.  A pair that would straddle the block's last column (odd nb) is read one column early
#pragma unroll
      for (int h = 0; h < 8; ++h) {
        const int k = kq + 2 * h;
        const f64x2_a8 v = *reinterpret_cast<const f64x2_a8 *>(xr + min(k, nb - 2));
        a[2 * h] = (k + 1 < nb) ? v.x : ((k < nb) ? v.y : 0.0);
        a[2 * h + 1] = (k + 1 < nb) ? v.y : 0.0;
      }
    } else {
      const double x0 = xr[0];
#pragma unroll
      for (int s = 0; s < 16; ++s) a[s] = 0.0;
      a[0] = (kq == 0) ? x0 : 0.0;
    }
    pta_f64x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = pta_f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < 16; ++s)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = pta_mfma_f64(a[s], bq[j][s], acc[j]);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int orow = r0 + blk * 16 + pta_mfma_row(l, r), ocol = j * 16 + pta_mfma_col(l);
        if (orow < nrow && ocol < nb) M[(int64_t)orow * n + (k0 + ocol)] = acc[j][r];
      }
  }
}

// The K = 64 update inside a 128-column group, A22[:, 0:64] -= L21 . L21[0:64, :]^T (lower triangle), with the operand scheme of
// k_trsm_mfma: both operands are rows of the panel just solved, 128 contiguous bytes per lane, straight from global memory into
// MFMA registers - no LDS at all.  The multiplier rows (the 64 rows right below the diagonal block) stay in registers for every
// 16-row block the wave walks; C is read 16 elements at a time from clamped addresses, only the stores are predicated.
__global__ __launch_bounds__(256) void k_syrk64(double *__restrict__ A, int nrow, int64_t n, int64_t sA, int c0) {
  double *M = A + (int64_t)blockIdx.y * sA;
  const int t = threadIdx.x, l = t & 63, w = t >> 6;
  const int q = l >> 4, c = l & 15;
  typedef double f64x2_a8 __attribute__((ext_vector_type(2), aligned(8)));
  const int r0 = c0 + CH_NB;           // first row below the diagonal block = first row AND first column of A22
  double bq[4][16];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int br = r0 + 16 * j + c;    // multiplier row; past the end of the matrix it contributes nothing that is stored
    const double *__restrict__ p = M + (int64_t)min(br, nrow - 1) * n + c0 + 16 * q;
#pragma unroll
    for (int h = 0; h < 8; ++h) {
      const f64x2_a8 v = *reinterpret_cast<const f64x2_a8 *>(p + 2 * h);
      bq[j][2 * h] = v.x;
      bq[j][2 * h + 1] = v.y;
    }
  }
  const int nblk = (nrow - r0 + 15) >> 4;
  for (int blk = blockIdx.x * 4 + w; blk < nblk; blk += gridDim.x * 4) {
    const double *__restrict__ xr = M + (int64_t)min(r0 + blk * 16 + c, nrow - 1) * n + c0 + 16 * q;
    double a[16];
#pragma unroll
    for (int h = 0; h < 8; ++h) {
      const f64x2_a8 v = *reinterpret_cast<const f64x2_a8 *>(xr + 2 * h);
      a[2 * h] = v.x;
      a[2 * h + 1] = v.y;
    }
    const int orow0 = r0 + blk * 16 + (l >> 4), ocol0 = r0 + c;  // + 4 r, + 16 j
    double cv[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) cv[j][r] = M[(int64_t)min(orow0 + 4 * r, nrow - 1) * n + min(ocol0 + 16 * j, nrow - 1)];
    pta_f64x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = pta_f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < 16; ++s)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = pta_mfma_f64(a[s], bq[j][s], acc[j]);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int orow = orow0 + 4 * r, ocol = ocol0 + 16 * j;
        if (orow < nrow && ocol <= orow) M[(int64_t)orow * n + ocol] = cv[j][r] - acc[j][r];
      }
  }
}

__global__ void k_zero_upper(double *__restrict__ A, int n, int64_t lda, int64_t sA) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  int r = blockIdx.y;
  if (c < n && c > r) A[(int64_t)blockIdx.z * sA + (int64_t)r * lda + c] = 0.0;
}


// Internal streams + events of the chained schedule below, created on first use, one context per (calling thread, device): a
// process that alternates devices keeps every device's handles (a single context recreated - and leaked - its four
// streams and five events whenever the current device changed).
#define PTA_POTRF_MAX_CHAINS 4
#define PTA_POTRF_MAX_DEVICES 16
struct pta_potrf_ctx {
  bool base = false;                    // ev_in exists
  bool made[PTA_POTRF_MAX_CHAINS] = {false, false, false, false};  // streams and events of chain i exist
  hipStream_t chain[PTA_POTRF_MAX_CHAINS] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev_in = nullptr, ev_out[PTA_POTRF_MAX_CHAINS] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev_diag[PTA_POTRF_MAX_CHAINS] = {nullptr, nullptr, nullptr, nullptr};  // chain c's first diagonal phase is done
  hipStream_t side[PTA_POTRF_MAX_CHAINS] = {nullptr, nullptr, nullptr, nullptr};   // look-ahead stream of chain c (workspace scheme)
  hipEvent_t ev_u1[PTA_POTRF_MAX_CHAINS] = {nullptr, nullptr, nullptr, nullptr};   // the next panel's diagonal block has been updated
  hipEvent_t ev_la[PTA_POTRF_MAX_CHAINS] = {nullptr, nullptr, nullptr, nullptr};   // the next panel's diagonal phase (look-ahead) is done
};
static thread_local pta_potrf_ctx g_potrf_ctx[PTA_POTRF_MAX_DEVICES];

// the context of (calling thread, current device) with the streams / events of chains 0 .. nchain - 1 in place.  Created ON DEMAND, chain
// by chain: a HIP stream is a hardware queue (~10 ms each to create on this stack - profiles/r05_prepare_td_first_call.txt), the default
// schedule uses two chains + their two look-ahead streams, and round 4 created all eight at the first call whatever it needed.
static int pta_potrf_ctx_get(pta_potrf_ctx **out, int nchain = PTA_POTRF_MAX_CHAINS) {
  int dev = 0;
  PTA_HIP(hipGetDevice(&dev));
  PTA_REQUIRE(dev >= 0 && dev < PTA_POTRF_MAX_DEVICES, PTA_E_ARG, "pta_potrf_batched: device ordinal %d beyond %d", dev, PTA_POTRF_MAX_DEVICES);
  pta_potrf_ctx &c = g_potrf_ctx[dev];
  if (!c.base) {
    PTA_HIP(hipEventCreateWithFlags(&c.ev_in, hipEventDisableTiming));
    c.base = true;
  }
  for (int i = 0; i < nchain && i < PTA_POTRF_MAX_CHAINS; ++i) {
    if (c.made[i]) continue;
    // created into locals and committed to the context only when all six exist: a failure half way destroys what it made instead of
    // leaving handles the next call would overwrite
    hipStream_t st[2] = {nullptr, nullptr};
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    hipError_t e = hipSuccess;
    // st[0] = the chain's stream, st[1] = its look-ahead stream.  PTA_POTRF_SIDE_PRIO=1 in the environment creates the look-ahead stream at
    // the device's highest queue priority (A/B of round 6, measured and NOT the default: a diagonal phase dispatched ahead of the tile
    // products' workgroups still runs several times slower beside them and now delays them too - 68 x 5000^2 right-looking 52.6 against
    // 52.3 ms, 16 x 10 000^2 98.0 against 94.1, left-looking with run-ahead diagonal phases 54.6 against 51.4: profiles/r06_potrf_left_looking.txt)
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);   // (least, greatest): numerically lower = higher priority
    const char *pe = getenv("PTA_POTRF_SIDE_PRIO");
    const bool side_hi = pe && pe[0] == '1';
    e = hipStreamCreateWithFlags(&st[0], hipStreamNonBlocking);
    if (e == hipSuccess) e = side_hi ? hipStreamCreateWithPriority(&st[1], hipStreamNonBlocking, prio_hi) : hipStreamCreateWithFlags(&st[1], hipStreamNonBlocking);
    for (int k = 0; k < 4 && e == hipSuccess; ++k) e = hipEventCreateWithFlags(&ev[k], hipEventDisableTiming);
    if (e != hipSuccess) {
      for (int k = 0; k < 2; ++k)
        if (st[k]) (void)hipStreamDestroy(st[k]);
      for (int k = 0; k < 4; ++k)
        if (ev[k]) (void)hipEventDestroy(ev[k]);
      pta_set_error("pta_potrf: creating the streams / events of chain %d failed: %s", i, hipGetErrorString(e));
      return PTA_E_HIP;
    }
    c.chain[i] = st[0], c.side[i] = st[1];
    c.ev_out[i] = ev[0], c.ev_diag[i] = ev[1], c.ev_u1[i] = ev[2], c.ev_la[i] = ev[3];
    c.made[i] = true;
  }
  *out = &c;
  return PTA_OK;
}

// creates the internal streams / events of the factorisation's default schedule now instead of inside the first factorisation (ABI 7)
extern "C" int pta_potrf_warmup(int nchain) {
  pta_potrf_ctx *cx = nullptr;
  return pta_potrf_ctx_get(&cx, nchain <= 0 ? 2 : (nchain > PTA_POTRF_MAX_CHAINS ? PTA_POTRF_MAX_CHAINS : nchain));
}

// Factor columns [c0, c0 + w) of every matrix of the batch for ALL rows below them, all updates from columns < c0 already
// applied: recursive halving.  The right half of a panel is updated with K = the left half's width in ONE product (at the top
// levels that is K = 512 / 256, where the MFMA GEMM runs at 47-50 / 36-40 TFLOP/s) instead of 64 columns at a time (K = 64: 16).
// Base case (<= 64 columns): diagonal block factored AND inverted in registers (k_potf2), rows below solved by an MFMA product
// with the parked inverse (k_trsm_mfma) or, on request, by forward substitution (k_trsm).
static int pta_factor_panel(double *A, int n, int64_t lda, int64_t sA, int B, int c0, int w, int32_t *info, int flags, int algo,
                            hipStream_t sp) {
  if (w <= CH_NB) {
    hipLaunchKernelGGL(k_potf2, dim3(B), dim3(256), 0, sp, A, lda, sA, c0, w, info);
    PTA_LAUNCH_CHECK();
    const int rows = n - c0 - w;
    if (rows <= 0) return PTA_OK;
    if (algo && !(flags & PTA_POTRF_SUBSTITUTION)) {
      // about 1024 workgroups per launch (2 resident per CU x 2 rounds), each wave walking its share of the 16-row blocks
      const int nblk = pta_cdiv(rows, CH_NB), per = pta_cdiv(1024, B);
      hipLaunchKernelGGL(k_trsm_mfma, dim3(nblk < per ? nblk : per, B), dim3(256), 0, sp, A, n, lda, sA, c0, w);
    }
    else
      hipLaunchKernelGGL(k_trsm, dim3(pta_cdiv(rows, CH_NB), B), dim3(256), 0, sp, A, n, lda, sA, c0, w);
    PTA_LAUNCH_CHECK();
    return PTA_OK;
  }
  // right part: a multiple of 64 (about half); the LEFT part takes the remainder, so that an odd width (the first panel's
  // n mod 128 extra columns) ends up in the very first base block and every later column boundary - hence every later
  // update's row count - stays aligned to the 64 / 128-wide tiles
  int cols = (w / 2 / CH_NB) * CH_NB;
  if (cols < CH_NB) cols = CH_NB;
  const int w1 = w - cols;
  int rc = pta_factor_panel(A, n, lda, sA, B, c0, w1, info, flags, algo, sp);
  if (rc != PTA_OK) return rc;
  const int rows = n - (c0 + w1);
  const double *L21 = A + (int64_t)(c0 + w1) * lda + c0;
  double *A22 = A + (int64_t)(c0 + w1) * lda + (c0 + w1);
  if (algo && w1 == CH_NB && cols == CH_NB) {  // the K = 64 update of a 128-column group: register-operand kernel, no LDS
    const int per = pta_cdiv(1024, B), nb64 = pta_cdiv(rows, CH_NB);
    hipLaunchKernelGGL(k_syrk64, dim3(nb64 < per ? nb64 : per, B), dim3(256), 0, sp, A, n, lda, sA, c0);
    PTA_LAUNCH_CHECK();
  } else {
    rc = pta_dgemm_launch(1, rows, cols, w1, -1.0, L21, lda, 1, L21, lda, 1.0, A22, lda, 1, B, sA, sA, sA, algo, sp);
    if (rc != PTA_OK) return rc;
  }
  return pta_factor_panel(A, n, lda, sA, B, c0 + w1, cols, info, flags, algo, sp);
}

// One step of a dependency chain (right-looking over panels of NBO columns, every launch on `s`): factor the panel that
// starts at column k0, then apply it to everything to its right in ONE product.  Returns the next panel's first column in *k0_io.
static int pta_potrf_step(double *A, int n, int64_t lda, int64_t strideA, int B, int32_t *info, int flags, int algo, int NBO, int *k0_io,
                          hipStream_t s) {
  // the FIRST panel also takes n mod 128 columns, so that every trailing update covers a multiple of 128 rows: whole 128 x 128
  // tiles only (at n = 5000: 1032 + 1024 + ... instead of 31.06, 23.06, ... tiles per side)
  const int k0 = *k0_io;
  const int want = (k0 == 0 && n > NBO) ? NBO + (n % 128) : NBO;
  const int nbo = (n - k0 < want) ? (n - k0) : want;
  const int pend = k0 + nbo;  // one past the panel's last column
  int rc = pta_factor_panel(A, n, lda, strideA, B, k0, nbo, info, flags, algo, s);
  if (rc != PTA_OK) return rc;
  *k0_io = pend;
  const int rows = n - pend;
  if (rows <= 0) return PTA_OK;
  const double *L21 = A + (int64_t)pend * lda + k0;
  double *A22 = A + (int64_t)pend * lda + pend;
  return pta_dgemm_launch(1, rows, rows, nbo, -1.0, L21, lda, 1, L21, lda, 1.0, A22, lda, 1, B, strideA, strideA, strideA, algo, s);
}

// ---- panel solve on the rows BELOW the panel only, through 128 x 128 inverses of the diagonal blocks (pta_potrf_batched_ws) ----
// pta_factor_panel above applies its recursion to the FULL height of a panel: every 64-column solve, K = 64 update and K = 128 /
// 256 / 512 product touches all rows below - ~1200 dispatches per 68-matrix batch, most of them latency chains.  With a workspace
// the same panel becomes
//   (1) the recursion on the nbo x nbo DIAGONAL block only (rows = nbo: a fifth of the work at n = 5000);
//   (2) W_jj = (L11's 128 x 128 diagonal block j)^-1 for all j at once, from the 64 x 64 inverses k_potf2 parks (k_inv_blocks);
//   (3) blocked substitution over the 128-column blocks, left to right, on ALL rows below.  Block j is X_j = (B_j - X_{<j} L11[j, <j]^T)
//       W_jj^T = B_j W_jj^T - X_{<j} T_j^T with T_j = W_jj L11[j, <j] (128 x 128 j, a small product per matrix on the panel's own rows):
//       the finished blocks X_{<j} and B_j are CONTIGUOUS columns of the rows below, so with the strip S_j = [-T_j | W_jj] kept in the
//       workspace the block is ONE tile product X_j = [X_{<j} | B_j] S_j^T, K = 128 (j + 1) - 8 launches per panel instead of 15, none
//       of them the K = 128 product that pays a full tile prologue and store for 128 columns of work (40 TFLOP/s against 57-61 at K >=
//       512; in place: one column tile per launch reads its columns before it writes them);
//   (4) the trailing update as before (K = nbo).
// cond(L11's diagonal blocks) * eps enters X, as it already does through the 64 x 64 inverses: the TD covariances have cond(L) ~
// 1e2-1e4, their factors agree with LAPACK to 1e-10 (tests); ill-conditioned inputs take PTA_POTRF_SUBSTITUTION (workspace ignored).

// one workgroup per (128-column block, matrix): W block = [[X1, 0], [-X2 L21 X1, X2]] from the two 64 x 64 (first block: narrower)
// diagonal tiles as k_potf2 left them - L below the diagonal, X^T = L^{-T} above it, X's diagonal = 1 / L's.  Tiles are read along
// their rows (coalesced) and transposed on the way into LDS.
__global__ __launch_bounds__(256) void k_inv_blocks(const double *__restrict__ A, int64_t lda, int64_t sA, int k0, int nbo, int f128,
                                                    double *__restrict__ W, int64_t ldw, int64_t sW) {
  __shared__ double X1[64][65], X2[64][65], L21[64][65], T[64][65];
  const int blk = blockIdx.x;
  const int o = blk == 0 ? 0 : f128 + 128 * (blk - 1);     // offset of the block inside the panel
  const int wd = blk == 0 ? f128 : 128;                     // its width
  const int w1 = wd > 64 ? wd - 64 : wd, w2 = wd - w1;      // base blocks inside it (w2 = 64 or 0)
  const double *M = A + (int64_t)blockIdx.y * sA + (int64_t)(k0 + o) * lda + (k0 + o);
  double *Wb = W + (int64_t)blockIdx.y * sW + (int64_t)blk * 128 * ldw + o;  // strip j = rows [128 j, 128 j + 128) of the workspace: W_jj at its columns [o, o + 128)
  const int t = threadIdx.x;
  for (int idx = t; idx < 64 * 64; idx += 256) {
    const int r = idx >> 6, c = idx & 63;  // tile element (r, c), c fastest: coalesced
    double x1 = 0.0, x2 = 0.0, l = 0.0;
    if (r < w1 && c < w1 && c >= r) x1 = M[(int64_t)r * lda + c];
    if (w2) {
      const double *M2 = M + (int64_t)w1 * lda + w1;
      if (c >= r) x2 = M2[(int64_t)r * lda + c];
      if (c < w1) l = M[(int64_t)(w1 + r) * lda + c];
    }
    // element (r, c) above the diagonal is X[c][r]; on it, 1 / L[r][r]
    X1[c][r] = (r < w1 && c < w1 && c >= r) ? (c == r ? 1.0 / x1 : x1) : 0.0;
    X2[c][r] = (w2 && c >= r) ? (c == r ? 1.0 / x2 : x2) : 0.0;
    L21[r][c] = l;
  }
  __syncthreads();
  // the off-diagonal block -X2 (L21 X1): two 64 x 64 x 64 products on the matrix cores, a wave per 16 rows, operands from LDS
  const int l = t & 63, wv = t >> 6, li = l & 15, lq = l >> 4;
  pta_f64x4 acc[4];
  if (w2) {
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) acc[cb] = pta_f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
    for (int k = 0; k < 64; k += 4) {
      const double a = L21[16 * wv + li][k + lq];
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) acc[cb] = pta_mfma_f64(a, X1[k + lq][16 * cb + li], acc[cb]);
    }
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) T[16 * wv + pta_mfma_row(l, r)][16 * cb + pta_mfma_col(l)] = acc[cb][r];
    __syncthreads();
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) acc[cb] = pta_f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
    for (int k = 0; k < 64; k += 4) {
      const double a = X2[16 * wv + li][k + lq];
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) acc[cb] = pta_mfma_f64(a, T[k + lq][16 * cb + li], acc[cb]);
    }
  }
  for (int idx = t; idx < 128 * 128; idx += 256) {
    const int r = idx >> 7, c = idx & 127;
    double v = 0.0;
    if (r < wd && c < wd) {
      if (r < w1) v = c < w1 ? X1[r][c] : 0.0;
      else if (c >= w1) v = X2[r - w1][c - w1];
      else continue;                // the product block: written from the accumulators below
    }
    Wb[(int64_t)r * ldw + c] = v;   // the whole 128 x 128 slot is written (zeros outside the block): the products read K = 128 of it
  }
  if (w2) {
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * wv + pta_mfma_row(l, r), col = 16 * cb + pta_mfma_col(l);
        if (col < w1) Wb[(int64_t)(w1 + row) * ldw + col] = -acc[cb][r];
      }
  }
}

// The base case of the workspace scheme's recursion: ONE workgroup per matrix factors a whole 128-column diagonal block (wd <= 128
// columns at (c0, c0): a first part of w1 = wd - 64 (or wd) and a second of 64) AND writes its inverse W = [[X1, 0], [-X2 L21 X1, X2]]
// into the block's 128 x 128 workspace slot - what took k_potf2 + k_trsm_mfma + k_syrk64 + k_potf2 (four dependent launches over
// the rows of the block) and a share of k_inv_blocks: the two 64-step sweeps stay (registers of 256 threads, one barrier per pivot),
// what lies between them - L21 = A21 X1^T, A22 -= L21 L21^T, the inverse's off-diagonal block - is 64 x 64 x 64 products on the
// matrix cores out of LDS.  The tiles leave as k_potf2 leaves them (L below the diagonal, X^T parked above it).
// RAG (ragged batch, end-aligned virtual coordinates - pta_common.h: pta_rag): the block starts at max(c0, front[b]); a matrix whose
// front lies behind the block leaves at once, one whose front lies inside it factors the part it has and parks the inverse in the
// bottom-right corner of the slot (the K / column masks of the ragged products never look at the rest); `idx` maps the chain position to
// the caller's matrix index (info) and pivots are reported in the matrix's own numbering.
template <bool RAG>
__global__ __launch_bounds__(256) void k_diag128(double *__restrict__ A, int64_t lda, int64_t sA, int c0, int wd, double *__restrict__ W,
                                                 int64_t ldw, int64_t sW, int32_t *__restrict__ info, pta_rag rg, const int64_t *__restrict__ idx) {
  __shared__ double T11[64][65], T21[64][65], T22[64][65], TT[64][65];
  __shared__ double colbuf[2][CH_NB];
  double *M, *Wb;
  int32_t *info_b;
  int col0;
  if (RAG) {
    const int f = (int)rg.front[blockIdx.x], cs = max(c0, f);
    wd = c0 + wd - cs;
    if (wd <= 0) return;
    lda = rg.ld[blockIdx.x];
    M = A + rg.off[blockIdx.x] + (int64_t)cs * lda + cs;
    Wb = W + (int64_t)blockIdx.x * sW + (int64_t)(cs - c0) * ldw + (cs - c0);  // the part the matrix has: bottom-right corner of the slot
    info_b = info + idx[blockIdx.x];
    col0 = cs - f;
  } else {
    M = A + (int64_t)blockIdx.x * sA + (int64_t)c0 * lda + c0;
    Wb = W + (int64_t)blockIdx.x * sW;  // the block's slot (the caller passes W already offset to strip j, column oj)
    info_b = info + blockIdx.x;
    col0 = c0;
  }
  const int w1 = wd > 64 ? wd - 64 : wd, w2 = wd - w1;
  const int t = threadIdx.x, ti = t >> 4, tc = t & 15;
  const int l = t & 63, wv = t >> 6, li = l & 15, lq = l >> 4;
  double v[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int pc = min(ti + 16 * a, w1 - 1);
      v[a][b] = M[(int64_t)pc * lda + min(tc + 16 * b, pc)];
    }
  if (w2) {  // the other two tiles: ALL their loads requested behind the first one's (unconditional, clamped), then into LDS along their rows
    const double *M2 = M + (int64_t)w1 * lda;
    const int c = t & 63, r0 = t >> 6;
    double x21[16], x22[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int r = r0 + 4 * i;
      x21[i] = M2[(int64_t)r * lda + min(c, w1 - 1)];
      x22[i] = M2[(int64_t)r * lda + w1 + min(c, r)];
    }
    asm volatile("" ::: "memory");
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int r = r0 + 4 * i;
      T21[r][c] = c < w1 ? x21[i] : 0.0;
      T22[r][c] = c <= r ? x22[i] : 0.0;
    }
  }
  asm volatile("" ::: "memory");
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int p = ti + 16 * a, q = tc + 16 * b;
      v[a][b] = (p < w1 && q <= p) ? v[a][b] : 0.0;
    }
  pta_potf2_sweep(v, colbuf, w1, ti, tc, info_b, col0);
  // tile 1 -> global as it is (L, X1^T parked above the diagonal) and -> LDS as XS[k][c] = X1[c][k]: the part above the diagonal
  // as it is, 1 / L on the diagonal, zero below
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int p = ti + 16 * a, q = tc + 16 * b;
      if (p < w1 && q < w1) M[(int64_t)p * lda + q] = v[a][b];
      T11[p][q] = (p < w1 && q < w1) ? (q > p ? v[a][b] : (q == p ? 1.0 / v[a][b] : 0.0)) : 0.0;
    }
  __syncthreads();
  pta_f64x4 acc[4];
  if (w2) {
    // L21 = A21 X1^T: a wave per 16 rows, in place (a wave reads and writes its own rows only)
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) acc[cb] = pta_f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
    for (int k = 0; k < 64; k += 4) {
      const double a = T21[16 * wv + li][k + lq];
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) acc[cb] = pta_mfma_f64(a, T11[k + lq][16 * cb + li], acc[cb]);
    }
    __syncthreads();  // (every lane of every wave has its A21 operands before rows are overwritten)
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) T21[16 * wv + pta_mfma_row(l, r)][16 * cb + pta_mfma_col(l)] = acc[cb][r];
    __syncthreads();
    // A22 -= L21 L21^T (the lower triangle is what the second sweep reads)
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) acc[cb] = pta_f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
    for (int k = 0; k < 64; k += 4) {
      const double a = T21[16 * wv + li][k + lq];
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) acc[cb] = pta_mfma_f64(a, T21[16 * cb + li][k + lq], acc[cb]);
    }
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) T22[16 * wv + pta_mfma_row(l, r)][16 * cb + pta_mfma_col(l)] -= acc[cb][r];
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int p = ti + 16 * a, q = tc + 16 * b;
        v[a][b] = q <= p ? T22[p][q] : 0.0;
      }
    __syncthreads();  // colbuf: the first sweep's last column has been read by everybody
    pta_potf2_sweep(v, colbuf, 64, ti, tc, info_b, col0 + w1);
    double *M2 = M + (int64_t)w1 * lda;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int p = ti + 16 * a, q = tc + 16 * b;
        M2[(int64_t)p * lda + w1 + q] = v[a][b];
        T22[p][q] = q > p ? v[a][b] : (q == p ? 1.0 / v[a][b] : 0.0);  // XS2[k][c] = X2[c][k]
      }
    for (int idx = t; idx < 64 * 64; idx += 256) {  // L21 -> global (coalesced)
      const int r = idx >> 6, c = idx & 63;
      if (c < w1) M2[(int64_t)r * lda + c] = T21[r][c];
    }
    __syncthreads();
    // TT = L21 X1 (X1[k][c] = T11[c][k]), then W21 = -X2 TT (X2[r][k] = T22[k][r])
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) acc[cb] = pta_f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
    for (int k = 0; k < 64; k += 4) {
      const double a = T21[16 * wv + li][k + lq];
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) acc[cb] = pta_mfma_f64(a, T11[16 * cb + li][k + lq], acc[cb]);
    }
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) TT[16 * wv + pta_mfma_row(l, r)][16 * cb + pta_mfma_col(l)] = acc[cb][r];
    __syncthreads();
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) acc[cb] = pta_f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
    for (int k = 0; k < 64; k += 4) {
      const double a = T22[k + lq][16 * wv + li];
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) acc[cb] = pta_mfma_f64(a, TT[k + lq][16 * cb + li], acc[cb]);
    }
  }
  // the inverse -> the block's whole 128 x 128 workspace slot (zeros outside the block: the products read K = 128 of it); the
  // off-diagonal product block comes from the accumulators below
  {
    const int c = t & 127, rh = t >> 7;
#pragma unroll 8
    for (int i = 0; i < 64; ++i) {
      const int r = rh + 2 * i;
      double x = 0.0;
      if (r < w1)
        x = c < w1 ? T11[c][r] : 0.0;
      else if (r < wd && c >= w1 && c < wd)
        x = T22[c - w1][r - w1];
      if (!(w2 && r >= w1 && r < wd && c < w1) && (!RAG || (r < wd && c < wd))) Wb[(int64_t)r * ldw + c] = x;
    }
  }
  if (w2) {
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * wv + pta_mfma_row(l, r), col = 16 * cb + pta_mfma_col(l);
        if (col < w1) Wb[(int64_t)(w1 + row) * ldw + col] = -acc[cb][r];
      }
  }
}

// -T_j = -W_jj L11[j, <j] for every block j >= 1 of a panel in ONE launch (the strips' left parts): one workgroup per (group of four
// 64-column chunks of a strip, matrix).  The 128 x 64 chunk of L11 goes through LDS (coalesced 16-byte loads, the next chunk in flight
// in registers while this one is multiplied).  The k slot of lane group q at step t = 2 u + s is m = 8 u + 2 q + s, so a lane's A
// elements are pairs of neighbours in its row of W_jj (16-byte loads, read ONCE and kept in registers for the chunks the workgroup
// walks).  W_jj is lower triangular: the 16-row block b needs m < 16 (b + 1) only, i.e. its first 4 (b + 1) steps; a wave takes the
// blocks w and 7 - w (36 of the 64 block-steps, the same for every wave).
#define PTA_WS_STRIP_GROUP 4
// RAG: a matrix whose (panel-relative) front fp lies at or behind the block has no left part and leaves; chunks wholly below fp are
// skipped, the chunk that straddles it loads from clamped columns and stores only columns >= fp.
template <bool RAG>
__global__ __launch_bounds__(256) void k_ws_strips(const double *__restrict__ A, int64_t lda, int64_t sA, int k0, int f128, double *__restrict__ W,
                                                   int64_t ldw, int64_t sW, pta_rag rg) {
  __shared__ double Bs[128][64];
  int g = blockIdx.x, j = 1, oj = f128;
  for (;;) {  // group -> (block j, group inside its chunks)
    const int ng = (((oj + 63) >> 6) + PTA_WS_STRIP_GROUP - 1) / PTA_WS_STRIP_GROUP;
    if (g < ng) break;
    g -= ng;
    ++j;
    oj += 128;
  }
  int fp = 0;
  if (RAG) {
    fp = max(0, (int)rg.front[blockIdx.y] - k0);
    if (fp >= oj || 64 * min((oj + 63) >> 6, (g + 1) * PTA_WS_STRIP_GROUP) <= fp) return;  // workgroup-uniform
    lda = rg.ld[blockIdx.y];
    A += rg.off[blockIdx.y];
  } else {
    A += (int64_t)blockIdx.y * sA;
  }
  double *S = W + (int64_t)blockIdx.y * sW + (int64_t)j * 128 * ldw;
  const int t = threadIdx.x, l = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6), c = l & 15, q = l >> 4;
  typedef double f64x2_a8 __attribute__((ext_vector_type(2), aligned(8)));  // k0, oj may be odd: 8-byte alignment only
  typedef double f64x2 __attribute__((ext_vector_type(2)));
  const int nch = (oj + 63) >> 6, ch0 = g * PTA_WS_STRIP_GROUP, ch1 = min(nch, ch0 + PTA_WS_STRIP_GROUP);
  // chunk rows oj .. oj + 127 of L11, columns c0 .. c0 + 63 (columns past oj are the block's own: valid memory, products not stored)
  const double *Lrow = A + (int64_t)(k0 + oj + (t >> 5)) * lda + k0;
  const int lc = 2 * (t & 31);  // column of this thread's pair inside a chunk
  f64x2_a8 pre[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) pre[i] = *reinterpret_cast<const f64x2_a8 *>(Lrow + (int64_t)(8 * i) * lda + (RAG ? max(64 * ch0 + lc, fp) : 64 * ch0 + lc));
  const int blo = w, bhi = 7 - w;                      // the wave's two 16-row blocks
  const int ulo = 2 * (blo + 1), uhi = 2 * (bhi + 1);  // 8-column groups of W_jj they reach into
  const double *alo = S + oj + (int64_t)(16 * blo + c) * ldw + 2 * q, *ahi = S + oj + (int64_t)(16 * bhi + c) * ldw + 2 * q;
  f64x2_a8 xlo[8], xhi[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    xhi[u] = u < uhi ? *reinterpret_cast<const f64x2_a8 *>(ahi + 8 * u) : f64x2_a8{0.0, 0.0};
    if (u < 8) xlo[u] = u < ulo ? *reinterpret_cast<const f64x2_a8 *>(alo + 8 * u) : f64x2_a8{0.0, 0.0};
  }
  for (int ch = ch0; ch < ch1; ++ch) {
    const int c0 = 64 * ch;
    if (ch > ch0) __syncthreads();  // every wave is done with the previous chunk
#pragma unroll
    for (int i = 0; i < 16; ++i) *reinterpret_cast<f64x2 *>(&Bs[8 * i + (t >> 5)][2 * (t & 31)]) = f64x2{pre[i].x, pre[i].y};
    __syncthreads();
    if (ch + 1 < ch1) {
#pragma unroll
      for (int i = 0; i < 16; ++i) pre[i] = *reinterpret_cast<const f64x2_a8 *>(Lrow + (int64_t)(8 * i) * lda + (RAG ? max(64 * (ch + 1) + lc, fp) : 64 * (ch + 1) + lc));
    }
    pta_f64x4 acc[2][4];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) acc[rb][cb] = pta_f64x4{0.0, 0.0, 0.0, 0.0};
    double bb[2][2][4];  // [u & 1]: the fragments of step u + 1 are read from LDS before the products of step u are issued
#pragma unroll
    for (int sft = 0; sft < 2; ++sft)
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) bb[0][sft][cb] = Bs[2 * q + sft][16 * cb + c];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (u < uhi) {  // wave-uniform
        if (u + 1 < 16) {
#pragma unroll
          for (int sft = 0; sft < 2; ++sft)
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) bb[(u + 1) & 1][sft][cb] = Bs[8 * (u + 1) + 2 * q + sft][16 * cb + c];
        }
        // four independent accumulators between two products into the same one
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) acc[1][cb] = pta_mfma_f64(xhi[u].x, bb[u & 1][0][cb], acc[1][cb]);
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) acc[1][cb] = pta_mfma_f64(xhi[u].y, bb[u & 1][1][cb], acc[1][cb]);
        if (u < 8 && u < ulo) {
#pragma unroll
          for (int cb = 0; cb < 4; ++cb) acc[0][cb] = pta_mfma_f64(xlo[u < 8 ? u : 0].x, bb[u & 1][0][cb], acc[0][cb]);
#pragma unroll
          for (int cb = 0; cb < 4; ++cb) acc[0][cb] = pta_mfma_f64(xlo[u < 8 ? u : 0].y, bb[u & 1][1][cb], acc[0][cb]);
        }
      }
    }
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * (rb ? bhi : blo) + pta_mfma_row(l, r), col = c0 + 16 * cb + pta_mfma_col(l);
          if (col < oj && (!RAG || col >= fp)) S[(int64_t)row * ldw + col] = -acc[rb][cb][r];
        }
  }
}

// workspace of one matrix: one strip S_j = [-T_j | W_jj] of 128 rows per 128-column block of a panel, leading dimension = the widest
// panel rounded up to whole blocks (NBO + 127 columns at most)
static inline int64_t pta_potrf_ws_ld(int NBO) { return (int64_t)((NBO + 127 + 127) / 128) * 128; }

extern "C" int64_t pta_potrf_workspace_doubles(int n, int B, int flags) {
  const int nbk = (flags >> 8) & 0xFF;
  const int NBO = (nbk ? nbk : 4) * 4 * CH_NB;
  if (n <= NBO || B <= 0 || (flags & (PTA_POTRF_VALU | PTA_POTRF_SUBSTITUTION))) return 0;
  const int64_t ldw = pta_potrf_ws_ld(NBO);
  return (int64_t)B * ldw * ldw;
}

// panel geometry of the workspace scheme
struct pta_ws_panel {
  int k0, nbo, pend, rows, nb, f128;
};
static inline pta_ws_panel pta_ws_panel_at(int n, int NBO, int k0) {
  pta_ws_panel p;
  p.k0 = k0;
  const int want = (k0 == 0 && n > NBO) ? NBO + (n % 128) : NBO;
  p.nbo = (n - k0 < want) ? (n - k0) : want;
  p.pend = k0 + p.nbo;
  p.rows = n - p.pend;
  p.nb = (p.nbo + 127) / 128;
  p.f128 = p.nbo - 128 * (p.nb - 1);  // first block narrower when nbo % 128 != 0
  return p;
}

// X <- X W^T in place for `rows` rows of `wj` columns (W = a lower-triangular wj x wj block of the workspace): a launch must cover
// ONE column tile (a second tile would read columns the first one is overwriting), so a block the launcher would split (few rows or
// a narrow block: 64-wide tiles) goes chunk by chunk, right to left (chunk [c0, c1) needs the block's columns [0, c1) only).
// `Xl` / `kl` > 0 prepend kl columns to the K range (the merged substitution: [X_{<j} | B_j] S_j^T, Sl = the strip's left part).
static int pta_ws_apply_block(double *X, int64_t lda, int64_t strideA, int B, int rows, int wj, int kl, const double *S, int64_t ldw,
                              int64_t sW, int algo, hipStream_t s) {
  const int tile = pta_dgemm_tile_n(rows, wj, kl + wj, algo);
  for (int c1 = wj; c1 > 0; c1 -= tile) {
    const int c0 = c1 > tile ? c1 - tile : 0;
    int rc = pta_dgemm_launch(1, rows, c1 - c0, kl + c1, 1.0, X - kl, lda, 1, S + (int64_t)c0 * ldw, ldw, 0.0, X + c0, lda, 0, B, strideA, sW,
                              strideA, algo, s);
    if (rc != PTA_OK) return rc;
  }
  return PTA_OK;
}

// The recursion of pta_factor_panel on the panel's DIAGONAL block (rows up to `pend`), with the 128-column group as its base case:
// k_diag128 factors the group's diagonal block and writes its inverse W_jj to the workspace, the rows below (inside the diagonal
// block) are one product with W_jj - no 64-column solves, K = 64 updates or separate inversion pass.
static int pta_factor_diag_ws(double *A, int pend, int64_t lda, int64_t sA, int B, int c0, int w, int32_t *info, int algo, const pta_ws_panel &p,
                              double *W, int64_t ldw, int64_t sW, hipStream_t sp) {
  if (w <= 128) {
    const int oj = c0 - p.k0;                                   // offset inside the panel: 0 for the first block, f128 + 128 (j - 1) after
    const int j = oj == 0 ? 0 : (oj - p.f128) / 128 + 1;
    double *Wjj = W + (int64_t)j * 128 * ldw + oj;
    hipLaunchKernelGGL(k_diag128<false>, dim3(B), dim3(256), 0, sp, A, lda, sA, c0, w, Wjj, ldw, sW, info, pta_rag{nullptr, nullptr, nullptr, 0}, nullptr);
    PTA_LAUNCH_CHECK();
    const int rows = pend - c0 - w;
    if (rows <= 0) return PTA_OK;
    return pta_ws_apply_block(A + (int64_t)(c0 + w) * lda + c0, lda, sA, B, rows, w, 0, Wjj, ldw, sW, algo, sp);
  }
  int cols = (w / 2 / 128) * 128;  // right part: a multiple of 128; the left part takes the remainder (the first panel's n mod 128)
  if (cols < 128) cols = 128;
  const int w1 = w - cols;
  int rc = pta_factor_diag_ws(A, pend, lda, sA, B, c0, w1, info, algo, p, W, ldw, sW, sp);
  if (rc != PTA_OK) return rc;
  const int rows = pend - (c0 + w1);
  const double *L21 = A + (int64_t)(c0 + w1) * lda + c0;
  double *A22 = A + (int64_t)(c0 + w1) * lda + (c0 + w1);
  rc = pta_dgemm_launch(1, rows, cols, w1, -1.0, L21, lda, 1, L21, lda, 1.0, A22, lda, 1, B, sA, sA, sA, algo, sp);
  if (rc != PTA_OK) return rc;
  return pta_factor_diag_ws(A, pend, lda, sA, B, c0 + w1, cols, info, algo, p, W, ldw, sW, sp);
}

// (1) + (2): the panel's diagonal block, the 128 x 128 inverses W_jj and the strips S_j = [-W_jj L11[j, <j] | W_jj] the substitution
// multiplies by
static int pta_ws_diag_phase(double *A, int64_t lda, int64_t strideA, int B, int32_t *info, int flags, int algo, const pta_ws_panel &p, double *W,
                             int64_t ldw, int64_t sW, hipStream_t s) {
  int rc;
  if (p.rows <= 0)  // the last panel: nothing below it needs the inverses
    return pta_factor_panel(A, p.pend, lda, strideA, B, p.k0, p.nbo, info, flags, algo, s);
  if (flags & PTA_POTRF_DIAG64) {  // A/B: the 64-column recursion of pta_factor_panel on the diagonal block + one inversion pass
    if ((rc = pta_factor_panel(A, p.pend, lda, strideA, B, p.k0, p.nbo, info, flags, algo, s)) != PTA_OK) return rc;
    hipLaunchKernelGGL(k_inv_blocks, dim3(p.nb, B), dim3(256), 0, s, A, lda, strideA, p.k0, p.nbo, p.f128, W, ldw, sW);
    PTA_LAUNCH_CHECK();
  } else if ((rc = pta_factor_diag_ws(A, p.pend, lda, strideA, B, p.k0, p.nbo, info, algo, p, W, ldw, sW, s)) != PTA_OK) {
    return rc;
  }
  if (p.nb > 1) {
    int groups = 0;
    for (int j = 1; j < p.nb; ++j) groups += ((p.f128 + 128 * (j - 1) + 63) / 64 + PTA_WS_STRIP_GROUP - 1) / PTA_WS_STRIP_GROUP;
    hipLaunchKernelGGL(k_ws_strips<false>, dim3(groups, B), dim3(256), 0, s, A, lda, strideA, p.k0, p.f128, W, ldw, sW, pta_rag{nullptr, nullptr, nullptr, 0});
    PTA_LAUNCH_CHECK();
  }
  return PTA_OK;
}

// (3): blocked substitution on the rows below, left to right: X_j = [X_{<j} | B_j] S_j^T
static int pta_ws_solve_phase(double *A, int64_t lda, int64_t strideA, int B, int algo, const pta_ws_panel &p, double *W, int64_t ldw, int64_t sW,
                              hipStream_t s) {
  double *Bp = A + (int64_t)p.pend * lda + p.k0;
  // algo 4 (PTA_POTRF_SOLVE_ROWS): the panel's blocks in ONE launch, a workgroup per 128-row tile walking them (csrc/pta_solve_rows.hip);
  // needs the 16-byte operand alignment of the DMA kernels (even offsets and pitches) and at least one whole tile of rows
  if (algo == 4 && p.rows >= 128 && !(p.f128 & 1) && !(p.k0 & 1) && !(lda & 1) && !(strideA & 1) && !(ldw & 1) && !(sW & 1) && ((uintptr_t)A % 16) == 0 &&
      ((uintptr_t)W % 16) == 0)
    return pta_ws_solve_rows_launch(Bp, lda, strideA, B, p.rows, p.nb, p.f128, W, ldw, sW, s);
  for (int j = 0; j < p.nb; ++j) {
    const int oj = j == 0 ? 0 : p.f128 + 128 * (j - 1), wj = j == 0 ? p.f128 : 128;
    int rc = pta_ws_apply_block(Bp + oj, lda, strideA, B, p.rows, wj, oj, W + (int64_t)j * 128 * ldw, ldw, sW, algo, s);
    if (rc != PTA_OK) return rc;
  }
  return PTA_OK;
}

// One step of a chain with the workspace scheme; returns the next panel's first column in *k0_io.
static int pta_potrf_step_ws(double *A, int n, int64_t lda, int64_t strideA, int B, int32_t *info, int flags, int algo, int NBO, int *k0_io,
                             double *W, int64_t sW, hipStream_t s, hipEvent_t ev_after_diag) {
  const pta_ws_panel p = pta_ws_panel_at(n, NBO, *k0_io);
  const int64_t ldw = pta_potrf_ws_ld(NBO);
  int rc = pta_ws_diag_phase(A, lda, strideA, B, info, flags, algo, p, W, ldw, sW, s);
  if (rc != PTA_OK) return rc;
  *k0_io = p.pend;
  if (p.rows <= 0) return PTA_OK;
  if (ev_after_diag) PTA_HIP(hipEventRecord(ev_after_diag, s));
  if ((rc = pta_ws_solve_phase(A, lda, strideA, B, algo, p, W, ldw, sW, s)) != PTA_OK) return rc;
  // (4) trailing update
  double *Bp = A + (int64_t)p.pend * lda + p.k0;
  double *A22 = A + (int64_t)p.pend * lda + p.pend;
  return pta_dgemm_launch(1, p.rows, p.rows, p.nbo, -1.0, Bp, lda, 1, Bp, lda, 1.0, A22, lda, 1, B, strideA, strideA, strideA, algo, s);
}

// A whole chain of the workspace scheme WITH LOOK-AHEAD (PTA_POTRF_DIAG_AHEAD): what separates two panels' tile products is the next
// panel's diagonal phase - ~2.4 ms of pivot-by-pivot latency chains on a 1024 x 1024 block per step, during which the matrix cores idle.
// It needs only that block of the trailing matrix, so the trailing update is issued in three pieces - U1 = the next panel's diagonal
// block (36 tiles per matrix), then its sub-diagonal rectangle and the rest - and the next diagonal phase runs on a side stream as
// soon as U1 is done, BESIDE the other two.  (Beside a tile product such kernels run ~7x slower - a dependent fp64 chain waits out the
// 64-cycle MFMA blocks of the waves it shares a SIMD with: 17.9 ms for the phase that takes 2.4 alone - so the look-ahead is only
// used while the rest of the update is long: rows >= PTA_WS_LA_MIN_ROWS below the next panel; 512 / 1024 / 1536 measured the same.)
#define PTA_WS_LA_MIN_ROWS 1024
static int pta_potrf_chain_ws_lookahead(double *A, int n, int64_t lda, int64_t strideA, int B, int32_t *info, int flags, int algo, int NBO, double *W,
                                        int64_t sW, hipStream_t s, hipStream_t side, hipEvent_t ev_u1, hipEvent_t ev_la) {
  pta_ws_panel p = pta_ws_panel_at(n, NBO, 0);
  const int64_t ldw = pta_potrf_ws_ld(NBO);
  int rc = pta_ws_diag_phase(A, lda, strideA, B, info, flags, algo, p, W, ldw, sW, s);
  if (rc != PTA_OK) return rc;
  bool joined = true;  // false while a look-ahead diagonal phase is in flight on `side`
  while (p.rows > 0) {
    if ((rc = pta_ws_solve_phase(A, lda, strideA, B, algo, p, W, ldw, sW, s)) != PTA_OK) break;
    const pta_ws_panel q = pta_ws_panel_at(n, NBO, p.pend);   // the next panel
    double *Bp = A + (int64_t)p.pend * lda + p.k0;            // X: rows below panel p, its columns
    double *A22 = A + (int64_t)p.pend * lda + p.pend;
    // U1: the next panel's diagonal block
    rc = pta_dgemm_launch(1, q.nbo, q.nbo, p.nbo, -1.0, Bp, lda, 1, Bp, lda, 1.0, A22, lda, 1, B, strideA, strideA, strideA, algo, s);
    if (rc != PTA_OK) break;
    const bool la = q.rows >= PTA_WS_LA_MIN_ROWS;
    if (la) {
      if ((rc = hipEventRecord(ev_u1, s)) != hipSuccess || (rc = hipStreamWaitEvent(side, ev_u1, 0)) != hipSuccess) { rc = PTA_E_HIP; break; }
      joined = false;
      rc = pta_ws_diag_phase(A, lda, strideA, B, info, flags, algo, q, W, ldw, sW, side);
      (void)hipEventRecord(ev_la, side);
      if (rc != PTA_OK) break;
    }
    if (q.rows > 0) {
      // U2: rows below the next panel - its sub-diagonal rectangle (what solve(q) will turn into X) and the lower triangle behind it
      const double *Xlo = Bp + (int64_t)q.nbo * lda;
      rc = pta_dgemm_launch(1, q.rows, q.nbo, p.nbo, -1.0, Xlo, lda, 1, Bp, lda, 1.0, A22 + (int64_t)q.nbo * lda, lda, 0, B, strideA, strideA, strideA, algo, s);
      if (rc != PTA_OK) break;
      rc = pta_dgemm_launch(1, q.rows, q.rows, p.nbo, -1.0, Xlo, lda, 1, Xlo, lda, 1.0, A22 + (int64_t)q.nbo * lda + q.nbo, lda, 1, B, strideA, strideA,
                            strideA, algo, s);
      if (rc != PTA_OK) break;
    }
    if (la) {
      (void)hipStreamWaitEvent(s, ev_la, 0);
      joined = true;
    } else if ((rc = pta_ws_diag_phase(A, lda, strideA, B, info, flags, algo, q, W, ldw, sW, s)) != PTA_OK) {
      break;
    }
    p = q;
  }
  if (!joined) (void)hipStreamWaitEvent(s, ev_la, 0);  // error exit: the chain's stream never runs ahead of its side stream
  return rc;
}

// A whole chain of the workspace scheme in LEFT-LOOKING panel order (PTA_POTRF_LEFT): nothing is applied to the trailing
// matrix when a panel is finished - instead, before panel q is factored, its block column (rows [k0_q, n), the panel's nbo_q columns) is
// updated ONCE with everything to its left, C -= L[rows, 0:k0_q] L[k0_q:pend_q, 0:k0_q]^T: one tile product of K = k0_q per block column
// instead of one K = 1024 product per finished panel - the same flops (each tile of the lower triangle still meets every column to its
// left once), but a C tile is read and written once instead of once per panel to its left (920 -> 496 tile round trips per matrix at
// n = 5000) and the product's per-tile intercept (prologue + epilogue, worth 84 columns of K) is paid 496 instead of 920 times.
// PTA_POTRF_LEFT_SPLIT: the update of block column q is split at the previous panel's boundary - U_a(q), K = [0, k0_{q-1}), needs only
// panels <= q - 2 and is issued on the side stream as soon as solve(q - 2) is done, BESIDE panel q - 1's update, diagonal phase and
// substitution (the left-looking counterpart of PTA_POTRF_DIAG_AHEAD); U_b(q), K = [k0_{q-1}, k0_q), follows solve(q - 1) on the chain's
// stream.  Same kernels, same workspace, same diagonal / substitution phases as the right-looking chains above.
static int pta_potrf_chain_ws_left(double *A, int n, int64_t lda, int64_t strideA, int B, int32_t *info, int flags, int algo, int NBO, double *W,
                                   int64_t sW, hipStream_t s, hipStream_t side, hipEvent_t ev_solved, hipEvent_t ev_ua, const pta_fuse *fz = nullptr) {
  if (fz) {
    // FUSED ASSEMBLY (pta_td_assemble_potrf; csrc/pta_td_fused.hip): nothing has been written to A - every block column is COMPUTED by its
    // one update, F phi F^T + diag + ECORR - L[:, <k0] L[:, <k0]^T, block column 0 (nothing to its left) first.  Plain left-looking order.
    const int64_t ldw = pta_potrf_ws_ld(NBO);
    pta_ws_panel p = pta_ws_panel_at(n, NBO, 0);
    int rc = pta_td_fused_launch(n, p.nbo, 0, A, lda, strideA, 0, B, *fz, s);
    if (rc != PTA_OK) return rc;
    if ((rc = pta_ws_diag_phase(A, lda, strideA, B, info, flags, algo, p, W, ldw, sW, s)) != PTA_OK) return rc;
    while (p.rows > 0) {
      if ((rc = pta_ws_solve_phase(A, lda, strideA, B, algo, p, W, ldw, sW, s)) != PTA_OK) return rc;
      const pta_ws_panel q = pta_ws_panel_at(n, NBO, p.pend);
      if ((rc = pta_td_fused_launch(n - q.k0, q.nbo, q.k0, A, lda, strideA, q.k0, B, *fz, s)) != PTA_OK) return rc;
      if ((rc = pta_ws_diag_phase(A, lda, strideA, B, info, flags, algo, q, W, ldw, sW, s)) != PTA_OK) return rc;
      p = q;
    }
    return PTA_OK;
  }
  const bool split = (flags & PTA_POTRF_LEFT_SPLIT) != 0 && side != nullptr;
  const int64_t ldw = pta_potrf_ws_ld(NBO);
  pta_ws_panel p = pta_ws_panel_at(n, NBO, 0);
  int rc;
  if ((flags & PTA_POTRF_DIAG_AHEAD) && !split && side != nullptr) {
    // PTA_POTRF_LEFT | PTA_POTRF_DIAG_AHEAD: every diagonal phase runs on the (high-priority) side stream.  The update of block column q is
    // issued top first - U_top(q): the panel's own nbo x nbo diagonal block, a triangular grid - and diag(q) starts behind it, beside
    // U_rest(q), the rectangle below (no tile is touched twice: top and rest are disjoint).  ev_solved here = "U_top done", ev_ua = "diag done".
    hipEvent_t ev_top = ev_solved, ev_diag = ev_ua;
    if (hipEventRecord(ev_top, s) != hipSuccess || hipStreamWaitEvent(side, ev_top, 0) != hipSuccess) return PTA_E_HIP;  // side starts behind the caller's work
    rc = pta_ws_diag_phase(A, lda, strideA, B, info, flags, algo, p, W, ldw, sW, side);
    (void)hipEventRecord(ev_diag, side);
    while (rc == PTA_OK && p.rows > 0) {
      if (hipStreamWaitEvent(s, ev_diag, 0) != hipSuccess) { rc = PTA_E_HIP; break; }
      if ((rc = pta_ws_solve_phase(A, lda, strideA, B, algo, p, W, ldw, sW, s)) != PTA_OK) break;
      const pta_ws_panel q = pta_ws_panel_at(n, NBO, p.pend);
      const double *Lq = A + (int64_t)q.k0 * lda;                 // rows of panel q, columns to its left
      rc = pta_dgemm_launch(1, q.nbo, q.nbo, q.k0, -1.0, Lq, lda, 1, Lq, lda, 1.0, A + (int64_t)q.k0 * lda + q.k0, lda, 1, B, strideA, strideA, strideA, algo, s);
      if (rc != PTA_OK) break;
      if (hipEventRecord(ev_top, s) != hipSuccess || hipStreamWaitEvent(side, ev_top, 0) != hipSuccess) { rc = PTA_E_HIP; break; }
      rc = pta_ws_diag_phase(A, lda, strideA, B, info, flags, algo, q, W, ldw, sW, side);
      (void)hipEventRecord(ev_diag, side);
      if (rc != PTA_OK) break;
      if (q.rows > 0) {
        const double *Lr = A + (int64_t)q.pend * lda;             // rows below panel q
        rc = pta_dgemm_launch(1, q.rows, q.nbo, q.k0, -1.0, Lr, lda, 1, Lq, lda, 1.0, A + (int64_t)q.pend * lda + q.k0, lda, 0, B, strideA, strideA, strideA, algo, s);
        if (rc != PTA_OK) break;
      }
      p = q;
    }
    (void)hipStreamWaitEvent(s, ev_diag, 0);  // every exit: the chain's stream never runs ahead of its side stream
    return rc;
  }
  rc = pta_ws_diag_phase(A, lda, strideA, B, info, flags, algo, p, W, ldw, sW, s);
  if (rc != PTA_OK) return rc;
  // block column of panel q (rows from its first column down) -= L[rows, ka:kb] L[panel rows, ka:kb]^T, lower part only
  auto update = [&](const pta_ws_panel &q, int ka, int kb, hipStream_t st) {
    if (kb <= ka) return (int)PTA_OK;
    const double *Lr = A + (int64_t)q.k0 * lda + ka;
    double *C = A + (int64_t)q.k0 * lda + q.k0;
    return pta_dgemm_launch(1, n - q.k0, q.nbo, kb - ka, -1.0, Lr, lda, 1, Lr, lda, 1.0, C, lda, 1, B, strideA, strideA, strideA, algo, st);
  };
  bool ua_pending = false;  // a U_a product is in flight on `side` (its completion = ev_ua): U_a(q) covers [0, p.k0), U_b(q) covers [p.k0, q.k0)
  while (p.rows > 0) {
    if ((rc = pta_ws_solve_phase(A, lda, strideA, B, algo, p, W, ldw, sW, s)) != PTA_OK) break;
    const pta_ws_panel q = pta_ws_panel_at(n, NBO, p.pend);  // the next panel
    if (split) {
      if (hipEventRecord(ev_solved, s) != hipSuccess) { rc = PTA_E_HIP; break; }
      if (ua_pending) {  // U_a(q) (K = [0, p.k0)) was issued one step ago: U_b(q) must follow it (same C tiles)
        if (hipStreamWaitEvent(s, ev_ua, 0) != hipSuccess) { rc = PTA_E_HIP; break; }
        ua_pending = false;
        if ((rc = update(q, p.k0, q.k0, s)) != PTA_OK) break;
      } else if ((rc = update(q, 0, q.k0, s)) != PTA_OK) {  // first step: nothing was run ahead
        break;
      }
      if (q.rows > 0) {  // run ahead: the panel after q, with everything up to the end of panel p
        const pta_ws_panel r = pta_ws_panel_at(n, NBO, q.pend);
        if (hipStreamWaitEvent(side, ev_solved, 0) != hipSuccess) { rc = PTA_E_HIP; break; }
        ua_pending = true;
        rc = update(r, 0, q.k0, side);
        (void)hipEventRecord(ev_ua, side);
        if (rc != PTA_OK) break;
      }
    } else if ((rc = update(q, 0, q.k0, s)) != PTA_OK) {
      break;
    }
    if ((rc = pta_ws_diag_phase(A, lda, strideA, B, info, flags, algo, q, W, ldw, sW, s)) != PTA_OK) break;
    p = q;
  }
  if (ua_pending) (void)hipStreamWaitEvent(s, ev_ua, 0);  // error exit: the chain's stream never runs ahead of its side stream
  return rc;
}

// Right-looking over panels of NB = 1024 columns; the trailing update of a panel is ONE product with K = NB over the
// lower-triangular 128 x 128 tiles.  A batch is split into up to four independent CHAINS of matrices, each on its own
// internal stream: the panel steps of a chain are short, serial and partly memory bound (potf2 on one workgroup per matrix;
// each 64-column solve reads and writes its panel once) while its trailing updates are MFMA bound, so letting the hardware
// interleave the chains fills one chain's panel phases with another chain's matrix-core work - look-ahead across the batch
// instead of inside one matrix (an in-matrix look-ahead, next panel on a high-priority stream beside the bulk update, measured
// slower: it splits every trailing update in two and the concurrent halves slow each other down).
static int pta_potrf_impl(double *A, int n, int64_t lda, int64_t strideA, int B, int32_t *info, int flags, double *work,
                          int64_t work_doubles, void *stream, const pta_fuse *fz = nullptr) {
  PTA_REQUIRE(A && info, PTA_E_ARG, "pta_potrf_batched: NULL argument");
  PTA_REQUIRE(n > 0 && n <= 65535 && B > 0 && B <= 65535, PTA_E_ARG, "pta_potrf_batched: n=%d B=%d", n, B);
  PTA_REQUIRE(lda >= n && (B == 1 || strideA >= (int64_t)(n - 1) * lda + n), PTA_E_ARG, "pta_potrf_batched: lda=%lld strideA=%lld too small",
              (long long)lda, (long long)strideA);
  hipStream_t s = pta_stream(stream);
  // 0: VALU reference GEMM + substitution panel solve (cross-check); 2 (default): MFMA kernels, the 128 x 128-tile products' operand
  // slabs brought in by LDS DMA (k_dgemm_glds128: 66 against 59 TFLOP/s at K = 1024, the whole 68 x 5000^2 batch 56.4 against 60.0
  // ms); 1 (PTA_POTRF_REG_STAGING): the same products with register-staged slabs (round 2's kernel, kept for the A/B)
  const int algo = (flags & PTA_POTRF_VALU) ? 0 : ((flags & PTA_POTRF_REG_STAGING) ? 1 : ((flags & PTA_POTRF_EPI1) ? 3 : ((flags & PTA_POTRF_SOLVE_ROWS) ? 4 : 2)));
  const int nbk = (flags >> 8) & 0xFF;
  const int NBO = (nbk ? nbk : 4) * 4 * CH_NB;      // panel width: 1024 columns unless overridden (PTA_POTRF_NB)
  int nchain = (flags >> 16) & 0xF;                  // PTA_POTRF_CHAINS; 0 = default
  // default: two chains (with PTA_POTRF_DIAG_AHEAD and the 128-column base case: 53.2 ms against 54.1 with one chain, 58.3 with three) - from
  // eight matrices up; a handful of matrices is one chain (3 x 10 000^2, round 6: 29.1-30.7 ms against 33.6-34.4 with two: each chain's
  // launches would fill a fraction of the chip and the chains only delay each other)
  if (nchain == 0) nchain = B >= 8 ? 2 : 1;
  if (nchain > PTA_POTRF_MAX_CHAINS) nchain = PTA_POTRF_MAX_CHAINS;
  if (nchain > B) nchain = B;
  if ((flags & PTA_POTRF_NO_LOOKAHEAD) || !algo || n <= NBO) nchain = 1;
  // workspace scheme (pta_potrf_batched_ws): panel solves through the explicit inverse of the panel's diagonal block
  const int64_t need = pta_potrf_workspace_doubles(n, B, flags);
  const bool use_ws = work != nullptr && need > 0 && work_doubles >= need && algo;
  const int64_t sWm = use_ws ? need / B : 0;  // workspace doubles per matrix
  auto chain_step = [&](double *Ab, int Bc, int32_t *infob, double *Wb, int *k0p, hipStream_t st, hipEvent_t ev) {
    return use_ws ? pta_potrf_step_ws(Ab, n, lda, strideA, Bc, infob, flags, algo, NBO, k0p, Wb, sWm, st, ev)
                  : pta_potrf_step(Ab, n, lda, strideA, Bc, infob, flags, algo, NBO, k0p, st);
  };
  PTA_REQUIRE(!fz || (use_ws && (flags & PTA_POTRF_LEFT) && !(flags & (PTA_POTRF_NO_LOOKAHEAD | PTA_POTRF_LEFT_SPLIT | PTA_POTRF_DIAG_AHEAD)) && algo >= 2),
              PTA_E_ARG, "pta_td_assemble_potrf: needs the workspace scheme (n > panel width, workspace of pta_potrf_workspace_doubles) in plain left-looking order");
  PTA_HIP(hipMemsetAsync(info, 0, sizeof(int32_t) * B, s));
  if (use_ws && (flags & (PTA_POTRF_DIAG_AHEAD | PTA_POTRF_LEFT)) && !(flags & PTA_POTRF_NO_LOOKAHEAD)) {
    pta_potrf_ctx *cx = nullptr;
    int rc = pta_potrf_ctx_get(&cx, nchain);
    if (rc != PTA_OK) return rc;
    PTA_HIP(hipEventRecord(cx->ev_in, s));
    int rc_chain = PTA_OK;
    for (int c = 0; c < nchain && rc_chain == PTA_OK; ++c) {
      const int b0 = (int)((int64_t)B * c / nchain), b1 = (int)((int64_t)B * (c + 1) / nchain);
      hipStream_t sc = nchain == 1 ? s : cx->chain[c];
      if (nchain > 1) PTA_HIP(hipStreamWaitEvent(sc, cx->ev_in, 0));
      pta_fuse fc;
      if (fz) {  // this chain's matrices: the assembly operands start b0 matrices further on
        fc = *fz;
        const int64_t o = (int64_t)b0 * fz->toa_stride;
        fc.Fr += o * 64, fc.Gr += o * 64, fc.sigma2 += o;
        if (fc.epoch) fc.epoch += o, fc.ecorr2 += o;
      }
      rc_chain = (flags & PTA_POTRF_LEFT)
                     ? pta_potrf_chain_ws_left(A + (int64_t)b0 * strideA, n, lda, strideA, b1 - b0, info + b0, flags, algo, NBO, work + (int64_t)b0 * sWm, sWm,
                                               sc, cx->side[c], cx->ev_u1[c], cx->ev_la[c], fz ? &fc : nullptr)
                     : pta_potrf_chain_ws_lookahead(A + (int64_t)b0 * strideA, n, lda, strideA, b1 - b0, info + b0, flags, algo, NBO,
                                                    work + (int64_t)b0 * sWm, sWm, sc, cx->side[c], cx->ev_u1[c], cx->ev_la[c]);
    }
    if (nchain > 1)
      for (int c = 0; c < nchain; ++c) {
        (void)hipEventRecord(cx->ev_out[c], cx->chain[c]);
        (void)hipStreamWaitEvent(s, cx->ev_out[c], 0);
      }
    if (rc_chain != PTA_OK) return rc_chain;
  } else if (nchain == 1) {
    for (int k0 = 0; k0 < n;) {
      int rc = chain_step(A, B, info, work, &k0, s, nullptr);
      if (rc != PTA_OK) return rc;
    }
  } else {
    pta_potrf_ctx *cx = nullptr;
    int rc = pta_potrf_ctx_get(&cx, nchain);
    if (rc != PTA_OK) return rc;
    PTA_HIP(hipEventRecord(cx->ev_in, s));
    // the launches are ENQUEUED panel step by panel step across the chains (a chain's ~235 launches take the host longer than
    // the first panel takes the device: enqueued chain after chain, the second chain starts late - 13 ms under rocprofv3 - and finishes alone; starting chain c only after chain c-1's first panel, so
    // that panel phases meet trailing updates from the start, measured 2 ms SLOWER: the panel kernels are real work, not idle time)
    int k0[PTA_POTRF_MAX_CHAINS] = {0, 0, 0, 0};
    int rc_chain = PTA_OK;
    for (int step = 0, live = nchain; live > 0 && rc_chain == PTA_OK; ++step) {
      live = 0;
      for (int c = 0; c < nchain; ++c) {
        if (k0[c] >= n) continue;
        const int b0 = (int)((int64_t)B * c / nchain), b1 = (int)((int64_t)B * (c + 1) / nchain);
        if (step == 0) {
          PTA_HIP(hipStreamWaitEvent(cx->chain[c], cx->ev_in, 0));
          // workspace scheme: the chains start OUT OF PHASE - chain c begins once chain c - 1 has finished its first diagonal phase - so
          // that from then on one chain's diagonal phase (a few small, latency-bound kernels) runs beside another's tile products
          // instead of all chains idling the matrix pipe through their diagonal phases together and then sharing it
          if (use_ws && c > 0 && !(flags & PTA_POTRF_LOCKSTEP)) PTA_HIP(hipStreamWaitEvent(cx->chain[c], cx->ev_diag[c - 1], 0));
        }
        rc_chain = chain_step(A + (int64_t)b0 * strideA, b1 - b0, info + b0, use_ws ? work + (int64_t)b0 * sWm : nullptr, &k0[c], cx->chain[c],
                              (use_ws && step == 0) ? cx->ev_diag[c] : nullptr);
        if (rc_chain != PTA_OK) break;
        if (k0[c] < n) ++live;
      }
    }
    // join on EVERY exit, error included: the caller's stream must not run ahead of (and the caller must not free A under)
    // chain kernels that are still in flight
    for (int c = 0; c < nchain; ++c) {
      (void)hipEventRecord(cx->ev_out[c], cx->chain[c]);
      (void)hipStreamWaitEvent(s, cx->ev_out[c], 0);
    }
    if (rc_chain != PTA_OK) return rc_chain;
  }
  if (flags & PTA_POTRF_ZERO_UPPER) {
    hipLaunchKernelGGL(k_zero_upper, dim3(pta_cdiv(n, 256), n, B), dim3(256), 0, s, A, n, lda, strideA);
    PTA_LAUNCH_CHECK();
  }
  return PTA_OK;
}

// ================================================================================================================================
// Ragged batches: matrices of DIFFERENT orders as ONE schedule (pta_potrf_ragged).
//
// A real pulsar timing array has as many TOA counts as pulsars (noise_dicts/ng15_dict.json: 68 pulsars, 68 different N_a;
// test_partim: 7758 / 23023 / 35037), and the reference handles that by construction - it loops over pulsars
// (red_noise.py:286-298).  Batching by equal order would run such an array as 68 batches of ONE: one workgroup in every
// diagonal-phase kernel on a 256-CU chip and a latency chain per matrix.  Instead the matrices are END-ALIGNED: embedded in a
// virtual matrix of order E = NBO (Tmax + 1) whose bottom-right corner they share (virtual index = real index + front[b],
// front[b] = E - n[b]).  Time step T = Tmax ... 0 factors the virtual panel [E - NBO (T + 1), E - NBO T): for every matrix that has
// reached it (n[b] > NBO T - a prefix of the batch sorted by decreasing order) panel boundaries, the trailing size NBO T and all
// tile grids are the SAME, so every kernel of the step is one launch over the active prefix; a matrix ENTERS at the step that
// contains its first column with a panel cut at its front, which the kernels mask (pta_rag: k_diag128<true>, k_ws_strips<true>,
// the ragged tile products).  Cost: the diagonal phases - the latency chains - are paid once per TIME STEP instead of once per
// matrix and panel; the tile products of a step cover all active matrices.  Uniform batches are the special case front[b] = const.
// Orders, offsets and leading dimensions must be even (16-byte operand rows; pad an odd order with an identity row / column at its end).
// ================================================================================================================================
#define PTA_RAG_MAGIC 0x5054415241474544LL
#define PTA_RAG_HDR 8
#define PTA_RAG_CHDR 8

struct pta_rag_chain {
  double *A;
  pta_rag rg;            // device arrays of this chain, sorted by decreasing order
  const int64_t *idx;    // device: caller's index of chain position
  const int64_t *n;      // HOST: orders, decreasing
  int Bc, E, Tmax, NBO;
  double *W;
  int64_t ldw, sW;
  int32_t *info;
};

static inline int pta_rag_active(const pta_rag_chain &c, int T) {  // matrices that have reached time step T: n > NBO T (a prefix)
  int b = 0;
  while (b < c.Bc && c.n[b] > (int64_t)c.NBO * T) ++b;
  return b;
}

// X <- [X_{<j} | B_j] S^T on `rows` rows from virtual row rv: block columns [cblk, cblk + wj), kl columns to their left prepended to the
// K range, S = the strip (its column kl <-> virtual column cblk).  One column tile per launch, right to left (in place).
static int pta_rag_apply_block(const pta_rag_chain &c, int B, int rv, int rows, int cblk, int wj, int kl, const double *S, hipStream_t s) {
  const int tile = pta_dgemm_tile_n(rows, wj, kl + wj, 2);
  for (int c1 = wj; c1 > 0; c1 -= tile) {
    const int c0 = c1 > tile ? c1 - tile : 0;
    int rc = pta_dgemm_launch_rag(rows, c1 - c0, kl + c1, 1.0, c.A, rv, cblk + c0, cblk - kl, S + (int64_t)c0 * c.ldw, c.ldw, c.sW, 0.0, 0, B, c.rg, s);
    if (rc != PTA_OK) return rc;
  }
  return PTA_OK;
}

// the recursion of pta_factor_diag_ws in virtual coordinates: diagonal block [k0, pend) of the panel, columns [c0, c0 + w)
static int pta_rag_factor_diag(const pta_rag_chain &c, int B, int k0, int pend, int c0, int w, hipStream_t sp) {
  if (w <= 128) {
    const int oj = c0 - k0, j = oj / 128;
    double *Wjj = c.W + (int64_t)j * 128 * c.ldw + oj;
    hipLaunchKernelGGL(k_diag128<true>, dim3(B), dim3(256), 0, sp, c.A, (int64_t)0, (int64_t)0, c0, w, Wjj, c.ldw, c.sW, c.info, c.rg, c.idx);
    PTA_LAUNCH_CHECK();
    const int rows = pend - c0 - w;
    if (rows <= 0) return PTA_OK;
    return pta_rag_apply_block(c, B, c0 + w, rows, c0, w, 0, Wjj, sp);
  }
  int cols = (w / 2 / 128) * 128;
  if (cols < 128) cols = 128;
  const int w1 = w - cols;
  int rc = pta_rag_factor_diag(c, B, k0, pend, c0, w1, sp);
  if (rc != PTA_OK) return rc;
  const int rows = pend - (c0 + w1);
  rc = pta_dgemm_launch_rag(rows, cols, w1, -1.0, c.A, c0 + w1, c0 + w1, c0, nullptr, 0, 0, 1.0, 1, B, c.rg, sp);
  if (rc != PTA_OK) return rc;
  return pta_rag_factor_diag(c, B, k0, pend, c0 + w1, cols, sp);
}

static int pta_rag_diag_phase(const pta_rag_chain &c, int T, int B, hipStream_t s) {
  const int k0 = c.E - c.NBO * (T + 1), pend = k0 + c.NBO;
  int rc = pta_rag_factor_diag(c, B, k0, pend, k0, c.NBO, s);
  if (rc != PTA_OK || T == 0) return rc;  // the last panel: nothing below it needs the strips
  const int nb = c.NBO / 128;
  int groups = 0;
  for (int j = 1; j < nb; ++j) groups += ((128 * j + 63) / 64 + PTA_WS_STRIP_GROUP - 1) / PTA_WS_STRIP_GROUP;
  hipLaunchKernelGGL(k_ws_strips<true>, dim3(groups, B), dim3(256), 0, s, c.A, (int64_t)0, (int64_t)0, k0, 128, c.W, c.ldw, c.sW, c.rg);
  PTA_LAUNCH_CHECK();
  return PTA_OK;
}

static int pta_rag_solve_phase(const pta_rag_chain &c, int T, int B, hipStream_t s) {
  const int k0 = c.E - c.NBO * (T + 1), pend = k0 + c.NBO, rows = c.NBO * T, nb = c.NBO / 128;
  for (int j = 0; j < nb; ++j) {
    int rc = pta_rag_apply_block(c, B, pend, rows, k0 + 128 * j, 128, 128 * j, c.W + (int64_t)j * 128 * c.ldw, s);
    if (rc != PTA_OK) return rc;
  }
  return PTA_OK;
}

// one chain, all time steps; `side` != nullptr: the next panel's diagonal phase runs ahead beside the bulk of the trailing update
// (as pta_potrf_chain_ws_lookahead; the phase of step T - 1 also covers the matrices that enter there)
static int pta_rag_chain_run(const pta_rag_chain &c, hipStream_t s, hipStream_t side, hipEvent_t ev_u1, hipEvent_t ev_la) {
  int T = c.Tmax, B = pta_rag_active(c, T);
  int rc = pta_rag_diag_phase(c, T, B, s);
  if (rc != PTA_OK) return rc;
  bool joined = true;
  while (T > 0) {
    if ((rc = pta_rag_solve_phase(c, T, B, s)) != PTA_OK) break;
    const int k0 = c.E - c.NBO * (T + 1), pend = k0 + c.NBO, Tq = T - 1, Bq = pta_rag_active(c, Tq), rows_q = c.NBO * Tq;
    // U1: the next panel's diagonal block
    if ((rc = pta_dgemm_launch_rag(c.NBO, c.NBO, c.NBO, -1.0, c.A, pend, pend, k0, nullptr, 0, 0, 1.0, 1, B, c.rg, s)) != PTA_OK) break;
    const bool la = side != nullptr && rows_q >= PTA_WS_LA_MIN_ROWS;
    if (la) {
      if (hipEventRecord(ev_u1, s) != hipSuccess || hipStreamWaitEvent(side, ev_u1, 0) != hipSuccess) { rc = PTA_E_HIP; break; }
      joined = false;
      rc = pta_rag_diag_phase(c, Tq, Bq, side);
      (void)hipEventRecord(ev_la, side);
      if (rc != PTA_OK) break;
    }
    if (rows_q > 0) {
      // U2: the rows below the next panel - its sub-diagonal rectangle and the lower triangle behind it
      if ((rc = pta_dgemm_launch_rag(rows_q, c.NBO, c.NBO, -1.0, c.A, pend + c.NBO, pend, k0, nullptr, 0, 0, 1.0, 0, B, c.rg, s)) != PTA_OK) break;
      if ((rc = pta_dgemm_launch_rag(rows_q, rows_q, c.NBO, -1.0, c.A, pend + c.NBO, pend + c.NBO, k0, nullptr, 0, 0, 1.0, 1, B, c.rg, s)) != PTA_OK) break;
    }
    if (la) {
      (void)hipStreamWaitEvent(s, ev_la, 0);
      joined = true;
    } else if ((rc = pta_rag_diag_phase(c, Tq, Bq, s)) != PTA_OK) {
      break;
    }
    T = Tq;
    B = Bq;
  }
  if (!joined) (void)hipStreamWaitEvent(s, ev_la, 0);
  return rc;
}

// The same chain in LEFT-LOOKING order (PTA_POTRF_LEFT in the plan's flags; round 6, as pta_potrf_chain_ws_left for uniform batches): no
// trailing updates - before the panel of time step Tq is factored, its block column (virtual rows [kq, E), the panel's NBO columns) is
// updated ONCE with every virtual column to its left, K = kq: one masked tile product over the matrices active at Tq (a matrix's K loop
// starts at the slab that holds its front; one that enters at this step has nothing to its left and leaves at once).  No look-ahead.
static int pta_rag_chain_run_left(const pta_rag_chain &c, hipStream_t s) {
  int T = c.Tmax, B = pta_rag_active(c, T);
  int rc = pta_rag_diag_phase(c, T, B, s);
  while (rc == PTA_OK && T > 0) {
    if ((rc = pta_rag_solve_phase(c, T, B, s)) != PTA_OK) break;
    const int Tq = T - 1, Bq = pta_rag_active(c, Tq), kq = c.E - c.NBO * (Tq + 1);
    if ((rc = pta_dgemm_launch_rag(c.E - kq, c.NBO, kq, -1.0, c.A, kq, kq, 0, nullptr, 0, 0, 1.0, 1, Bq, c.rg, s)) != PTA_OK) break;
    rc = pta_rag_diag_phase(c, Tq, Bq, s);
    T = Tq;
    B = Bq;
  }
  return rc;
}

extern "C" int64_t pta_potrf_ragged_plan_words(int B) { return B > 0 ? PTA_RAG_HDR + PTA_RAG_CHDR * PTA_POTRF_MAX_CHAINS + 5 * (int64_t)B : 0; }

extern "C" int pta_potrf_ragged_plan(const int32_t *n, const int64_t *off, const int64_t *ld, int B, int flags, int64_t *plan, int64_t *work_doubles) {
  PTA_REQUIRE(n && off && ld && plan && work_doubles, PTA_E_ARG, "pta_potrf_ragged_plan: NULL argument");
  PTA_REQUIRE(B > 0 && B <= 65535, PTA_E_ARG, "pta_potrf_ragged_plan: B=%d", B);
  PTA_REQUIRE(!(flags & (PTA_POTRF_VALU | PTA_POTRF_SUBSTITUTION | PTA_POTRF_REG_STAGING | PTA_POTRF_DIAG64 | PTA_POTRF_ZERO_UPPER)), PTA_E_ARG,
              "pta_potrf_ragged_plan: flags 0x%x not supported by the ragged schedule (chains, panel width and NO_LOOKAHEAD are)", flags);
  for (int b = 0; b < B; ++b)
    PTA_REQUIRE(n[b] >= 2 && n[b] <= (1 << 20) && !(n[b] & 1) && !(off[b] & 1) && !(ld[b] & 1) && ld[b] >= n[b] && off[b] >= 0, PTA_E_ARG,
                "pta_potrf_ragged_plan: matrix %d: n=%d off=%lld ld=%lld (even order / offset / leading dimension needed, ld >= n)", b, n[b],
                (long long)off[b], (long long)ld[b]);
  // panel width (PTA_POTRF_NB(k): 256 k columns).  Default: 1024, or 2048 when the work sits in large matrices - flop-weighted mean
  // order sum n^4 / sum n^3 >= 16384: half as many trailing-update passes over the big trailing matrices (each reads and writes its C
  // tiles once) and half as many time steps; measured on the ng15-like array (orders 526 ... 33 274): 63.4 against 61.8 TFLOP/s, 1536
  // columns 62.7; at 68 x 5000^2, where a panel is a fifth of the matrix, 2048 columns cost 12 %
  int nbk = (flags >> 8) & 0xFF;
  if (!nbk) {
    double s3 = 0.0, s4 = 0.0;
    for (int b = 0; b < B; ++b) {
      const double x = (double)n[b];
      s3 += x * x * x;
      s4 += x * x * x * x;
    }
    // ... from six matrices up: a handful of large matrices keeps 1024 columns (round 6, scripts/gpu_r6_run15.sh: config 2's three
    // matrices 0.709 against 0.698 of peak, 3 x 20 000^2 0.684 against 0.671; 12 large matrices 0.824 against 0.835, 24: 0.794 / 0.805)
    nbk = (s4 >= 16384.0 * s3 && B >= 6) ? 8 : 4;
  }
  const int NBO = nbk * 4 * CH_NB;
  int nchain = (flags >> 16) & 0xF;
  if (nchain == 0) nchain = 2;
  if (nchain > PTA_POTRF_MAX_CHAINS) nchain = PTA_POTRF_MAX_CHAINS;
  if (nchain > B) nchain = B;
  if (flags & PTA_POTRF_NO_LOOKAHEAD) nchain = 1;
  // matrices sorted by decreasing order (ties: caller's order) and dealt to the chains in turn, so that every chain gets the same
  // mix of orders and its active set is a prefix at every time step
  int *order = (int *)malloc(sizeof(int) * (size_t)B);
  PTA_REQUIRE(order, PTA_E_ARG, "pta_potrf_ragged_plan: out of host memory");
  for (int b = 0; b < B; ++b) order[b] = b;
  for (int i = 1; i < B; ++i) {  // insertion sort, stable (B is a pulsar count)
    const int v = order[i];
    int k = i;
    while (k > 0 && n[order[k - 1]] < n[v]) order[k] = order[k - 1], --k;
    order[k] = v;
  }
  const int64_t ldw = NBO, sW = ldw * ldw;
  int64_t words = PTA_RAG_HDR + PTA_RAG_CHDR * PTA_POTRF_MAX_CHAINS, wdoubles = 0;
  for (int c = 0; c < nchain; ++c) {
    const int Bc = (B - c + nchain - 1) / nchain;
    int64_t *h = plan + PTA_RAG_HDR + PTA_RAG_CHDR * c;
    const int nmax = n[order[c]];
    const int Tmax = (nmax - 1) / NBO, E = NBO * (Tmax + 1);
    h[0] = Bc, h[1] = E, h[2] = Tmax, h[3] = words, h[4] = wdoubles, h[5] = h[6] = h[7] = 0;
    int64_t *a_off = plan + words, *a_ld = a_off + Bc, *a_front = a_ld + Bc, *a_idx = a_front + Bc, *a_n = a_idx + Bc;
    for (int k = 0; k < Bc; ++k) {
      const int b = order[c + k * nchain];
      const int64_t front = E - n[b];
      a_off[k] = off[b] - front * (ld[b] + 1);
      a_ld[k] = ld[b];
      a_front[k] = front;
      a_idx[k] = b;
      a_n[k] = n[b];
    }
    words += 5 * (int64_t)Bc;
    wdoubles += (int64_t)Bc * sW;
  }
  free(order);
  plan[0] = PTA_RAG_MAGIC, plan[1] = B, plan[2] = flags, plan[3] = nchain, plan[4] = NBO, plan[5] = wdoubles, plan[6] = words, plan[7] = 0;
  *work_doubles = wdoubles;
  return PTA_OK;
}

extern "C" int pta_potrf_ragged(double *A, const int64_t *plan, const int64_t *plan_dev, int32_t *info, double *work, int64_t work_doubles,
                                void *stream) {
  PTA_REQUIRE(A && plan && plan_dev && info && work, PTA_E_ARG, "pta_potrf_ragged: NULL argument");
  PTA_REQUIRE(plan[0] == PTA_RAG_MAGIC, PTA_E_ARG, "pta_potrf_ragged: `plan` was not written by pta_potrf_ragged_plan");
  PTA_REQUIRE(work_doubles >= plan[5], PTA_E_ARG, "pta_potrf_ragged: workspace of %lld doubles, %lld needed", (long long)work_doubles, (long long)plan[5]);
  PTA_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)work % 16) == 0, PTA_E_ARG, "pta_potrf_ragged: A and work must be 16-byte aligned");
  const int B = (int)plan[1], flags = (int)plan[2], nchain = (int)plan[3], NBO = (int)plan[4];
  PTA_REQUIRE(B > 0 && B <= 65535 && nchain >= 1 && nchain <= PTA_POTRF_MAX_CHAINS && nchain <= B && NBO >= 128 && NBO <= 16384 && NBO % 128 == 0 &&
                  plan[6] == pta_potrf_ragged_plan_words(B),
              PTA_E_ARG, "pta_potrf_ragged: inconsistent plan header (B=%d chains=%d panel=%d words=%lld)", B, nchain, NBO, (long long)plan[6]);
  {
    int64_t seen = 0;
    for (int c = 0; c < nchain; ++c) {
      const int64_t *h = plan + PTA_RAG_HDR + PTA_RAG_CHDR * c;
      PTA_REQUIRE(h[0] > 0 && h[0] <= B && h[3] >= PTA_RAG_HDR + PTA_RAG_CHDR * PTA_POTRF_MAX_CHAINS && h[3] + 5 * h[0] <= plan[6] && h[1] == (int64_t)NBO * (h[2] + 1) &&
                      h[4] >= 0 && h[4] + h[0] * (int64_t)NBO * NBO <= plan[5],
                  PTA_E_ARG, "pta_potrf_ragged: inconsistent plan (chain %d)", c);
      seen += h[0];
    }
    PTA_REQUIRE(seen == B, PTA_E_ARG, "pta_potrf_ragged: the plan's chains hold %lld of %d matrices", (long long)seen, B);
  }
  hipStream_t s = pta_stream(stream);
  PTA_HIP(hipMemsetAsync(info, 0, sizeof(int32_t) * B, s));
  pta_potrf_ctx *cx = nullptr;
  int rc = pta_potrf_ctx_get(&cx, nchain);
  if (rc != PTA_OK) return rc;
  const bool la = !(flags & PTA_POTRF_NO_LOOKAHEAD);
  if (nchain > 1 || la) PTA_HIP(hipEventRecord(cx->ev_in, s));
  int rc_chain = PTA_OK;
  for (int c = 0; c < nchain && rc_chain == PTA_OK; ++c) {
    const int64_t *h = plan + PTA_RAG_HDR + PTA_RAG_CHDR * c;
    pta_rag_chain ch;
    ch.A = A;
    ch.Bc = (int)h[0], ch.E = (int)h[1], ch.Tmax = (int)h[2], ch.NBO = NBO;
    const int64_t *d = plan_dev + h[3];
    ch.rg = pta_rag{d, d + ch.Bc, d + 2 * ch.Bc, (flags & PTA_POTRF_EPI1) ? 1 : 0};
    ch.idx = d + 3 * ch.Bc;
    ch.n = plan + h[3] + 4 * ch.Bc;
    ch.W = work + h[4];
    ch.ldw = NBO, ch.sW = (int64_t)NBO * NBO;
    ch.info = info;
    hipStream_t sc = nchain == 1 ? s : cx->chain[c];
    // (no early return from here on: the join below must run on every path)
    if ((nchain > 1 && hipStreamWaitEvent(sc, cx->ev_in, 0) != hipSuccess) || (la && hipStreamWaitEvent(cx->side[c], cx->ev_in, 0) != hipSuccess)) {
      pta_set_error("pta_potrf_ragged: hipStreamWaitEvent failed");
      rc_chain = PTA_E_HIP;
      break;
    }
    rc_chain = (flags & PTA_POTRF_LEFT) ? pta_rag_chain_run_left(ch, sc) : pta_rag_chain_run(ch, sc, la ? cx->side[c] : nullptr, cx->ev_u1[c], cx->ev_la[c]);
  }
  if (nchain > 1)
    for (int c = 0; c < nchain; ++c) {  // join on every exit, error included
      (void)hipEventRecord(cx->ev_out[c], cx->chain[c]);
      (void)hipStreamWaitEvent(s, cx->ev_out[c], 0);
    }
  return rc_chain;
}

extern "C" int pta_potrf_batched_ex(double *A, int n, int64_t lda, int64_t strideA, int B, int32_t *info, int flags,
                                    void *stream) {
  return pta_potrf_impl(A, n, lda, strideA, B, info, flags, nullptr, 0, stream);
}

// Assembly + factorisation of a uniform batch of TD covariances in one call: the matrices are never written as covariances - every block
// column is computed by its left-looking update (csrc/pta_td_fused.hip).  A = the factor buffer (contents ignored), even order / pitch / stride.
extern "C" int pta_td_assemble_potrf(const double *Fr, const double *Gr, int kf, const double *sigma2, const int32_t *epoch_of, const double *ecorr2,
                                     double *A, int n, int64_t lda, int64_t strideA, int B, int32_t *info, int flags, double *work,
                                     int64_t work_doubles, void *stream) {
  PTA_REQUIRE(Fr && Gr && sigma2 && (!epoch_of || ecorr2), PTA_E_ARG, "pta_td_assemble_potrf: NULL argument");
  PTA_REQUIRE(kf >= 0 && kf <= 64 && n > 0 && !(n & 1) && !(lda & 1) && !(strideA & 1), PTA_E_ARG,
              "pta_td_assemble_potrf: kf=%d n=%d lda=%lld strideA=%lld (kf <= 64; even order, pitch and stride)", kf, n, (long long)lda, (long long)strideA);
  pta_fuse fz{Fr, Gr, sigma2, epoch_of, epoch_of ? ecorr2 : nullptr, (int64_t)n, kf};
  return pta_potrf_impl(A, n, lda, strideA, B, info, flags | PTA_POTRF_LEFT, work, work_doubles, stream, &fz);
}

extern "C" int pta_potrf_batched_ws(double *A, int n, int64_t lda, int64_t strideA, int B, int32_t *info, int flags, double *work,
                                    int64_t work_doubles, void *stream) {
  return pta_potrf_impl(A, n, lda, strideA, B, info, flags, work, work_doubles, stream);
}

extern "C" int pta_potrf_batched(double *A, int n, int B, int32_t *info, void *stream) {
  return pta_potrf_batched_ex(A, n, n, (int64_t)n * n, B, info, PTA_POTRF_ZERO_UPPER, stream);
}
