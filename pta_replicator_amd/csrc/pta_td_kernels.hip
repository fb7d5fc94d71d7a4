// TD ("time-domain") mode: the dense path BASELINE.json's north_star describes - per-pulsar covariance
// assembly, blocked fp64 Cholesky (pta_potrf_batched, pta_potrf.hip) and L . Z.  The reference has no
// such path; the covariance is the one implied by its RN/WN/ECORR synthesis (App. A.1):
//   C[i,j] = sum_c phi[c] F[i,c] F[j,c] + (i==j) sigma2[i] + (epoch_i == epoch_j) ecorr2[i]
// (red_noise.py:98-101,126-128; white_noise.py:105-109,182).
#include <type_traits>
#include "pta_common.h"
#include "pta_mfma.h"
#include "pta_rng.h"

#define TBM 64
#define TBK 16
#define TLD 80

// Lower-triangular tiles only (the Cholesky reads nothing else).  Ft is already K-major, which is the LDS
// operand layout of the MFMA GEMM, so both slabs are loaded with fully coalesced rows; phi is applied on
// the way in.  HBM traffic = the C tiles written once (8 bytes per element of the lower triangle).
__global__ __launch_bounds__(256) void k_td_cov(const double *__restrict__ Ft, int64_t ldf, int N, int K,
                                                const double *__restrict__ phi, const double *__restrict__ sigma2,
                                                const int32_t *__restrict__ epoch_of, const double *__restrict__ ecorr2,
                                                double *__restrict__ C, int64_t ldc) {
  // launched over the nt (nt + 1) / 2 lower-triangular tiles only, row by row: no empty workgroups, XCDs evenly loaded
  const int tix = blockIdx.x;
  int bm = (int)((sqrt(8.0 * (double)tix + 1.0) - 1.0) * 0.5);
  while ((bm + 1) * (bm + 2) / 2 <= tix) ++bm;
  while (bm * (bm + 1) / 2 > tix) --bm;
  const int bn = tix - bm * (bm + 1) / 2;
  __shared__ double As[TBK][TLD];
  __shared__ double Bs[TBK][TLD];
  const int t = threadIdx.x, l = t & 63, w = t >> 6;
  const int wm = w >> 1, wn = w & 1;
  const int m0 = bm * TBM, n0 = bn * TBM;
  pta_f64x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = pta_f64x4{0.0, 0.0, 0.0, 0.0};
  for (int k0 = 0; k0 < K; k0 += TBK) {
    int kr = t >> 4, q = (t & 15) * 4;
    int gk = k0 + kr;
    double ph = (gk < K) ? phi[gk] : 0.0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int gi = m0 + q + j, gj = n0 + q + j;
      As[kr][q + j] = (gk < K && gi < N) ? ph * Ft[(int64_t)gk * ldf + gi] : 0.0;
      Bs[kr][q + j] = (gk < K && gj < N) ? Ft[(int64_t)gk * ldf + gj] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < TBK; kk += 4) {
      double a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = As[kk + (l >> 4)][wm * 32 + i * 16 + (l & 15)];
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = Bs[kk + (l >> 4)][wn * 32 + j * 16 + (l & 15)];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = pta_mfma_f64(a[i], b[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int row = m0 + wm * 32 + i * 16 + pta_mfma_row(l, r);
        int col = n0 + wn * 32 + j * 16 + pta_mfma_col(l);
        if (row < N && col < N && col <= row) {
          double v = acc[i][j][r];
          if (row == col) v = v + sigma2[row];
          if (epoch_of && epoch_of[row] == epoch_of[col]) v = v + ecorr2[row];
          C[(int64_t)row * ldc + col] = v;
        }
      }
}

extern "C" int pta_td_cov_assemble(const double *Ft, int64_t ldf, int N, int K, const double *phi, const double *sigma2,
                                   const int32_t *epoch_of, const double *ecorr2, double *C, int64_t ldc, void *stream) {
  PTA_REQUIRE(sigma2 && C && (K == 0 || (Ft && phi)), PTA_E_ARG, "pta_td_cov_assemble: NULL argument");
  PTA_REQUIRE(!epoch_of || ecorr2, PTA_E_ARG, "pta_td_cov_assemble: ecorr2 missing");
  PTA_REQUIRE(N > 0 && K >= 0 && (K == 0 || ldf >= N) && ldc >= N && pta_cdiv(N, TBM) <= 65535u, PTA_E_ARG, "pta_td_cov_assemble: N=%d K=%d", N, K);
  unsigned nt = pta_cdiv(N, TBM);
  hipLaunchKernelGGL(k_td_cov, dim3(nt * (nt + 1) / 2), dim3(256), 0, pta_stream(stream), Ft, ldf, N, K, phi, sigma2, epoch_of, ecorr2, C,
                     ldc);
  PTA_LAUNCH_CHECK();
  return PTA_OK;
}

// The same assembly for ALL pulsars of an array in one launch, 128 x 128 tiles (4 waves as 2 x 2, each 64 x 64 = 4 x 4 MFMA
// tiles), K slabs of 16 double-buffered in LDS: 4x fewer workgroups and operand loads per output than the 64 x 64 kernel above,
// no tail between 68 per-pulsar launches.  grid = (lower-triangular tiles of the largest block, blocks).  Both operand slabs
// are rows of the K-major design matrix (coalesced, conflict-free LDS stores: 16 lanes write 16 consecutive doubles).
#define TC_T 128
#define TC_LD 144  // == 16 (mod 32)
// TM = rows of the output tile: 128 (round 2) or 64.  The kernel is a short product (K = 60: four slabs) in front of a long
// epilogue (epoch / variance vectors, LDS staging, 64-128 KB of stores that must be acknowledged before the workgroup retires), so
// what it needs is workgroups in DIFFERENT phases on a CU: the 64-row tile halves the accumulators (64 instead of 128 VGPRs) and the
// operand slabs, which lets a third workgroup in - one's stores drain under the others' MFMAs.
template <int TM>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(TM == 128 ? 2 : 3, TM == 128 ? 2 : 3))) void k_td_cov128(const double *__restrict__ Ft, int64_t ldf, int K, const double *__restrict__ phi,
                                                      const double *__restrict__ sigma2, const int32_t *__restrict__ epoch_of,
                                                      const double *__restrict__ ecorr2, double *__restrict__ Cbase,
                                                      const int64_t *__restrict__ blk_pos, const int32_t *__restrict__ blk_ld,
                                                      const int32_t *__restrict__ blk_n, const int32_t *__restrict__ blk_off) {
  constexpr int TMI = TM / 32;          // 16-row MFMA tiles per wave (waves as 2 x 2: each TM / 2 rows x 64 columns)
  constexpr int ALD = TM + 16;          // A slab pitch (doubles), == 16 (mod 32)
  constexpr int NBUF = TM == 128 ? 2 : 1;  // 64-row tiles: single-buffered slabs (34 KB of LDS with the staging tile: the third workgroup
                                           // must fit beside the other two's 160 KB; its extra barrier per slab is what they overlap)
  const int blk = blockIdx.y;
  const int N = blk_n[blk];
  const int tix = blockIdx.x;
  int bm, bn;
  if (TM == 128) {  // lower-triangular tiles, row by row: tix = bm (bm + 1) / 2 + bn
    bm = (int)((sqrt(8.0 * (double)tix + 1.0) - 1.0) * 0.5);
    while ((bm + 1) * (bm + 2) / 2 <= tix) ++bm;
    while (bm * (bm + 1) / 2 > tix) --bm;
    bn = tix - bm * (bm + 1) / 2;
  } else {          // 64-row blocks 2 p and 2 p + 1 both meet column tiles 0 .. p: tix = p (p + 1) + r, r < 2 (p + 1)
    int p = (int)((sqrt(4.0 * (double)tix + 1.0) - 1.0) * 0.5);
    while ((p + 1) * (p + 2) <= tix) ++p;
    while (p * (p + 1) > tix) --p;
    const int r = tix - p * (p + 1);
    bm = 2 * p + (r > p ? 1 : 0);
    bn = r > p ? r - (p + 1) : r;
  }
  const int m0 = bm * TM, n0 = bn * TC_T;
  if (m0 >= N) return;  // a block smaller than the largest one
  const int64_t off = blk_off[blk];
  const int64_t ldc = blk_ld[blk];
  double *__restrict__ C = Cbase + blk_pos[blk];
  const double *__restrict__ F = Ft + off;
  const double *__restrict__ ph = phi ? phi + (int64_t)blk * K : nullptr;
  // operand slabs As[2][16][ALD], Bs[2][16][144]; reused as the output staging tile [TM / 2][132]
  constexpr int SLABS = NBUF * TBK * ALD + NBUF * TBK * TC_LD, STAGE = (TM / 2) * (TC_T + 4);
  __shared__ double smem[SLABS > STAGE ? SLABS : STAGE];
  double (*As)[TBK][ALD] = reinterpret_cast<double (*)[TBK][ALD]>(smem);
  double (*Bs)[TBK][TC_LD] = reinterpret_cast<double (*)[TBK][TC_LD]>(smem + NBUF * TBK * ALD);
  const int t = threadIdx.x, l = t & 63, w = t >> 6;
  const int wm = w >> 1, wn = w & 1;
  pta_f64x4 acc[TMI][4];
#pragma unroll
  for (int i = 0; i < TMI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = pta_f64x4{0.0, 0.0, 0.0, 0.0};
  const int kr = t >> 4, cq = t & 15;  // slab row (k) and column phase of this thread; columns cq + 16 j
  double ra[TM / 16], rb[8];
  // every load unconditional, from a clamped (always valid) address: a bin beyond K enters with phi = 0, a row / column beyond N
  // lands in a part of the tile that is never stored.  (Predicated, each of the 17 loads of a slab sat in its own exec-masked
  // block behind an s_waitcnt vmcnt(0): 17 serialised L2 round trips per slab.)
  // (the scaling by phi happens in stash(), after the MFMAs of the current slab: nothing consumes a load inside fetch())
  double pk = 0.0;
  bool pk_in = false;
  auto fetch = [&](int k0) {
    const int gk = k0 + kr, gkc = min(gk, K - 1);
    pk = ph[gkc];
    pk_in = gk < K;
    const double *__restrict__ Fk = F + (int64_t)gkc * ldf;
#pragma unroll
    for (int j = 0; j < TM / 16; ++j) ra[j] = Fk[min(m0 + cq + 16 * j, N - 1)];
#pragma unroll
    for (int j = 0; j < 8; ++j) rb[j] = Fk[min(n0 + cq + 16 * j, N - 1)];
  };
  auto stash = [&](int buf) {
    const double p = pk_in ? pk : 0.0;
#pragma unroll
    for (int j = 0; j < TM / 16; ++j) As[buf][kr][cq + 16 * j] = p * ra[j];
#pragma unroll
    for (int j = 0; j < 8; ++j) Bs[buf][kr][cq + 16 * j] = rb[j];
  };
  const int nslab = (K + TBK - 1) / TBK;
  if (nslab > 0) {
    fetch(0);
    stash(0);
  }
  __syncthreads();
  for (int sidx = 0; sidx < nslab; ++sidx) {
    const int cur = NBUF == 2 ? (sidx & 1) : 0;
    if (sidx + 1 < nslab) fetch((sidx + 1) * TBK);
#pragma unroll
    for (int kk = 0; kk < TBK; kk += 4) {
      double a[TMI], b[4];
#pragma unroll
      for (int i = 0; i < TMI; ++i) a[i] = As[cur][kk + (l >> 4)][wm * (TM / 2) + i * 16 + (l & 15)];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = Bs[cur][kk + (l >> 4)][wn * 64 + j * 16 + (l & 15)];
#pragma unroll
      for (int i = 0; i < TMI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = pta_mfma_f64(a[i], b[j], acc[i][j]);
    }
    if (NBUF == 1) __syncthreads();  // every wave has read the slab before it is replaced
    if (sidx + 1 < nslab) stash(NBUF == 2 ? (cur ^ 1) : 0);
    __syncthreads();
  }
  // epilogue.  The white and ECORR terms are added in registers (the epochs of the rows and 4 columns a lane owns are read
  // once - here, behind the last MFMA: requesting them at the top of the kernel and parking them in 1.8 KB of LDS measured 3.22
  // against 2.98 ms, holding them in registers across the product spills); the tile then goes through LDS so that every store instruction of a wave writes ONE whole 1 KB row segment (64 lanes
  // x 16 bytes) instead of four 128-byte pieces of four different rows: DRAM pages are opened once per row, not per piece.
  int erow[TMI][4], ecol[4];
#pragma unroll
  for (int i = 0; i < TMI; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = m0 + wm * (TM / 2) + i * 16 + pta_mfma_row(l, r);
      erow[i][r] = (epoch_of && row < N) ? epoch_of[off + row] : -2;
    }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int col = n0 + wn * 64 + j * 16 + pta_mfma_col(l);
    ecol[j] = (epoch_of && col < N) ? epoch_of[off + col] : -1;
  }
  const bool diag_tile = n0 <= m0 + TM - 1 && n0 + TC_T - 1 >= m0;  // workgroup-uniform
#pragma unroll
  for (int i = 0; i < TMI; ++i) {
    double s2[4], e2[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rowc = min(m0 + wm * (TM / 2) + i * 16 + pta_mfma_row(l, r), N - 1);
      s2[r] = diag_tile ? sigma2[off + rowc] : 0.0;
      e2[r] = epoch_of ? ecorr2[off + rowc] : 0.0;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = m0 + wm * (TM / 2) + i * 16 + pta_mfma_row(l, r);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int col = n0 + wn * 64 + j * 16 + pta_mfma_col(l);
        double add = (row == col) ? s2[r] : 0.0;
        if (col <= row && erow[i][r] == ecol[j]) add += e2[r];
        acc[i][j][r] = acc[i][j][r] + add;
      }
    }
  }
  constexpr int SLD = TC_T + 4;                      // staging pitch (doubles): [TM / 2][132] fits inside the slabs
  double (*S)[SLD] = reinterpret_cast<double (*)[SLD]>(smem);
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    if (wm == half) {
#pragma unroll
      for (int i = 0; i < TMI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) S[i * 16 + pta_mfma_row(l, r)][wn * 64 + j * 16 + pta_mfma_col(l)] = acc[i][j][r];
    }
    __syncthreads();
    for (int rr = w; rr < TM / 2; rr += 4) {         // wave w stores rows w, w + 4, ... of this half; lane l columns 2l, 2l + 1
      const int row = m0 + half * (TM / 2) + rr;
      if (row >= N) break;
      const int col = n0 + 2 * l;
      const double2 v = *reinterpret_cast<const double2 *>(&S[rr][2 * l]);
      double *__restrict__ dst = C + (int64_t)row * ldc + col;
      if (!diag_tile)                                    // strictly below the diagonal: every column of the row is stored
        *reinterpret_cast<double2 *>(dst) = v;          // ldc and n0 are even, the block base 16-byte aligned
      else if (col + 1 <= row)
        *reinterpret_cast<double2 *>(dst) = v;
      else if (col == row)
        dst[0] = v.x;
    }
    __syncthreads();
  }
}

// ---- column-walking assembly (round 5) ---------------------------------------------------------------------------------------------
// The tile kernel above is a short product (K = 60: four slabs) between a latency chain (block table -> operand rows -> LDS -> barrier)
// and 64 KB of stores that must be acknowledged before the workgroup retires: per workgroup 3.4 us of matrix-pipe work in a 20 us
// lifetime, three workgroups per CU -> 53 % MFMA-busy, 2.3 TB/s written.  Here a wave OWNS 64 columns of one pulsar's covariance for a
// segment of TCW_SEG rows: its B operand - phi_k F[k, col], 4 column tiles x NKS k-steps - is loaded ONCE into registers (120 VGPRs at
// K = 60) and stays there while the wave walks down the rows 16 at a time.  Per step: the A fragments of the NEXT step are requested
// straight from the K-major design matrix (a lane's value F[4 ks + (l >> 4), row0 + (l & 15)]: four 128-byte row pieces per load, L1 /
// L2 hits - a pulsar's F is 2.4 MB), 4 x NKS MFMAs, then the white / ECORR terms are added in the accumulators' own layout and the 16 x 64
// result goes straight to memory: the column tiles are interleaved in pairs (see the B operand), so a lane owns two NEIGHBOURING columns
// and a store instruction writes four rows x 256 bytes of whole cache lines.  No LDS staging, no barrier inside the walk; the stores of
// step s drain under the products of the following steps.
//
// Round 4's first column-walking kernel (32 columns per wave, 128-byte store pieces; commit ee73ef5, withdrawn in 18da960) "produced NaN
// until a device-wide synchronisation" in one test sequence.  Cause (round 5; scripts/gpu_r5_nan_repro.py reproduces it on demand): for
// K = 58 (components = 29: the failing case) its A-fragment loads of the cut k-step read rows k = 58, 59 of the [K, N] design matrix -
// BEHIND its end - on the strength of "a masked k meets b = 0".  0 x finite = 0, but 0 x NaN = NaN: whenever the allocator had placed
// a recycled NaN-poisoned block (the test's own d_Ltd.fill_(nan) clones) behind Ft, the accumulators turned NaN; any change of the
// allocation history - a synchronisation with its frees included - moved finite bytes there and the symptom vanished.  No ordering
// hazard was involved.  Here every k index is clamped to K - 1 (finite data) before it forms an address, and the tests run the kernel
// on an Ft that is a view into a NaN-filled slab.
#define TCW_SEG_MAX 512    // rows per work item (1024 measured 3 % slower: profiles/r05_tcw_diag.txt)
// NT = 16-column tiles a wave owns (4: 64 columns, two workgroups per CU, 248 of 256 VGPRs).  The A fragments are requested ONE step ahead,
// each into the register just consumed.  (On gfx950 loads and stores share one counter, vmcnt, and complete out of order with respect to
// each other, so the wait in front of a group of products can only be "at most as many operations outstanding as there are YOUNGER LOADS" -
// 14 here - and outstanding stores count against that allowance.  A form with 128 columns per wave and the fragments two steps ahead (one
// workgroup per CU, 506 registers, vmcnt(29)) measured 2.67 against 2.33 ms: a single wave per SIMD exposes every stall.)
// DIAG (probe builds only, -DPTA_TCW_DIAG; results are WRONG by construction): 1 = no global stores, 2 = no fragment loads inside the
// steps, 3 = no products - which of the three streams bounds the kernel (scripts/gpu_r5_tcw_diag.py)
template <int NKS, bool EP, int NT, int DIAG = 0>
__global__ __launch_bounds__(256, NT == 4 ? 2 : 1) void k_td_cov_walk(const double *__restrict__ Ft, int64_t ldf, int K, const double *__restrict__ phi,
                                                        const double *__restrict__ sigma2, const int32_t *__restrict__ epoch_of,
                                                        const double *__restrict__ ecorr2, double *__restrict__ Cbase,
                                                        const int64_t *__restrict__ blk_pos, const int32_t *__restrict__ blk_ld,
                                                        const int32_t *__restrict__ blk_n, const int32_t *__restrict__ blk_off,
                                                        const int32_t *__restrict__ item0, int n_blocks, int seg_rows, const int32_t *__restrict__ epoch_first) {
  constexpr int COLS = 16 * NT;        // columns per wave
  constexpr int WGC = 4 * COLS;        // columns per workgroup
  static_assert(NT % 2 == 0, "column tiles come in pairs (a lane's two neighbouring columns)");
  __shared__ double __attribute__((aligned(16))) Rinfo[TCW_SEG_MAX][2];  // (ecorr2, sigma2) of the segment's rows
  __shared__ double Repoch[TCW_SEG_MAX];                                 // their epochs (int32 -> double is exact)
  __shared__ int Rlo16[TCW_SEG_MAX / 16];                                // per 16-row step: the smallest column that shares an epoch with one of its rows
  // work item -> (pulsar, WGC-column group, row segment); item0[b] = first item of pulsar b (host: pta_td_cov_walk_items)
  int blk = 0;
  {
    int lo = 0, hi = n_blocks;  // largest blk with item0[blk] <= blockIdx.x
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (item0[mid] <= (int)blockIdx.x) lo = mid; else hi = mid;
    }
    blk = lo;
  }
  const int N = blk_n[blk];
  int item = (int)blockIdx.x - item0[blk], cb = 0;
  for (;; ++cb) {  // workgroup-uniform
    const int ns = (N - WGC * cb + seg_rows - 1) / seg_rows;
    if (item < ns) break;
    item -= ns;
    if (WGC * (cb + 1) >= N) return;  // (cannot happen with a consistent item table)
  }
  const int t = threadIdx.x, l = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6), li = l & 15, lq = l >> 4;
  const int c0 = WGC * cb + COLS * w;  // this wave's columns
  const int rbeg = WGC * cb + seg_rows * item, rend = min(N, rbeg + seg_rows);
  const int64_t off = blk_off[blk];
  // the white / ECORR terms of the segment's rows -> LDS, once per workgroup (its four waves walk the same rows): the kernel's ONLY
  // workgroup barrier; the steps then issue nothing but fragment loads and stores
  for (int rr = t; rr < seg_rows; rr += 256) {
    const int row = min(rbeg + rr, N - 1);
    Rinfo[rr][0] = EP ? ecorr2[off + row] : 0.0;
    Rinfo[rr][1] = sigma2[off + row];
    Repoch[rr] = EP ? (double)epoch_of[off + row] : -2.0;
  }
  // ECORR couples a row only to the columns of its own epoch.  epoch_first[i] = the smallest TOA index (inside the pulsar) that shares TOA
  // i's epoch: a step whose rows all have epoch_first > the wave's last column cannot meet an epoch partner there and skips the sixteen
  // comparisons and conditional adds of its epilogue (with time-ordered TOAs that is every step off the diagonal; unsorted TOAs simply
  // keep the arithmetic where a partner may lie).  NULL: every step keeps it.
  if (EP && t < seg_rows / 16) {
    int m = 0;
    if (epoch_first) {
      m = 0x7fffffff;
      for (int j = 0; j < 16; ++j) {
        const int row = rbeg + 16 * t + j;
        if (row < N) m = min(m, epoch_first[off + row]);
      }
    }
    Rlo16[t] = m;
  }
  __syncthreads();
  if (c0 >= N) return;                      // no barrier below: a wave may leave alone
  const int64_t ldc = blk_ld[blk];
  double *__restrict__ C = Cbase + blk_pos[blk];
  const double *__restrict__ F = Ft + off;
  const double *__restrict__ ph = phi + (int64_t)blk * K;
  // k rows of this lane: 4 ks + lq.  EVERY k is clamped to K - 1 BEFORE it forms an address (finite data meets b = 0; K <= 4 NKS, and a
  // small K may leave whole k-steps past it).  A k-step that lies wholly inside K takes a wave-uniform base (SGPRs) + the lane's 32-bit
  // element offset lq ldf + x (the host checks 64 ldf < 2^29): no 64-bit per-lane pointers in registers, one offset per step for all k-steps.
  // (addresses = wave-uniform base + 32-bit per-lane BYTE offset: the form global_load / global_store take with a scalar base register,
  // no 64-bit address arithmetic on the vector ALU inside the walk)
  const uint32_t lqo = (uint32_t)lq * (uint32_t)ldf * 8u;
  auto f_at = [&](int ks, uint32_t x) -> double {  // F[min(4 ks + lq, K - 1), x]
    const bool whole = 4 * ks + 3 < K;             // kernel-uniform
    const char *__restrict__ base = reinterpret_cast<const char *>(whole ? F + (int64_t)(4 * ks) * ldf : F);
    const uint32_t o = whole ? lqo : (uint32_t)min(4 * ks + lq, K - 1) * (uint32_t)ldf * 8u;
    return *reinterpret_cast<const double *>(base + (o + 8u * x));
  };
  // resident B operand: lane holds B[k = 4 ks + lq][j = li] = phi_k F[k, col(jt, li)]; k >= K enters as zero.  The sixteen columns of MFMA
  // tile jt are NOT sixteen neighbours: col(jt, li) = c0 + 32 (jt >> 1) + 2 li + (jt & 1) - tiles 2 p and 2 p + 1 interleave, so that a lane's
  // accumulators (row q + 4 reg, its column of tile jt) hold TWO NEIGHBOURING columns per tile pair: the result goes from the accumulators
  // straight to memory as 16-byte stores, sixteen lanes = one 256-byte piece of a row, four rows per instruction - whole 128-byte lines,
  // no transposition through LDS (pta_microbench kind 7: HBM takes row pieces of 128 bytes ... 4 KB at the same 5.4-5.6 TB/s).
  double b[NT][NKS];
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    const double p = (4 * ks + lq < K) ? ph[min(4 * ks + lq, K - 1)] : 0.0;
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) b[jt][ks] = p * f_at(ks, (uint32_t)min(c0 + 32 * (jt >> 1) + 2 * li + (jt & 1), N - 1));
  }
  // store side: lane (lq, li) -> rows lq + 4 reg of a step, columns c0 + 32 p + 2 li, + 1 for p < NT / 2
  const int scol0 = c0 + 2 * li;
  const uint32_t sto = ((uint32_t)lq * (uint32_t)ldc + (uint32_t)scol0) * 8u;  // byte offset of the lane's first column in row r0 + lq from row r0's start
  double ec[NT / 2][2];  // epochs of the lane's columns, as doubles (int32 -> double is exact); -1 past the block
#pragma unroll
  for (int pp = 0; pp < NT / 2; ++pp)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int col = scol0 + 32 * pp + h;
      ec[pp][h] = (EP && col < N) ? (double)epoch_of[off + col] : -1.0;
    }
  typedef double f64x2 __attribute__((ext_vector_type(2)));
  const int rfirst = max(rbeg, c0);  // rows above the wave's own columns are not in the lower triangle (both multiples of 16)
  if (rfirst >= rend) return;
  double a[NKS];
  {
    const uint32_t x = (uint32_t)min(rfirst + li, N - 1);
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) a[ks] = f_at(ks, x);
  }
  auto step = [&](const int r0) {  // 16 rows x COLS columns; r0 wave-uniform
    const bool diag_step = r0 < c0 + COLS;
    int lo16 = 0;
    if (EP && !diag_step) lo16 = Rlo16[(r0 - rbeg) >> 4];         // (requested here, used behind the products)
    const uint32_t xn = (uint32_t)min(r0 + 16 + li, N - 1);  // the next step's rows (past the segment: clamped, unused)
    pta_f64x4 acc[NT];
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) acc[jt] = pta_f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
#pragma unroll
      for (int jt = 0; jt < NT; ++jt) {
        if (DIAG == 3) acc[jt][ks & 3] = fma(a[ks], b[jt][ks], acc[jt][ks & 3]);
        else acc[jt] = pta_mfma_f64(a[ks], b[jt][ks], acc[jt]);
      }
      if (DIAG != 2 && DIAG != 5) a[ks] = f_at(ks, xn);  // the next step's fragment into the register just consumed
      __builtin_amdgcn_sched_barrier(0);  // (left alone, the scheduler gathers the loads behind the twelfth group of products)
    }
    if (DIAG == 6) {  // the epilogue as a pure delay of its own length (~600 cycles): is a wave's gap filled by the SIMD's other wave?
      __builtin_amdgcn_s_sleep(9);
    }
    if (DIAG == 4 || DIAG == 5 || DIAG == 6) {  // no epilogue at all: the products (and, 4, their fragment loads) alone
      double sacc = 0.0;
#pragma unroll
      for (int jt = 0; jt < NT; ++jt) sacc += acc[jt][0] + acc[jt][1] + acc[jt][2] + acc[jt][3];
      if (sacc == 1.2345e300) C[0] = sacc;
      return;
    }
    // epilogue: white / ECORR terms added in the accumulators' own layout, then 16-byte stores (see the B operand above)
    if (!diag_step && r0 + 16 <= N && (!EP || __builtin_amdgcn_readfirstlane(lo16) >= c0 + COLS)) {
      // wave-uniform: strictly below the diagonal block, all 16 rows exist, no row has an epoch partner among the wave's columns: the
      // accumulators ARE the result
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        char *__restrict__ rowb = reinterpret_cast<char *>(C + (int64_t)(r0 + 4 * reg) * ldc);  // wave-uniform
#pragma unroll
        for (int pp = 0; pp < NT / 2; ++pp) {
          const f64x2 v = {acc[2 * pp][reg], acc[2 * pp + 1][reg]};
          if (DIAG != 1) *reinterpret_cast<f64x2 *>(rowb + (sto + 256u * pp)) = v;
          else if (v.x == 1.2345e300) C[0] = v.y;
        }
      }
    } else if (!diag_step && r0 + 16 <= N) {  // the same with the ECORR term (nothing is predicated)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int row = r0 + lq + 4 * reg;
        double e2 = 0.0, er = -2.0;
        if (EP) {
          e2 = Rinfo[row - rbeg][0];   // (four distinct rows per wave: broadcast reads)
          er = Repoch[row - rbeg];
        }
        double *__restrict__ dst = C + (int64_t)row * ldc + scol0;
#pragma unroll
        for (int pp = 0; pp < NT / 2; ++pp) {
          f64x2 v = {acc[2 * pp][reg], acc[2 * pp + 1][reg]};
          if (EP) {
            v.x = er == ec[pp][0] ? v.x + e2 : v.x;
            v.y = er == ec[pp][1] ? v.y + e2 : v.y;
          }
          if (DIAG != 1) *reinterpret_cast<f64x2 *>(dst + 32 * pp) = v;
          else if (v.x == 1.2345e300) C[0] = v.y;
        }
      }
    } else {
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int row = r0 + lq + 4 * reg, rr = min(row, rend - 1) - rbeg;
        const f64x2 rt = *reinterpret_cast<const f64x2 *>(&Rinfo[rr][0]);  // (ecorr2[row], sigma2[row])
        const double er = Repoch[rr];
        double *__restrict__ dst = C + (int64_t)row * ldc + scol0;
#pragma unroll
        for (int pp = 0; pp < NT / 2; ++pp) {
          const int col = scol0 + 32 * pp;
          f64x2 v = {acc[2 * pp][reg], acc[2 * pp + 1][reg]};
          if (er == ec[pp][0] && col <= row) v.x = v.x + rt.x;
          if (er == ec[pp][1] && col + 1 <= row) v.y = v.y + rt.x;
          if (col == row) v.x = v.x + rt.y;
          if (col + 1 == row) v.y = v.y + rt.y;
          if (row < N) {
            if (col + 1 <= row)
              *reinterpret_cast<f64x2 *>(dst + 32 * pp) = v;
            else if (col == row)
              dst[32 * pp] = v.x;
          }
        }
      }
    }
  };
  // first step peeled: inside the loop the wait in front of a group of products then counts the fragment loads behind the one it needs as
  // YOUNGER operations; merged with the kernel's prologue it would be stricter
  step(rfirst);
  for (int r0 = rfirst + 16; r0 < rend; r0 += 16) step(r0);
}

// variant of the column-walking kernel: 0 / 1 = 64 columns per wave, two workgroups per CU, fragments one step ahead (the only form built).
// Measured and not kept (profiles/r05_tcw_diag.txt): 128 columns per wave with fragments two steps ahead (one
// workgroup per CU, 506 registers, vmcnt(29) waits) - 2.67 against 2.33 ms, a single wave per SIMD exposes every stall; 64 columns with
// two register sets does not fit 256 registers (35 spilled).
static inline int pta_tcw_variant(int v, int K) { (void)v; (void)K; return 1; }
static inline int pta_tcw_wgcols(int v, int K) { (void)v; (void)K; return 256; }
static inline int pta_tcw_seg(int v) { (void)v; return 512; }  // rows per work item (measured: 256 -> same, 1024 -> 3 % slower)

// number of work items of the column-walking kernel per block and in all: item0[b] = first item of block b, item0[n_blocks] = total
extern "C" int64_t pta_td_cov_walk_items(const int32_t *blk_n_host, int n_blocks, int K, int variant, int32_t *item0_host) {
#ifdef PTA_TCW_DIAG
  variant &= 15;
#endif
  if (!blk_n_host || !item0_host || n_blocks <= 0 || variant < 0 || variant > 1) return -1;
  const int wgc = pta_tcw_wgcols(variant, K), seg = pta_tcw_seg(variant);
  int64_t tot = 0;
  for (int b = 0; b < n_blocks; ++b) {
    item0_host[b] = (int32_t)tot;
    const int n = blk_n_host[b];
    for (int cb = 0; wgc * cb < n; ++cb) tot += (n - wgc * cb + seg - 1) / seg;
    if (tot >= (1LL << 31)) return -1;
  }
  item0_host[n_blocks] = (int32_t)tot;
  return tot;
}

static int pta_td_cov_walk_nks(int K) { return K <= 0 || K > 64 ? 0 : (K > 56 && K <= 60 ? 15 : 4 * ((K + 15) / 16)); }

extern "C" int pta_td_cov_assemble_walk(const double *Ft, int64_t ldf, int K, const double *phi, const double *sigma2,
                                        const int32_t *epoch_of, const double *ecorr2, double *Cbase, const int64_t *blk_pos,
                                        const int32_t *blk_ld, const int32_t *blk_n, const int32_t *blk_off, int n_blocks,
                                        const int32_t *item0, int64_t n_items, const int32_t *epoch_first, int variant, void *stream) {
  PTA_REQUIRE(Ft && phi && sigma2 && Cbase && blk_pos && blk_ld && blk_n && blk_off && item0, PTA_E_ARG, "pta_td_cov_assemble_walk: NULL argument");
  PTA_REQUIRE(!epoch_of || ecorr2, PTA_E_ARG, "pta_td_cov_assemble_walk: ecorr2 missing");
  PTA_REQUIRE(n_blocks > 0 && n_blocks <= 65535 && n_items > 0 && n_items < (1LL << 31), PTA_E_ARG, "pta_td_cov_assemble_walk: n_blocks=%d n_items=%lld",
              n_blocks, (long long)n_items);
  PTA_REQUIRE(((uintptr_t)Cbase % 16) == 0, PTA_E_ARG, "pta_td_cov_assemble_walk: Cbase must be 16-byte aligned (blk_pos and blk_ld even)");
#ifdef PTA_TCW_DIAG
  const int diag = variant >> 4;
  variant &= 15;
#endif
  PTA_REQUIRE(variant >= 0 && variant <= 1, PTA_E_ARG, "pta_td_cov_assemble_walk: variant=%d (0 or 1)", variant);
  const int nks = pta_td_cov_walk_nks(K);
  PTA_REQUIRE(nks > 0, PTA_E_ARG, "pta_td_cov_assemble_walk: needs 1 <= K <= 64 (K=%d): use pta_td_cov_assemble_all", K);
  PTA_REQUIRE(ldf > 0 && 64 * ldf < (1LL << 29), PTA_E_ARG, "pta_td_cov_assemble_walk: ldf=%lld too large for 32-bit operand offsets", (long long)ldf);
  const int v = pta_tcw_variant(variant, K);
#define PTA_TCW_LAUNCH2(NKSV, EPV, NTV)                                                                                              \
  hipLaunchKernelGGL((k_td_cov_walk<NKSV, EPV, NTV>), dim3((unsigned)n_items), dim3(256), 0, pta_stream(stream), Ft, ldf, K, phi, sigma2, \
                     epoch_of, ecorr2, Cbase, blk_pos, blk_ld, blk_n, blk_off, item0, n_blocks, pta_tcw_seg(variant), epoch_first)
#define PTA_TCW_LAUNCH1(NKSV, EPV) PTA_TCW_LAUNCH2(NKSV, EPV, 4)
#define PTA_TCW_LAUNCH(NKSV)      \
  if (epoch_of) {                 \
    PTA_TCW_LAUNCH1(NKSV, true);  \
  } else {                        \
    PTA_TCW_LAUNCH1(NKSV, false); \
  }
#ifdef PTA_TCW_DIAG
  if (diag && nks == 15 && epoch_of) {
#define PTA_TCW_D(NTV, DG) hipLaunchKernelGGL((k_td_cov_walk<15, true, NTV, DG>), dim3((unsigned)n_items), dim3(256), 0, pta_stream(stream), Ft, ldf, K, phi, sigma2, epoch_of, ecorr2, Cbase, blk_pos, blk_ld, blk_n, blk_off, item0, n_blocks, pta_tcw_seg(variant), epoch_first)
    { if (diag == 1) PTA_TCW_D(4, 1); else if (diag == 2) PTA_TCW_D(4, 2); else if (diag == 3) PTA_TCW_D(4, 3); else if (diag == 4) PTA_TCW_D(4, 4); else if (diag == 5) PTA_TCW_D(4, 5); else PTA_TCW_D(4, 6); }
#undef PTA_TCW_D
    PTA_LAUNCH_CHECK();
    return PTA_OK;
  }
#endif
  switch (nks) {
    case 4: PTA_TCW_LAUNCH(4); break;
    case 8: PTA_TCW_LAUNCH(8); break;
    case 12: PTA_TCW_LAUNCH(12); break;
    case 15: PTA_TCW_LAUNCH(15); break;
    default: PTA_TCW_LAUNCH(16); break;
  }
#undef PTA_TCW_LAUNCH
#undef PTA_TCW_LAUNCH1
#undef PTA_TCW_LAUNCH2
  PTA_LAUNCH_CHECK();
  return PTA_OK;
}

extern "C" int pta_td_cov_assemble_all(const double *Ft, int64_t ldf, int K, const double *phi, const double *sigma2,
                                       const int32_t *epoch_of, const double *ecorr2, double *Cbase, const int64_t *blk_pos,
                                       const int32_t *blk_ld, const int32_t *blk_n, const int32_t *blk_off, int n_blocks, int max_n,
                                       void *stream) {
  PTA_REQUIRE(sigma2 && Cbase && blk_pos && blk_ld && blk_n && blk_off && (K == 0 || (Ft && phi)), PTA_E_ARG,
              "pta_td_cov_assemble_all: NULL argument");
  PTA_REQUIRE(!epoch_of || ecorr2, PTA_E_ARG, "pta_td_cov_assemble_all: ecorr2 missing");
  PTA_REQUIRE(n_blocks > 0 && n_blocks <= 65535 && max_n > 0 && K >= 0, PTA_E_ARG, "pta_td_cov_assemble_all: n_blocks=%d max_n=%d K=%d",
              n_blocks, max_n, K);
  PTA_REQUIRE(((uintptr_t)Cbase % 16) == 0, PTA_E_ARG, "pta_td_cov_assemble_all: Cbase must be 16-byte aligned (blk_pos and blk_ld even)");
  const int64_t nt = pta_cdiv(max_n, TC_T);
  PTA_REQUIRE(nt * (nt + 1) < (1LL << 31), PTA_E_ARG, "pta_td_cov_assemble_all: max_n=%d too large", max_n);
  // 64-row tiles (row blocks 2 p, 2 p + 1 x column tiles 0 .. p), three workgroups per CU: 2.97 ms for the 68 x 5000^2 lower triangles
  // against 3.22 ms with round 2's 128-row tiles at two per CU (k_td_cov128<128>, same source)
  hipLaunchKernelGGL(k_td_cov128<64>, dim3((unsigned)(nt * (nt + 1)), n_blocks), dim3(256), 0, pta_stream(stream), Ft, ldf, K, phi, sigma2,
                     epoch_of, ecorr2, Cbase, blk_pos, blk_ld, blk_n, blk_off);
  PTA_LAUNCH_CHECK();
  return PTA_OK;
}

// out[r, i] (+)= sum_j z[r, j] L[i, j]  =  (Z . L^T)[r, i]; L's strict upper triangle is zero.
extern "C" int pta_td_trmm(const double *L, int64_t ldl, int N, const double *z, int64_t ld_z, int R, double *out, int64_t ld_out,
                           int accumulate, int algo, void *stream) {
  PTA_REQUIRE(L && z && out, PTA_E_ARG, "pta_td_trmm: NULL argument");
  PTA_REQUIRE(N > 0 && R > 0 && ldl >= N && ld_z >= N && ld_out >= N, PTA_E_ARG, "pta_td_trmm: N=%d R=%d", N, R);
  return pta_dgemm_launch(1, R, N, N, 1.0, z, ld_z, 1, L, ldl, accumulate ? 1.0 : 0.0, out, ld_out, 0, 1, 0, 0, 0, algo ? 1 : 0,
                          pta_stream(stream));
}

// ---- L . z with the deviates drawn in registers -------------------------------------------------------------------
// The draw of the dense path: out[m, i] = sum_{j <= i} L[i, j] z[m, j], i.e. Z . L^T with Z never stored anywhere - lane l of
// a wave holds the MFMA A operand Z[m = l & 15][k = l >> 4] and simply GENERATES it (Philox + Box-Muller, pta_rng.h), the way
// k_gwb_idft_sym_rng does for the GWB draws.  One workgroup = a strip of TDS_N = 256 rows of one factor (= 256 output
// columns) x TDS_M = 64 rows of Z (16 per wave); the strip of L streams once through LDS in K slabs of 16, double buffered,
// and is shared by the four waves.  Per slab a wave issues 64 MFMAs (16 column tiles x 4 K steps) against two Box-Muller
// pairs per lane (the four K steps use k = 4 q + s, q = l >> 4, so a lane's four deviates of a slab are exactly the pairs
// (k0 >> 1) + 2 q and + 1): ~300 VALU instructions beside 64 x 64 matrix-pipe cycles.  Every deviate is regenerated by each
// strip that needs it (N / 512 times on average) - VALU work the matrix pipe hides - instead of being written to and re-read
// from an [R, N] buffer.
// LDS pitch 260 doubles: the two half-wave lane groups of a ds_read_b64 read rows 4 apart -> 4 * 260 * 2 = 32 (mod 64) dwords,
// i.e. disjoint bank halves; a 16-lane store group writes 16 consecutive doubles = all 32 banks once.
#define TDS_N PTA_TD_STRIP
#define TDS_M 64
#define TDS_K 16
#define TDS_LD 260
#define TDS_NT (TDS_N / 16)
#define TDS_MANY_ITEMS 64  // from this many strips on, whole strips are dealt to the XCDs (see the work item order below)

// Operand path (round 3): the strip of L is brought in by LDS DMA (global_load_lds_dwordx4, 16 bytes per lane straight into LDS; no
// staging registers, no ds_write pass, no s_waitcnt vmcnt(0) in front of it) as a row-major, unpadded slab image - 256 rows x 16 k =
// 128 bytes per row - whose 16-byte chunks are XOR-swizzled per row (pta_gl_f below: the source address carries the swizzle, the DMA
// writes lane-linearly) so that the fragment reads are conflict-free ds_read_b128: with k = 4 q + st a lane's four B values of a
// column tile are 32 contiguous bytes = two reads instead of four ds_read_b64.  Elements above L's diagonal hold scratch (the
// factorisation parks inverses there) and are masked in registers, in the slabs that cross the strip's diagonal block only.
// One barrier per slab; the DMA of slab s + 1 is issued behind the first MFMAs of slab s and waited for at its closing barrier.
#define TDS_ROWB 128                      // bytes per LDS row
__device__ __forceinline__ int pta_td_swz(int row) {
  const int e = (row >> 1) & 7;           // see pta_gemm.hip (k_dgemm_glds128): chunk c of row r lives in slot c ^ f(r)
  return (e & 1) | (((e >> 2) & 1) * 6);
}

// ZMEM: the deviates are READ (pl.z, the same numbers pta_rng_fill_normal writes for stream (TD, pulsar)) instead of generated: lane
// (c, q) needs z[m][k0 + 4 q .. + 3] of its own Z row - 32 contiguous bytes that no other wave uses, so they go from global memory
// straight into the MFMA A operand registers (no LDS), requested one slab ahead.  Same deviates, same MFMA order: bit-identical
// output; what it buys is the matrix pipe's time - two fp64 Box-Muller pairs per lane and slab share the double-precision ALUs with
// the MFMAs - for 8 bytes of traffic per deviate and strip.
template <bool FAST, bool ZMEM>
__global__ __launch_bounds__(256, 2) void k_td_trmm_rng(pta_td_plan pl, uint64_t seed, uint64_t r0, int M, double *__restrict__ out,
                                                        int64_t ld_out) {
  constexpr int fast = FAST ? 1 : 0;
  __shared__ double __attribute__((aligned(256))) Bs[2][TDS_N * TDS_K];  // [stage][row * 16 + k], chunks swizzled per row: 2 x 32 KB
  // work item order: items are sorted by decreasing K extent (host); consecutive workgroups go to the 8 XCDs round-robin, so
  // XCD x takes items x, x + 8, ... (each XCD gets the same mix of long and short strips) and walks the Z-row groups of one
  // item back to back: the strip of L is fetched into ONE L2 and re-read there by the other row groups.
  // With FEW items (the single grid factor of the GWB: 3 strips) that scheme would leave XCDs idle; then every XCD works on
  // every item and takes the Z-row groups mg = xcd (mod 8) instead - the factor is small enough to sit in all eight L2s.
  const int nmg = (M + TDS_M - 1) / TDS_M;
  const int lin = blockIdx.x;
  const int seq = lin >> 3;
  int item, mg;
  if (pl.n_items >= TDS_MANY_ITEMS) {
    item = (seq / nmg) * 8 + (lin & 7);
    mg = seq % nmg;
  } else {
    const int nmg8 = (nmg + 7) >> 3;
    item = seq / nmg8;
    mg = (seq % nmg8) * 8 + (lin & 7);
  }
  if (item >= pl.n_items || mg >= nmg) return;
  const int blk = pl.item_blk[item];
  const int n0 = pl.item_n0[item];
  const int n = pl.blk_n[blk];
  const int64_t ldl = pl.blk_ld[blk];
  const double *__restrict__ L = pl.Lbase + pl.blk_pos[blk];
  const int t = threadIdx.x, l = t & 63, wv = t >> 6;
  const int c = l & 15, q = l >> 4;
  const int srows = pl.item_rows ? pl.item_rows[item] : TDS_N;  // rows of this strip (a factor's partial strip may be its FIRST: pta_td_plan.item_rows)
  const int kend = min(n, n0 + srows);  // L[i, j] = 0 for j > i: columns beyond the strip's last row contribute nothing
  const int nslab = (kend + TDS_K - 1) / TDS_K;
  typedef double pta_f64x2 __attribute__((ext_vector_type(2)));

  // the Z row of this lane's A operand: m = realisation (rows_per_real == 1) or (realisation, pulsar) of the grid factor
  const int m_a = mg * TDS_M + wv * 16 + c;
  const int rpr = pl.rows_per_real;
  const uint64_t real_a = r0 + (uint64_t)(m_a / rpr);
  const uint32_t strm_a = pta_stream_id(pl.stream_kind, (uint32_t)(rpr == 1 ? blk : (m_a % rpr)));

  pta_f64x4 acc[TDS_NT];
#pragma unroll
  for (int j = 0; j < TDS_NT; ++j) acc[j] = pta_f64x4{0.0, 0.0, 0.0, 0.0};

  // DMA side: wave wv stages rows [64 wv, 64 wv + 64) of the strip, 8 rows per instruction (lane -> row l >> 3, slot l & 7).  Rows
  // past the factor's last one re-read it (their outputs are never stored); a chunk past the row's pitch is clamped into it (it
  // lies above the diagonal and is masked).
  const int row0 = n0 + 64 * wv + (l >> 3);  // row of the factor this lane fetches in instruction g: row0 + 8 g, clamped (addresses
                                             // are formed at issue: eight 64-bit pointers would cost 16 VGPRs the accumulators need)
  const int kmax = (int)ldl - 2;
  auto stage = [&](int k0, int st) {
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      // chunk of the row that belongs in slot l & 7: (l & 7) ^ f(row), f from (row >> 1) & 7 = (4 g + (l >> 4)) & 7
      const int e = (4 * g + (l >> 4)) & 7;
      const int kcg = 2 * ((l & 7) ^ ((e & 1) | (((e >> 2) & 1) * 6)));
      char *dst = reinterpret_cast<char *>(&Bs[st][0]) + (64 * wv + 8 * g) * TDS_ROWB;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(L + (int64_t)min(row0 + 8 * g, n - 1) * ldl + min(k0 + kcg, kmax)),
                                       (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
    }
  };
  // fragment side: column tile j = rows 16 j + c of the strip; chunks 2 q (k = 4 q, 4 q + 1) and 2 q + 1 (k = 4 q + 2, 4 q + 3)
  const int fsw = pta_td_swz(c);
  const int offc = c * TDS_ROWB, c0 = ((2 * q) ^ fsw) * 16, c1 = ((2 * q + 1) ^ fsw) * 16;

  // one slab: two Box-Muller pairs (this lane's four deviates), 32 fragment reads, 64 MFMAs.  MASK: the slab crosses the strip's
  // diagonal block - entries with k > row are not part of L
  // ZMEM: this lane's Z row, and the two 16-byte pieces of the slab about to be multiplied (requested during the previous slab)
  typedef double pta_f64x2_a8 __attribute__((ext_vector_type(2), aligned(8)));
  const double *__restrict__ zrow = nullptr;
  pta_f64x2_a8 zn0 = {0.0, 0.0}, zn1 = {0.0, 0.0};
  const int zlim = (n - 1) & ~3;  // the last 4-aligned group of k that holds a deviate of this block
  auto zfetch = [&](int k0) {  // a slab may reach up to 15 columns past the factor's order: whole groups past the block are clamped back
    const int kk = min(k0 + 4 * q, zlim);  // onto its last group (alignment kept: k0 + 4 q and zlim are multiples of 4), so at most the 3
                                           // doubles that complete that group are read behind the block; their values never count (below)
    zn0 = *reinterpret_cast<const pta_f64x2_a8 *>(zrow + kk);
    zn1 = *reinterpret_cast<const pta_f64x2_a8 *>(zrow + kk + 2);
  };
  if (ZMEM) {
    zrow = pl.z + (int64_t)min(m_a, M - 1) * pl.ld_z + pl.blk_zoff[blk];
    zfetch(0);
  }
  const int sdiag = n0 / TDS_K;  // first slab that holds an element above the diagonal (n0 is a multiple of 16)
  auto slab = [&](int s, auto mask_tag) {
    constexpr bool MASK = decltype(mask_tag)::value;
    const int cur = s & 1, k0 = s * TDS_K;
    const char *pb = reinterpret_cast<const char *>(&Bs[cur][0]) + offc;
    double z[4];
    if (ZMEM) {
      z[0] = zn0.x, z[1] = zn0.y, z[2] = zn1.x, z[3] = zn1.y;
      if (MASK) {  // k >= n: not a deviate of this block (clamped duplicates, the next block's, or row padding - possibly not even finite):
#pragma unroll     // zeroed, so that 0 x NaN cannot reach an accumulator (diagonal slabs only: every k >= n lies behind the strip's diagonal)
        for (int i = 0; i < 4; ++i) z[i] = (k0 + 4 * q + i < n) ? z[i] : 0.0;
      }
      if (s + 1 < nslab) zfetch(k0 + TDS_K);
    } else {
      const uint32_t p0 = (uint32_t)((k0 >> 1) + 2 * q);
      pta_normal_pair(seed, real_a, strm_a, p0, z[0], z[1], fast);
      pta_normal_pair(seed, real_a, strm_a, p0 + 1u, z[2], z[3], fast);
    }
    const int kq = k0 + 4 * q;  // this lane's first k of the slab
    const int dlive = MASK ? __builtin_amdgcn_readfirstlane(min(max(s - sdiag, 0), TDS_NT - 1)) : 0;  // first column tile with an element on or below the diagonal
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      pta_f64x2 b[TDS_NT];
#pragma unroll
      for (int j = 0; j < TDS_NT; ++j) b[j] = *reinterpret_cast<const pta_f64x2 *>(pb + j * 16 * TDS_ROWB + (half ? c1 : c0));
      if (MASK) {
#pragma unroll
        for (int j = 0; j < TDS_NT; ++j) {
          const int row = n0 + 16 * j + c;
          b[j].x = (kq + 2 * half <= row) ? b[j].x : 0.0;
          b[j].y = (kq + 2 * half + 1 <= row) ? b[j].y : 0.0;
        }
      }
      // MASK slabs: column tile j (rows n0 + 16 j .. + 15 of the factor) lies wholly ABOVE the diagonal from slab sdiag + j + 1 on - its
      // products would multiply sixteen zeros.  They are skipped with ONE computed jump per run of products (a fall-through switch over
      // the first live tile d = s - sdiag; MFMAs ignore EXEC, so a real branch it must be): the diagonal slabs are 16 of a strip's ~164
      // and nearly half of their tile products were of this kind.
#define PTA_TD_RUN(ZV, BF)                                                                          \
      if (MASK) {                                                                                   \
        switch (dlive) {                                                                            \
          case 0: acc[0] = pta_mfma_f64(ZV, b[0].BF, acc[0]); [[fallthrough]];                      \
          case 1: acc[1] = pta_mfma_f64(ZV, b[1].BF, acc[1]); [[fallthrough]];                      \
          case 2: acc[2] = pta_mfma_f64(ZV, b[2].BF, acc[2]); [[fallthrough]];                      \
          case 3: acc[3] = pta_mfma_f64(ZV, b[3].BF, acc[3]); [[fallthrough]];                      \
          case 4: acc[4] = pta_mfma_f64(ZV, b[4].BF, acc[4]); [[fallthrough]];                      \
          case 5: acc[5] = pta_mfma_f64(ZV, b[5].BF, acc[5]); [[fallthrough]];                      \
          case 6: acc[6] = pta_mfma_f64(ZV, b[6].BF, acc[6]); [[fallthrough]];                      \
          case 7: acc[7] = pta_mfma_f64(ZV, b[7].BF, acc[7]); [[fallthrough]];                      \
          case 8: acc[8] = pta_mfma_f64(ZV, b[8].BF, acc[8]); [[fallthrough]];                      \
          case 9: acc[9] = pta_mfma_f64(ZV, b[9].BF, acc[9]); [[fallthrough]];                      \
          case 10: acc[10] = pta_mfma_f64(ZV, b[10].BF, acc[10]); [[fallthrough]];                  \
          case 11: acc[11] = pta_mfma_f64(ZV, b[11].BF, acc[11]); [[fallthrough]];                  \
          case 12: acc[12] = pta_mfma_f64(ZV, b[12].BF, acc[12]); [[fallthrough]];                  \
          case 13: acc[13] = pta_mfma_f64(ZV, b[13].BF, acc[13]); [[fallthrough]];                  \
          case 14: acc[14] = pta_mfma_f64(ZV, b[14].BF, acc[14]); [[fallthrough]];                  \
          default: acc[15] = pta_mfma_f64(ZV, b[15].BF, acc[15]);                                   \
        }                                                                                           \
      } else {                                                                                      \
        _Pragma("unroll") for (int j = 0; j < TDS_NT; ++j) acc[j] = pta_mfma_f64(ZV, b[j].BF, acc[j]); \
      }
      PTA_TD_RUN(z[2 * half], x)
      if (half == 0 && s + 1 < nslab) stage(k0 + TDS_K, cur ^ 1);  // behind the first 16 MFMAs; its buffer was last read before the previous barrier
      PTA_TD_RUN(z[2 * half + 1], y)
#undef PTA_TD_RUN
    }
    __builtin_amdgcn_sched_barrier(0);  // keep the closing barrier (and its vmcnt(0)) behind the MFMAs
    __syncthreads();
  };
  stage(0, 0);
  if (!ZMEM) pta_rng_stage_tables();  // Box-Muller tables -> LDS (pta_rng.h)
  __syncthreads();
  int s = 0;
  for (; s < min(sdiag, nslab); ++s) slab(s, std::false_type{});
  for (; s < nslab; ++s) slab(s, std::true_type{});

  // epilogue: lane holds rows (Z rows) q + 4 reg of the wave's 16 and column 16 j + c of the strip
  const bool epi = pl.gw_G != nullptr;
  const int ocol0 = pl.blk_off[blk];
#pragma unroll
  for (int j = 0; j < TDS_NT; ++j) {
    const int i = n0 + 16 * j + c;
    if (i >= n || 16 * j + c >= srows) continue;  // past the factor, or a row of the NEXT strip (computed on a cut K range: not this strip's to store)
    const int64_t oc = (int64_t)ocol0 + i;
    double add = 0.0, wgt = 0.0;
    int jl = 0;
    if (pl.det) add = pl.det[oc];
    if (epi) {
      jl = pl.gw_jlo[oc];
      wgt = pl.gw_w[oc];
    }
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int m = mg * TDS_M + wv * 16 + q + 4 * reg;
      if (m >= M) continue;
      double v = acc[j][reg];
      if (epi) {  // + GWB: linear interpolation of the mixed grid series of (realisation m, pulsar blk) (red_noise.py:286-287)
        const double *gp = pl.gw_G + ((int64_t)m * pl.n_blocks + blk) * pl.gw_npts;
        const double y0 = gp[jl];
        v = v + ((gp[jl + 1] - y0) * wgt + y0);
      }
      out[(int64_t)m * ld_out + oc] = v + add;
    }
  }
}

extern "C" int pta_td_trmm_rng(const pta_td_plan *plan_host, uint64_t seed, uint64_t r0, int M, double *out, int64_t ld_out,
                               void *stream) {
  PTA_REQUIRE(plan_host && out, PTA_E_ARG, "pta_td_trmm_rng: NULL argument");
  const pta_td_plan &p = *plan_host;
  PTA_REQUIRE(p.Lbase && p.blk_pos && p.blk_ld && p.blk_n && p.blk_off && p.item_blk && p.item_n0, PTA_E_ARG,
              "pta_td_trmm_rng: plan arrays missing");
  PTA_REQUIRE(p.n_blocks > 0 && p.n_items > 0 && p.rows_per_real > 0 && M > 0, PTA_E_ARG, "pta_td_trmm_rng: n_blocks=%d n_items=%d M=%d",
              p.n_blocks, p.n_items, M);
  PTA_REQUIRE(p.rows_per_real == 1 || p.n_blocks == 1, PTA_E_ARG, "pta_td_trmm_rng: rows_per_real > 1 needs a single factor block");
  PTA_REQUIRE(p.rows_per_real == 1 || M % p.rows_per_real == 0, PTA_E_ARG, "pta_td_trmm_rng: M=%d is not a multiple of rows_per_real=%d",
              M, p.rows_per_real);
  PTA_REQUIRE(!p.gw_G || (p.rows_per_real == 1 && p.gw_jlo && p.gw_w && p.gw_npts >= 2), PTA_E_ARG,
              "pta_td_trmm_rng: GWB epilogue needs rows_per_real == 1, gw_jlo, gw_w, gw_npts >= 2");
  PTA_REQUIRE(((uintptr_t)p.Lbase % 16) == 0, PTA_E_ARG, "pta_td_trmm_rng: Lbase must be 16-byte aligned");
  const int64_t nmg = pta_cdiv(M, TDS_M);
  const int64_t nwg = p.n_items >= TDS_MANY_ITEMS ? (int64_t)((p.n_items + 7) / 8) * 8 * nmg : (int64_t)p.n_items * ((nmg + 7) / 8) * 8;
  PTA_REQUIRE(nwg < (1LL << 31), PTA_E_ARG, "pta_td_trmm_rng: %lld workgroups exceed one launch", (long long)nwg);
  if (p.z) {
    // (rows_per_real > 1 - the shared grid factor - reads row m of z for output row m = (realisation, pulsar): ld_z = one row's deviates)
    PTA_REQUIRE(p.ld_z >= 4 && p.blk_zoff, PTA_E_ARG, "pta_td_trmm_rng: supplied deviates (plan.z) need blk_zoff and ld_z");
    // (measured and not kept: the same product as a 128 x 128-tile GEMM with BOTH operands by LDS DMA - the tile kernel of
    // pta_gemm.hip with a triangular K range, masked diagonal slabs and this epilogue, 4 x 4 fragments per wave instead of 1 x 16:
    // 28.8-29.0 ms per 1024 realisations of the 68 x 5000 array against 29.0 for this kernel - the DMA slab pipeline bounds both at
    // ~62 TFLOP/s executed - and a different summation order, i.e. no longer bit-identical to the register form)
    hipLaunchKernelGGL((k_td_trmm_rng<false, true>), dim3((unsigned)nwg), dim3(256), 0, pta_stream(stream), p, seed, r0, M, out, ld_out);
  } else if (p.rng_fast)
    hipLaunchKernelGGL((k_td_trmm_rng<true, false>), dim3((unsigned)nwg), dim3(256), 0, pta_stream(stream), p, seed, r0, M, out, ld_out);
  else
    hipLaunchKernelGGL((k_td_trmm_rng<false, false>), dim3((unsigned)nwg), dim3(256), 0, pta_stream(stream), p, seed, r0, M, out, ld_out);
  PTA_LAUNCH_CHECK();
  return PTA_OK;
}
