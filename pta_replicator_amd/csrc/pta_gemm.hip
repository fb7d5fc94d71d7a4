// fp64 GEMM on the CDNA4 matrix cores: C = alpha * A * op(B) + beta * C.
//
// The dense products of the path all come through here: the pruned inverse DFT of the GWB chain
// (W . T, red_noise.py:275-285), the ORF mix on the coarse grid (M . G0, red_noise.py:268), the
// trailing-submatrix update of the blocked Cholesky (red_noise.py:235 and TD mode) and TD mode's L . Z.
//
// v_mfma_f64_16x16x4_f64 lane layout (cdna_hip_programming.md §3, verified at run time by
// pta_selftest_mfma_f64):  A operand: lane l holds A[i = l & 15][k = l >> 4]
//                          B operand: lane l holds B[k = l >> 4][j = l & 15]
//                          C/D:       acc[reg] is C[row = (l >> 4) + 4*reg][col = l & 15]
//
// Tiling: 64x64 C tile per 256-thread workgroup (4 waves as 2x2, each 32x32 = 2x2 MFMA tiles),
// K staged through LDS in slabs of 16.  Both operand slabs are stored K-major with a row pitch of
// 80 doubles (== 16 mod 32), which makes the ds_read_b64 fragment reads bank-conflict free: the 32
// lanes of a group read k in {kk, kk+1} x 16 consecutive doubles, i.e. dword offsets
// 2*(k*80 + i) that cover all 64 banks exactly once.
#include "pta_common.h"
#include "pta_mfma.h"

#define GBM 64
#define GBN 64
#define GBK 16
#define GLD 80

// RAG: a ragged batch in end-aligned virtual coordinates (pta_common.h: pta_rag) - A / C (and B unless `bws`) are the batch's base
// pointer, the operands' origins come from (r0, c0, kv0) and the matrix's own offset / leading dimension, and rows < mbeg, columns
// < nbeg, k < kbeg (everything below the matrix's `front`) are masked: loads clamped into the valid range, products of masked k slots
// zeroed, stores predicated.
template <bool BT, bool RAG = false>
__global__ __launch_bounds__(256) void k_dgemm_mfma(int M, int N, int K, double alpha, const double *__restrict__ A,
                                                    int64_t lda, int64_t ska, const double *__restrict__ B, int64_t ldb,
                                                    double beta, double *__restrict__ C, int64_t ldc, int lower_only,
                                                    int64_t sA, int64_t sB, int64_t sC, pta_rag rg = pta_rag{nullptr, nullptr, nullptr, 0},
                                                    int r0 = 0, int c0 = 0, int kv0 = 0, int bws = 0) {
  const int bm = blockIdx.y, bn = blockIdx.x;
  if (lower_only && bn * GBN > bm * GBM + (GBM - 1)) return;  // tile entirely above the diagonal
  int mbeg = 0, nbeg = 0, kbeg = 0;
  if (RAG) {
    const int64_t o = rg.off[blockIdx.z], l = rg.ld[blockIdx.z];
    const int f = (int)rg.front[blockIdx.z];
    mbeg = max(0, f - r0), nbeg = max(0, f - c0), kbeg = max(0, f - kv0);
    if ((bm + 1) * GBM <= mbeg || (bn + 1) * GBN <= nbeg || mbeg >= M || nbeg >= N) return;
    if (kbeg >= K && beta == 1.0) return;
    B = bws ? B + (int64_t)blockIdx.z * sB : A + o + (int64_t)c0 * l + kv0;
    C = const_cast<double *>(A) + o + (int64_t)r0 * l + c0;
    A = A + o + (int64_t)r0 * l + kv0;
    lda = l, ldc = l;
    if (!bws) ldb = l;
  } else {
    A += (int64_t)blockIdx.z * sA;
    B += (int64_t)blockIdx.z * sB;
    C += (int64_t)blockIdx.z * sC;
  }
  __shared__ double As[GBK][GLD];
  __shared__ double Bs[GBK][GLD];
  const int t = threadIdx.x, l = t & 63, w = t >> 6;
  const int wm = w >> 1, wn = w & 1;
  const int m0 = bm * GBM, n0 = bn * GBN;
  pta_f64x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = pta_f64x4{0.0, 0.0, 0.0, 0.0};
  // beta != 0: this thread's 16 elements of C are requested BEFORE the K loop, from clamped - always valid - addresses (only
  // the stores are predicated): the panel-internal updates this kernel serves have K = 64, four slabs, so read one element at
  // a time after the last MFMA the C tile was most of the kernel's duration
  const int rowb = m0 + wm * 32 + (l >> 4), colb = n0 + wn * 32 + pta_mfma_col(l);
  double cv[2][2][4];
  if (beta != 0.0) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          cv[i][j][r] = C[(int64_t)min(max(rowb + i * 16 + 4 * r, mbeg), M - 1) * ldc + min(max(colb + j * 16, nbeg), N - 1)];
  }

  for (int k0 = (kbeg / GBK) * GBK; k0 < K; k0 += GBK) {
    // slab loads: unconditional, from clamped (always valid) addresses, all issued before the first select - predicated, each
    // load sits in its own exec-masked block and is waited for on its own
    double la[4], lb[4];
    {  // A slab: 64 rows x 16 k, 4 consecutive k per thread
      const int gm = min(max(m0 + (t >> 2), mbeg), M - 1), kq = (t & 3) * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) la[j] = A[(int64_t)gm * lda + (int64_t)min(max(k0 + kq + j, kbeg), K - 1) * ska];
    }
    if (BT) {  // B given as [N x K]: element (k, n) at B[n*ldb + k]
      const int gn = min(max(n0 + (t >> 2), nbeg), N - 1), kq = (t & 3) * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) lb[j] = B[(int64_t)gn * ldb + min(max(k0 + kq + j, kbeg), K - 1)];
    } else {  // B given as [K x N]: coalesced along n
      const int gk = min(k0 + (t >> 4), K - 1), nq = (t & 15) * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) lb[j] = B[(int64_t)gk * ldb + min(n0 + nq + j, N - 1)];
    }
    {
      const int row = t >> 2, kq = (t & 3) * 4;
      const bool rin = m0 + row < M;
#pragma unroll
      for (int j = 0; j < 4; ++j) As[kq + j][row] = (rin && k0 + kq + j < K && k0 + kq + j >= kbeg) ? la[j] : 0.0;
    }
    if (BT) {
      const int col = t >> 2, kq = (t & 3) * 4;
      const bool cin = n0 + col < N;
#pragma unroll
      for (int j = 0; j < 4; ++j) Bs[kq + j][col] = (cin && k0 + kq + j < K && k0 + kq + j >= kbeg) ? lb[j] : 0.0;
    } else {
      const int kr = t >> 4, nq = (t & 15) * 4;
      const bool kin = k0 + kr < K;
#pragma unroll
      for (int j = 0; j < 4; ++j) Bs[kr][nq + j] = (kin && n0 + nq + j < N) ? lb[j] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < GBK; kk += 4) {
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
        const int row = rowb + i * 16 + 4 * r, col = colb + j * 16;
        double v = alpha * acc[i][j][r];
        if (beta != 0.0) v = fma(beta, cv[i][j][r], v);
        if (row < M && col < N && row >= mbeg && col >= nbeg && (!lower_only || col <= row)) C[(int64_t)row * ldc + col] = v;
      }
}

// Large-operand variant: 128x128 C tile per workgroup (4 waves as 2x2, each 64x64 = 4x4 MFMA tiles, 128
// accumulator VGPRs), K slabs of 16 double-buffered in LDS with the next slab prefetched into registers under
// the 64 MFMAs of the current one: one barrier per slab, 8 fragment reads per 16 MFMAs.  Used by the Cholesky
// trailing update and TD mode's L.Z, where the operands are large enough to fill 128-wide tiles.
#define HBM_T 128
#define HLD 144  // == 16 (mod 32)

// VEC (needs BT, ska == 1, even K, 16-byte aligned rows): the operand slabs are fetched as double2 with 16 lanes on 16
// rows and the four quads on consecutive k pairs - 16 cache lines per wave load.  The scalar pattern (one row per lane, eight
// strided 8-byte loads) touches 64 lines per load and kept the kernel L1/TA-bound at ~60 % MFMA utilisation.
template <bool BT, bool VEC>
__global__ __launch_bounds__(256, 2) void k_dgemm_mfma128(int M, int N, int K, double alpha, const double *__restrict__ A,
                                                          int64_t lda, int64_t ska, const double *__restrict__ B,
                                                          int64_t ldb, double beta, double *__restrict__ C, int64_t ldc,
                                                          int lower_only, int64_t sA, int64_t sB, int64_t sC) {
  int bm = blockIdx.y, bn = blockIdx.x;
  if (lower_only == 2) {
    // square lower-triangular update launched over its nt (nt + 1) / 2 tiles only: tile t <-> (bm, bn), bn <= bm, row by row.
    // Besides skipping the empty workgroups this keeps the 8 XCDs (workgroups are dealt round-robin) evenly loaded - with a
    // 2-D grid whose width is a multiple of 8 every XCD owns fixed tile columns and the first ones carry 1.5x the work.
    const int tix = blockIdx.x;
    bm = (int)((sqrt(8.0 * (double)tix + 1.0) - 1.0) * 0.5);
    while ((bm + 1) * (bm + 2) / 2 <= tix) ++bm;
    while (bm * (bm + 1) / 2 > tix) --bm;
    bn = tix - bm * (bm + 1) / 2;
  } else if (lower_only && bn * HBM_T > bm * HBM_T + (HBM_T - 1)) {
    return;
  }
  A += (int64_t)blockIdx.z * sA;
  B += (int64_t)blockIdx.z * sB;
  C += (int64_t)blockIdx.z * sC;
  __shared__ double As[2][GBK][HLD];
  __shared__ double Bs[2][GBK][HLD];
  const int t = threadIdx.x, l = t & 63, w = t >> 6;
  const int wm = w >> 1, wn = w & 1;
  const int m0 = bm * HBM_T, n0 = bn * HBM_T;
  pta_f64x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = pta_f64x4{0.0, 0.0, 0.0, 0.0};
  double ra[8], rb[8];
  // A slab: 128 rows x 16 k, 8 consecutive k per thread.  B slab: [N x K] -> same pattern; [K x N] -> 16 k-rows x 8 n
  const int a_row = t >> 1, a_kq = (t & 1) * 8;
  const int b_kr = t >> 4, b_nq = (t & 15) * 8;
  const int v_row = w * 32 + (l & 15), v_kp = l >> 4;  // VEC: rows v_row + 16 g, k pairs v_kp + 4 h
  auto fetch = [&](int k0) {
    if (VEC) {
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int gm = m0 + v_row + 16 * g, gn = n0 + v_row + 16 * g, gk = k0 + 2 * (v_kp + 4 * h);
          double2 va = make_double2(0.0, 0.0), vb = make_double2(0.0, 0.0);
          if (gm < M && gk < K) va = *reinterpret_cast<const double2 *>(A + (int64_t)gm * lda + gk);
          if (gn < N && gk < K) vb = *reinterpret_cast<const double2 *>(B + (int64_t)gn * ldb + gk);
          ra[4 * g + 2 * h] = va.x;
          ra[4 * g + 2 * h + 1] = va.y;
          rb[4 * g + 2 * h] = vb.x;
          rb[4 * g + 2 * h + 1] = vb.y;
        }
      return;
    }
    int gm = m0 + a_row;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      int gk = k0 + a_kq + j;
      ra[j] = (gm < M && gk < K) ? A[(int64_t)gm * lda + (int64_t)gk * ska] : 0.0;
    }
    if (BT) {
      int gn = n0 + a_row;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        int gk = k0 + a_kq + j;
        rb[j] = (gn < N && gk < K) ? B[(int64_t)gn * ldb + gk] : 0.0;
      }
    } else {
      int gk = k0 + b_kr;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        int gn = n0 + b_nq + j;
        rb[j] = (gk < K && gn < N) ? B[(int64_t)gk * ldb + gn] : 0.0;
      }
    }
  };
  auto stash = [&](int buf) {
    if (VEC) {
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            As[buf][2 * (v_kp + 4 * h) + e][v_row + 16 * g] = ra[4 * g + 2 * h + e];
            Bs[buf][2 * (v_kp + 4 * h) + e][v_row + 16 * g] = rb[4 * g + 2 * h + e];
          }
      return;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) As[buf][a_kq + j][a_row] = ra[j];
    if (BT) {
#pragma unroll
      for (int j = 0; j < 8; ++j) Bs[buf][a_kq + j][a_row] = rb[j];
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) Bs[buf][b_kr][b_nq + j] = rb[j];
    }
  };
  fetch(0);
  stash(0);
  __syncthreads();
  double fa[4], fb[4];  // fragments of the first K-step of the slab about to be consumed
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    fa[i] = As[0][l >> 4][wm * 64 + i * 16 + (l & 15)];
    fb[i] = Bs[0][l >> 4][wn * 64 + i * 16 + (l & 15)];
  }
  const int nslab = (K + GBK - 1) / GBK;
  for (int sidx = 0; sidx < nslab; ++sidx) {
    const int cur = sidx & 1;
    const int knext = min(sidx + 1, nslab - 1) * GBK;  // the last slab harmlessly re-fetches itself
    fetch(knext);
#pragma unroll
    for (int kk = 0; kk < GBK; kk += 4) {
      double a[4], b[4];
      if (kk == 0) {  // read right behind the previous iteration's barrier (fa / fb below)
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = fa[i], b[i] = fb[i];
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = As[cur][kk + (l >> 4)][wm * 64 + i * 16 + (l & 15)];
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = Bs[cur][kk + (l >> 4)][wn * 64 + j * 16 + (l & 15)];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = pta_mfma_f64(a[i], b[j], acc[i][j]);
    }
    stash(cur ^ 1);
    __syncthreads();
    // the first fragments of the slab just published are requested HERE, in front of whatever MFMAs of this iteration the
    // compiler sinks below the barrier, instead of at the top of the next iteration behind its address arithmetic and loads
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      fa[i] = As[cur ^ 1][l >> 4][wm * 64 + i * 16 + (l & 15)];
      fb[i] = Bs[cur ^ 1][l >> 4][wn * 64 + i * 16 + (l & 15)];
    }
  }
  // epilogue: C = alpha acc + beta C.  The read of C is the kernel's one dependent HBM round trip after the last MFMA
  // (measured with beta = 0: it costs 10 % at K = 1024, 17 % at 512, 35 % at 256 when done element by element), so the 16
  // elements of one i-row of tiles are loaded back to back from clamped - always valid - addresses, waited for once, and
  // only the stores are predicated: 4 round trips per thread instead of 64.
  const int colb = n0 + wn * 64 + pta_mfma_col(l);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int rowb = m0 + wm * 64 + i * 16 + (l >> 4);
    double cv[4][4];
    if (beta != 0.0) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          cv[j][r] = C[(int64_t)min(rowb + 4 * r, M - 1) * ldc + min(colb + j * 16, N - 1)];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = rowb + 4 * r, col = colb + j * 16;
        double v = alpha * acc[i][j][r];
        if (beta != 0.0) v = fma(beta, cv[j][r], v);
        if (row < M && col < N && (!lower_only || col <= row)) C[(int64_t)row * ldc + col] = v;
      }
  }
}

// ---- direct-to-LDS variant (algo 2) ------------------------------------------------------------------------------------
// Same 128 x 128 tile, same wave layout and epilogue as k_dgemm_mfma128<true, true>, but the operand slabs never pass through
// registers: every wave issues global_load_lds_dwordx4 (16 bytes per lane straight into LDS; destination = wave-uniform base +
// lane * 16) for the NEXT slab at the top of an iteration and the data is only waited for at that iteration's closing barrier,
// 64 MFMAs later.  That removes from the K loop the 16 staging VGPRs x 2, the s_waitcnt vmcnt(0) in front of 16 ds_write_b64
// per thread, the writes themselves (ds_write_b64 sustains a third of the read rate) and, with the k-permutation below, half of
// the fragment reads:
//   * LDS image of a slab: row-major, 128 rows x 16 k = 128 bytes per row, no padding (the DMA writes lane-linearly), 16-byte
//     chunk c of row r stored at slot c ^ f(r), f(r) = (e & 1) | 6 * ((e >> 2) & 1), e = (r >> 1) & 7.  The swizzle is applied on
//     the SOURCE side (each lane fetches the chunk that belongs in the slot it writes) and again on the fragment reads, where it
//     makes every ds_read_b128 conflict-free: the instruction is serviced in the four lane groups {0-3, 12-15, 20-27}, {4-11,
//     16-19, 28-31}, +32 (MI355X_MICROARCH.md, LDS), each of which holds the 16 rows of a fragment once, eight of them reading
//     chunk c and eight chunk c ^ 2 - f() sends those two sets to complementary slots of the 256-byte bank row.
//   * the sum over k is order independent, so MFMA step t of a slab gives k-slot q = lane >> 4 the column k = 4 q + t (not
//     4 t + q): a lane's four A (or B) values of a slab are 32 contiguous bytes of its row = two ds_read_b128 instead of four
//     ds_read_b64, 16 per wave and slab in all, issued as two waves of eight so that the first 32 MFMAs start while the second
//     half is still landing.
// Where the rest goes (throw-away probe builds, M = N = 3968 lower / 4096 full, batch 34 / 16, beta = 0): the loop without its DMA runs
// at 70.4 TFLOP/s at K = 1024 and 75.2 at K = 4096 (96 % of the 78.6 peak: fragment reads and barriers cost nothing); issuing the
// eight DMA pieces per slab costs 3 %, waiting for them to land at the closing barrier another 5 % (issue -> landed is ~1 us under
// this load, a slab lasts ~2.5 us) - 64.9 / 70.9 as shipped.  Issuing the DMA at the top of the slab instead of behind the first 16
// MFMAs: no change; an XCD-contiguous tile order (each XCD walking neighbouring tiles of one matrix): no change at K = 1024, -4 % at
// K = 4096 - the operands come from the 256 MB memory-side cache either way; a deeper ring (four stages of 8 k, the DMA of stage
// s + 3 in flight across raw s_barriers with counted vmcnt(8) waits, one ds_read_b128 per fragment): 65.2 / 60.8 / 42.1 / 70.6 at
// K = 1024 / 512 / 128 / 4096 against 66.2 / 63.0 / 45.6 / 70.6 - the wait moves, the total does not: what the DMA costs is its
// share of the LDS write path and of the texture addresser, not its latency.
// Rows past M / N and k past K are fetched from clamped (valid) addresses; a K tail (K % 16 != 0) zeroes the fragments of the
// slots past K in registers, in the last iteration only.  Requirements as for VEC: transB, unit k stride, even K and leading
// dimensions, 16-byte aligned operands.
#define GL_ROWB 128                       // bytes per LDS row (16 doubles)
#define GL_OPB (HBM_T * GL_ROWB)          // bytes per operand slab: 16 KB
__device__ __forceinline__ int pta_gl_f(int row) {
  const int e = (row >> 1) & 7;
  return (e & 1) | (((e >> 2) & 1) * 6);
}

// RAG (pta_common.h: pta_rag): a ragged batch in end-aligned virtual coordinates - operand origins from (r0, c0, kv0) and the matrix's own
// offset / leading dimension; rows < mbeg, columns < nbeg and k < kbeg (= below the matrix's `front`) are masked: DMA sources clamped into
// the valid range, the fragments of masked k slots zeroed in the one slab that straddles kbeg (the K loop starts at that slab), stores
// predicated; tiles wholly below the front leave at once.  The uniform instantiation carries none of it.
// EPI = 1 (A/B, PTA_POTRF_EPI1 / pta_dgemm algo 3): the C tile's first 16 elements per lane are requested BEFORE the last slab's products and
// every later batch before the previous batch's stores (two register sets in rotation), and a tile that no bound, mask or diagonal cuts
// stores unpredicated - the epilogue's four dependent load round trips become one that the last 64 MFMAs cover.
template <bool RAG, int EPI = 0>
__global__ __launch_bounds__(256, 2) void k_dgemm_glds128(int M, int N, int K, double alpha, const double *__restrict__ A, int64_t lda,
                                                          const double *__restrict__ B, int64_t ldb, double beta, double *__restrict__ C,
                                                          int64_t ldc, int lower_only, int64_t sA, int64_t sB, int64_t sC, pta_rag rg, int r0,
                                                          int c0v, int kv0, int bws) {
  int bm = blockIdx.y, bn = blockIdx.x;
  if (lower_only == 2) {
    // 1-D grid over the tiles on or below the diagonal of a TRAPEZOID (M >= N): the nt (nt + 1) / 2 tiles of the top square row by row,
    // then the (mt - nt) x nt rectangle below it row by row.  Workgroups are dealt to the 8 XCDs round-robin, so every XCD gets the same
    // number of tiles (a 2-D grid with early exits above the diagonal gives XCD x = column tile x its 31 - x tiles of a 31 x 8 block
    // column: the first XCD carries 29 % more than the last - round 6, profiles/r06_potrf_left_looking.txt), and in the rectangle an XCD
    // keeps ONE column tile (nt = 8), whose operand stays in its L2.
    const int tix = blockIdx.x;
    const int nt = (N + HBM_T - 1) / HBM_T, tri = nt * (nt + 1) / 2;
    if (tix < tri) {
      bm = (int)((sqrt(8.0 * (double)tix + 1.0) - 1.0) * 0.5);
      while ((bm + 1) * (bm + 2) / 2 <= tix) ++bm;
      while (bm * (bm + 1) / 2 > tix) --bm;
      bn = tix - bm * (bm + 1) / 2;
    } else {
      bm = nt + (tix - tri) / nt;
      bn = (tix - tri) % nt;
    }
  } else if (lower_only && bn * HBM_T > bm * HBM_T + (HBM_T - 1)) {
    return;
  }
  int mbeg = 0, nbeg = 0, kbeg = 0;
  if (RAG) {
    const int64_t o = rg.off[blockIdx.z], l = rg.ld[blockIdx.z];
    const int f = (int)rg.front[blockIdx.z];
    mbeg = max(0, f - r0), nbeg = max(0, f - c0v), kbeg = max(0, f - kv0);
    if ((bm + 1) * HBM_T <= mbeg || (bn + 1) * HBM_T <= nbeg || mbeg >= M || nbeg >= N) return;
    if (kbeg >= K && beta == 1.0) return;
    B = bws ? B + (int64_t)blockIdx.z * sB : A + o + (int64_t)c0v * l + kv0;
    C = const_cast<double *>(A) + o + (int64_t)r0 * l + c0v;
    A = A + o + (int64_t)r0 * l + kv0;
    lda = l, ldc = l;
    if (!bws) ldb = l;
  } else {
    A += (int64_t)blockIdx.z * sA;
    B += (int64_t)blockIdx.z * sB;
    C += (int64_t)blockIdx.z * sC;
  }
  __shared__ double __attribute__((aligned(256))) slab[2][2][HBM_T * GBK];  // [stage][operand][row * 16 + k], swizzled per row
  const int t = threadIdx.x, l = t & 63, w = t >> 6;
  const int wm = w >> 1, wn = w & 1;
  const int m0 = bm * HBM_T, n0 = bn * HBM_T;
  typedef double pta_f64x2 __attribute__((ext_vector_type(2)));
  // ---- DMA side: wave w stages rows [32 w, 32 w + 32) of both operands, 8 rows per instruction
  const double *__restrict__ srcA[4];
  const double *__restrict__ srcB[4];
  int kc[4];  // first k (inside a slab) of the chunk this lane fetches in instruction j
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = 32 * w + 8 * j + (l >> 3);
    kc[j] = 2 * ((l & 7) ^ pta_gl_f(row));
    srcA[j] = A + (int64_t)min(RAG ? max(m0 + row, mbeg) : m0 + row, M - 1) * lda;
    srcB[j] = B + (int64_t)min(RAG ? max(n0 + row, nbeg) : n0 + row, N - 1) * ldb;
  }
  auto stage = [&](int k0, int st) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = min(RAG ? max(k0 + kc[j], kbeg) : k0 + kc[j], K - 2);
      char *dA = reinterpret_cast<char *>(&slab[st][0][0]) + (32 * w + 8 * j) * GL_ROWB;
      char *dB = reinterpret_cast<char *>(&slab[st][1][0]) + (32 * w + 8 * j) * GL_ROWB;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(srcA[j] + k),
                                       (__attribute__((address_space(3))) void *)dA, 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(srcB[j] + k),
                                       (__attribute__((address_space(3))) void *)dB, 16, 0, 0);
    }
  };
  // ---- fragment side: lane (i = l & 15, q = l >> 4) reads chunks 2 q and 2 q + 1 of rows 16 blk + i
  const int fi = l & 15, fq = l >> 4;
  const int fsw = pta_gl_f(fi);  // rows 16 blk + i share (row >> 1) & 7 with i
  const int offA = (wm * 64 + fi) * GL_ROWB, offB = (wn * 64 + fi) * GL_ROWB;
  const int c0 = ((2 * fq) ^ fsw) * 16, c1 = ((2 * fq + 1) ^ fsw) * 16;
  pta_f64x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = pta_f64x4{0.0, 0.0, 0.0, 0.0};
  // one slab: 16 fragment reads (the eight that feed MFMA steps 0 / 1 first), then 64 MFMAs.  `kv` < 16 only for a K tail, which runs
  // as a peeled last iteration so that the steady-state loop carries no selects and its waits stay progressive.
  auto slab_product = [&](int cur, int kv, int next_k0, int kh = 0) {
    const char *pa = reinterpret_cast<const char *>(&slab[cur][0][0]) + offA;
    const char *pb = reinterpret_cast<const char *>(&slab[cur][1][0]) + offB;
    pta_f64x2 a0[4], b0[4], a1[4], b1[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      a0[i] = *reinterpret_cast<const pta_f64x2 *>(pa + i * 16 * GL_ROWB + c0);
      b0[i] = *reinterpret_cast<const pta_f64x2 *>(pb + i * 16 * GL_ROWB + c0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      a1[i] = *reinterpret_cast<const pta_f64x2 *>(pa + i * 16 * GL_ROWB + c1);
      b1[i] = *reinterpret_cast<const pta_f64x2 *>(pb + i * 16 * GL_ROWB + c1);
    }
    if (kv < GBK) {  // K tail: slots past K contribute nothing (their loads were clamped duplicates, possibly NaN scratch)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (4 * fq + 0 >= kv) a0[i].x = 0.0, b0[i].x = 0.0;
        if (4 * fq + 1 >= kv) a0[i].y = 0.0, b0[i].y = 0.0;
        if (4 * fq + 2 >= kv) a1[i].x = 0.0, b1[i].x = 0.0;
        if (4 * fq + 3 >= kv) a1[i].y = 0.0, b1[i].y = 0.0;
      }
    }
    if (RAG && kh > 0) {  // K head of a ragged matrix: slots below its front (clamped duplicates / unwritten workspace)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (4 * fq + 0 < kh) a0[i].x = 0.0, b0[i].x = 0.0;
        if (4 * fq + 1 < kh) a0[i].y = 0.0, b0[i].y = 0.0;
        if (4 * fq + 2 < kh) a1[i].x = 0.0, b1[i].x = 0.0;
        if (4 * fq + 3 < kh) a1[i].y = 0.0, b1[i].y = 0.0;
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = pta_mfma_f64(a0[i].x, b0[j].x, acc[i][j]);
    // the next slab's DMA is issued HERE, behind the fragment reads and the first 16 MFMAs: a DMA piece costs ~60 issue cycles
    // (MI355X_MICROARCH.md), which then pass while the matrix pipe works through the queue instead of in front of the reads; its
    // buffer (cur ^ 1) was last read before the previous barrier
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
    // the closing barrier carries s_waitcnt vmcnt(0) for the DMA issued at the top of the iteration: it must stay BEHIND the 64
    // MFMAs (left alone, the scheduler hoists it to right after the first one - the product only touches registers - and the DMA
    // latency is then waited for, every slab, instead of hiding under 4096 matrix-pipe cycles)
    __builtin_amdgcn_sched_barrier(0);
  };
  const int nfull = K / GBK, ktail = K - nfull * GBK, nslab = nfull + (ktail ? 1 : 0);
  const int s0 = RAG ? kbeg / GBK : 0;  // first slab with a valid k
  const int colb = n0 + wn * 64 + pta_mfma_col(l);
  // C element (batch i, column tile j, register r) of this lane, from a clamped - always valid - address
  auto c_load = [&](int i, double (&cv)[4][4]) {
    const int rowb = m0 + wm * 64 + i * 16 + (l >> 4);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        cv[j][r] = C[(int64_t)min(RAG ? max(rowb + 4 * r, mbeg) : rowb + 4 * r, M - 1) * ldc + min(RAG ? max(colb + j * 16, nbeg) : colb + j * 16, N - 1)];
  };
  double cva[4][4], cvb[4][4];
  const bool has_c = beta != 0.0;
  if (!RAG || s0 < nslab) {
    stage(s0 * GBK, s0 & 1);
    __syncthreads();  // drains the DMA (vmcnt(0)) before any wave reads the slab
    if (EPI) {
      const int last = nslab - 1;  // the last slab (a K tail, or the last full one) runs behind the C prefetch
      for (int sidx = s0; sidx < last; ++sidx) {
        slab_product(sidx & 1, GBK, (sidx + 1) * GBK, (RAG && sidx == s0) ? kbeg - s0 * GBK : 0);
        __syncthreads();
      }
      if (has_c) c_load(0, cva);
      slab_product(last & 1, ktail ? ktail : GBK, -1, (RAG && last == s0) ? kbeg - s0 * GBK : 0);
    } else {
      for (int sidx = s0; sidx < nfull; ++sidx) {
        slab_product(sidx & 1, GBK, sidx + 1 < nslab ? (sidx + 1) * GBK : -1, (RAG && sidx == s0) ? kbeg - s0 * GBK : 0);
        __syncthreads();  // every wave is done reading this slab; the DMA of the next one has landed
      }
      if (ktail) slab_product(nfull & 1, ktail, -1, (RAG && nfull == s0) ? kbeg - s0 * GBK : 0);
    }
  } else if (EPI && has_c) {
    c_load(0, cva);
  }
  if (EPI) {
    // workgroup-uniform: no row / column bound, no ragged mask and no diagonal cuts this tile
    const bool interior = m0 + HBM_T <= M && n0 + HBM_T <= N && (!RAG || (mbeg <= m0 && nbeg <= n0)) && (!lower_only || n0 + HBM_T - 1 <= m0);
    auto c_store = [&](int i, const double (&cv)[4][4]) {
      const int rowb = m0 + wm * 64 + i * 16 + (l >> 4);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = rowb + 4 * r, col = colb + j * 16;
          double v = alpha * acc[i][j][r];
          if (has_c) v = fma(beta, cv[j][r], v);
          if (interior)
            C[(int64_t)row * ldc + col] = v;
          else if (row < M && col < N && (!RAG || (row >= mbeg && col >= nbeg)) && (!lower_only || col <= row))
            C[(int64_t)row * ldc + col] = v;
        }
    };
    if (has_c) c_load(1, cvb);
    c_store(0, cva);
    if (has_c) c_load(2, cva);
    c_store(1, cvb);
    if (has_c) c_load(3, cvb);
    c_store(2, cva);
    c_store(3, cvb);
    return;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int rowb = m0 + wm * 64 + i * 16 + (l >> 4);
    double cv[4][4];
    if (beta != 0.0) c_load(i, cv);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = rowb + 4 * r, col = colb + j * 16;
        double v = alpha * acc[i][j][r];
        if (beta != 0.0) v = fma(beta, cv[j][r], v);
        if (row < M && col < N && (!RAG || (row >= mbeg && col >= nbeg)) && (!lower_only || col <= row)) C[(int64_t)row * ldc + col] = v;
      }
  }
}

// plain VALU kernel: one thread per C element.  Cross-check for the MFMA kernel and the fallback
// for operand shapes too small to fill a tile.
template <bool BT>
__global__ void k_dgemm_valu(int M, int N, int K, double alpha, const double *__restrict__ A, int64_t lda, int64_t ska,
                             const double *__restrict__ B, int64_t ldb, double beta, double *__restrict__ C, int64_t ldc,
                             int lower_only, int64_t sA, int64_t sB, int64_t sC) {
  int col = blockIdx.x * blockDim.x + threadIdx.x;
  int row = blockIdx.y;
  if (col >= N || row >= M || (lower_only && col > row)) return;
  A += (int64_t)blockIdx.z * sA;
  B += (int64_t)blockIdx.z * sB;
  C += (int64_t)blockIdx.z * sC;
  double acc = 0.0;
  for (int k = 0; k < K; ++k) {
    double b = BT ? B[(int64_t)col * ldb + k] : B[(int64_t)k * ldb + col];
    acc = fma(A[(int64_t)row * lda + (int64_t)k * ska], b, acc);
  }
  int64_t o = (int64_t)row * ldc + col;
  double v = alpha * acc;
  C[o] = (beta == 0.0) ? v : v + beta * C[o];
}

// width of the column tiles pta_dgemm_launch will use for this shape (callers that update an operand IN PLACE must keep each
// launch to ONE column tile: pta_potrf_step_ws)
int pta_dgemm_tile_n(int M, int N, int K, int algo) { return (algo >= 1 && M >= 256 && N >= 128 && K >= 32) ? HBM_T : GBN; }

int pta_dgemm_launch(int transB, int M, int N, int K, double alpha, const double *A, int64_t lda, int64_t ska,
                     const double *B, int64_t ldb, double beta, double *C, int64_t ldc, int lower_only, int batch,
                     int64_t sA, int64_t sB, int64_t sC, int algo, hipStream_t stream) {
  PTA_REQUIRE(A && B && C, PTA_E_ARG, "pta_dgemm: NULL argument");
  PTA_REQUIRE(M > 0 && N > 0 && K > 0 && batch > 0 && batch <= 65535, PTA_E_ARG, "pta_dgemm: M=%d N=%d K=%d batch=%d", M, N, K, batch);
  if (algo == 0) {
    PTA_REQUIRE(M <= 65535, PTA_E_ARG, "pta_dgemm: VALU kernel supports M <= 65535 (got %d)", M);
    dim3 g(pta_cdiv(N, 128), M, batch);
    if (transB)
      hipLaunchKernelGGL(k_dgemm_valu<true>, g, dim3(128), 0, stream, M, N, K, alpha, A, lda, ska, B, ldb, beta, C, ldc, lower_only, sA, sB, sC);
    else
      hipLaunchKernelGGL(k_dgemm_valu<false>, g, dim3(128), 0, stream, M, N, K, alpha, A, lda, ska, B, ldb, beta, C, ldc, lower_only, sA, sB, sC);
  } else if (algo >= 1 && M >= 256 && N >= 128 && K >= 32) {  // large operands: 128x128 tiles (algo 2: operand slabs by LDS DMA)
    PTA_REQUIRE(pta_cdiv(M, HBM_T) <= 65535u, PTA_E_ARG, "pta_dgemm: M=%d too large", M);
    dim3 g(pta_cdiv(N, HBM_T), pta_cdiv(M, HBM_T), batch);
    if (lower_only && M == N) {
      const unsigned nt = pta_cdiv(M, HBM_T);
      g = dim3(nt * (nt + 1) / 2, 1, batch);
      lower_only = 2;
    }
    const bool vec = transB && ska == 1 && (K % 2) == 0 && (lda % 2) == 0 && (ldb % 2) == 0 && (sA % 2) == 0 && (sB % 2) == 0 &&
                     ((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0;
    if (lower_only == 1 && M > N && vec && algo >= 2) {  // trapezoid (a block column with its diagonal block on top): 1-D grid over its tiles
      const unsigned nt = pta_cdiv(N, HBM_T), mt = pta_cdiv(M, HBM_T);
      g = dim3(nt * (nt + 1) / 2 + (mt - nt) * nt, 1, batch);
      lower_only = 2;
    }
    if (vec && algo >= 2) {
      auto kern = algo == 3 ? k_dgemm_glds128<false, 1> : k_dgemm_glds128<false, 0>;
      hipLaunchKernelGGL(kern, g, dim3(256), 0, stream, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc,
                         lower_only, sA, sB, sC, pta_rag{nullptr, nullptr, nullptr, 0}, 0, 0, 0, 0);
    } else if (vec)
      hipLaunchKernelGGL((k_dgemm_mfma128<true, true>), g, dim3(256), 0, stream, M, N, K, alpha, A, lda, ska, B, ldb, beta, C, ldc, lower_only, sA, sB, sC);
    else if (transB)
      hipLaunchKernelGGL((k_dgemm_mfma128<true, false>), g, dim3(256), 0, stream, M, N, K, alpha, A, lda, ska, B, ldb, beta, C, ldc, lower_only, sA, sB, sC);
    else
      hipLaunchKernelGGL((k_dgemm_mfma128<false, false>), g, dim3(256), 0, stream, M, N, K, alpha, A, lda, ska, B, ldb, beta, C, ldc, lower_only, sA, sB, sC);
  } else {
    PTA_REQUIRE(pta_cdiv(M, GBM) <= 65535u, PTA_E_ARG, "pta_dgemm: M=%d too large", M);
    dim3 g(pta_cdiv(N, GBN), pta_cdiv(M, GBM), batch);
    if (transB)
      hipLaunchKernelGGL(k_dgemm_mfma<true>, g, dim3(256), 0, stream, M, N, K, alpha, A, lda, ska, B, ldb, beta, C, ldc, lower_only, sA, sB, sC);
    else
      hipLaunchKernelGGL(k_dgemm_mfma<false>, g, dim3(256), 0, stream, M, N, K, alpha, A, lda, ska, B, ldb, beta, C, ldc, lower_only, sA, sB, sC);
  }
  PTA_LAUNCH_CHECK();
  return PTA_OK;
}

int pta_dgemm_launch_rag(int M, int N, int K, double alpha, double *Abase, int r0, int c0, int k0, const double *Bws, int64_t ldb, int64_t sB,
                         double beta, int lower_only, int batch, pta_rag rg, hipStream_t stream) {
  PTA_REQUIRE(Abase && rg.off && rg.ld && rg.front, PTA_E_ARG, "pta_dgemm_rag: NULL argument");
  PTA_REQUIRE(M > 0 && N > 0 && K > 0 && batch > 0 && batch <= 65535, PTA_E_ARG, "pta_dgemm_rag: M=%d N=%d K=%d batch=%d", M, N, K, batch);
  PTA_REQUIRE(!(r0 & 1) && !(c0 & 1) && !(k0 & 1) && !(K & 1) && !(ldb & 1) && !(sB & 1) && ((uintptr_t)Abase % 16) == 0 && ((uintptr_t)Bws % 16) == 0,
              PTA_E_ARG, "pta_dgemm_rag: origins, K and the workspace operand must be even / 16-byte aligned");
  const double *B = Bws ? Bws : Abase;
  const int bws = Bws ? 1 : 0;
  if (M >= 256 && N >= 128 && K >= 32) {
    PTA_REQUIRE(pta_cdiv(M, HBM_T) <= 65535u, PTA_E_ARG, "pta_dgemm_rag: M=%d too large", M);
    dim3 g(pta_cdiv(N, HBM_T), pta_cdiv(M, HBM_T), batch);
    if (lower_only && M == N) {
      const unsigned nt = pta_cdiv(M, HBM_T);
      g = dim3(nt * (nt + 1) / 2, 1, batch);
      lower_only = 2;
    } else if (lower_only && M > N) {  // trapezoid (a block column): 1-D grid over its live tiles, dealt evenly to the XCDs (see the kernel)
      const unsigned nt = pta_cdiv(N, HBM_T), mt = pta_cdiv(M, HBM_T);
      g = dim3(nt * (nt + 1) / 2 + (mt - nt) * nt, 1, batch);
      lower_only = 2;
    }
    auto kern = rg.epi ? k_dgemm_glds128<true, 1> : k_dgemm_glds128<true, 0>;
    hipLaunchKernelGGL(kern, g, dim3(256), 0, stream, M, N, K, alpha, Abase, (int64_t)0, B, ldb, beta, Abase, (int64_t)0, lower_only,
                       (int64_t)0, sB, (int64_t)0, rg, r0, c0, k0, bws);
  } else {
    PTA_REQUIRE(pta_cdiv(M, GBM) <= 65535u, PTA_E_ARG, "pta_dgemm_rag: M=%d too large", M);
    dim3 g(pta_cdiv(N, GBN), pta_cdiv(M, GBM), batch);
    hipLaunchKernelGGL((k_dgemm_mfma<true, true>), g, dim3(256), 0, stream, M, N, K, alpha, Abase, (int64_t)0, (int64_t)1, B, ldb, beta, Abase, (int64_t)0,
                       lower_only, (int64_t)0, sB, (int64_t)0, rg, r0, c0, k0, bws);
  }
  PTA_LAUNCH_CHECK();
  return PTA_OK;
}

extern "C" int pta_dgemm(int transB, int M, int N, int K, double alpha, const double *A, int64_t lda, int64_t ska,
                         const double *B, int64_t ldb, double beta, double *C, int64_t ldc, int lower_only, int batch,
                         int64_t strideA, int64_t strideB, int64_t strideC, int algo, void *stream) {
  return pta_dgemm_launch(transB, M, N, K, alpha, A, lda, ska, B, ldb, beta, C, ldc, lower_only, batch, strideA, strideB,
                          strideC, algo, pta_stream(stream));
}

// ---- run-time check of the lane layout assumed above ------------------------------------------
__global__ void k_selftest_mfma(const double *__restrict__ A, const double *__restrict__ B, double *__restrict__ C) {
  int l = threadIdx.x;
  pta_f64x4 acc = {0.0, 0.0, 0.0, 0.0};
  acc = pta_mfma_f64(A[(l & 15) * 4 + (l >> 4)], B[(l >> 4) * 16 + (l & 15)], acc);
  for (int r = 0; r < 4; ++r) C[pta_mfma_row(l, r) * 16 + pta_mfma_col(l)] = acc[r];
}

extern "C" int pta_selftest_mfma_f64(double *max_err_host) {
  PTA_REQUIRE(max_err_host, PTA_E_ARG, "pta_selftest_mfma_f64: NULL argument");
  double hA[64], hB[64], hC[256], *dA, *dB, *dC;
  for (int i = 0; i < 16; ++i)
    for (int k = 0; k < 4; ++k) hA[i * 4 + k] = 1.0 + 0.37 * i - 0.11 * k * k + 0.013 * i * k;  // asymmetric on purpose
  for (int k = 0; k < 4; ++k)
    for (int j = 0; j < 16; ++j) hB[k * 16 + j] = -0.5 + 0.21 * j + 0.77 * k - 0.031 * j * k * k;
  PTA_HIP(hipMalloc(&dA, sizeof(hA)));
  PTA_HIP(hipMalloc(&dB, sizeof(hB)));
  PTA_HIP(hipMalloc(&dC, sizeof(hC)));
  PTA_HIP(hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice));
  PTA_HIP(hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_selftest_mfma, dim3(1), dim3(64), 0, 0, dA, dB, dC);
  PTA_LAUNCH_CHECK();
  PTA_HIP(hipMemcpy(hC, dC, sizeof(hC), hipMemcpyDeviceToHost));
  (void)hipFree(dA);
  (void)hipFree(dB);
  (void)hipFree(dC);
  double worst = 0.0;
  for (int i = 0; i < 16; ++i)
    for (int j = 0; j < 16; ++j) {
      double ref = 0.0;
      for (int k = 0; k < 4; ++k) ref = fma(hA[i * 4 + k], hB[k * 16 + j], ref);
      double e = fabs(hC[i * 16 + j] - ref);
      if (e > worst) worst = e;
    }
  *max_err_host = worst;
  if (worst > 1e-12) {
    pta_set_error("fp64 MFMA lane layout self-test failed: max |err| = %g", worst);
    return PTA_E_HIP;
  }
  return PTA_OK;
}
