// GWB chain on the device (add_gwb, red_noise.py:238-287).
//
//   reference:  w -> M @ w -> * sqrt(C), zero DC & Nyquist -> Hermitian pack (n = 2Nf-2) -> real(ifft)/dt
//               -> crop samples [10, 10+npts) -> interp1d onto each pulsar's TOAs
//   device:     G0 = W . T          pruned inverse DFT of the raw draws, a dense fp64 GEMM on MFMA
//               G  = M . G0         ORF mix, now on npts = 600 samples instead of Nf = 3000 bins
//               out = lerp(G)       numpy.interp's two-point formula
// Every step is linear, so the re-association changes results only at the 1e-15 level; what it buys:
// no length-5998 (= 2 x 2999, prime) FFT, 10x fewer flops in the mix, and the only large operand
// (T, 28.8 MB at the headline size) is realisation independent and stays L2/Infinity-Cache resident.
#include "pta_common.h"
#include "pta_mfma.h"
#include "pta_rng.h"

int pta_get_gemm_algo();

// T0[k-1][jj] =  amp_k cos(2 pi j k / n),  T1[k-1][jj] = -amp_k sin(2 pi j k / n),  j = i0 + jj,
// amp_k = 2 sqrtC[k] / (n dt): the factor 2 is the Hermitian partner bin, 1/n numpy's ifft norm,
// 1/dt red_noise.py:279.  j*k is reduced mod n in integers, so the angle is exact to one rounding.
__global__ void k_gwb_twiddle(const double *__restrict__ sqrtC, int Nf, int npts, int i0, double inv_dt,
                              double *__restrict__ T, int64_t ldt) {
  int jj = blockIdx.x * blockDim.x + threadIdx.x;
  int kq = blockIdx.y;  // k - 1
  if (jj >= ldt) return;
  const int Kf = Nf - 2;
  double c = 0.0, s = 0.0;
  if (jj < npts) {
    const int64_t n = 2 * (int64_t)Nf - 2;
    int64_t m = ((int64_t)(i0 + jj) * (int64_t)(kq + 1)) % n;
    double sn, cs;
    sincospi((double)(2 * m) / (double)n, &sn, &cs);
    double amp = 2.0 * inv_dt / (double)n * sqrtC[kq + 1];
    c = amp * cs;
    s = -(amp * sn);
  }
  T[(int64_t)kq * ldt + jj] = c;
  T[((int64_t)Kf + kq) * ldt + jj] = s;
}

extern "C" int pta_gwb_twiddle(const double *sqrtC, int Nf, int npts, int i0, double inv_dt, double *T, int64_t ldt,
                               void *stream) {
  PTA_REQUIRE(sqrtC && T, PTA_E_ARG, "pta_gwb_twiddle: NULL argument");
  PTA_REQUIRE(Nf >= 3 && Nf - 2 <= 65535 && npts > 0 && ldt >= npts && i0 >= 0, PTA_E_ARG, "pta_gwb_twiddle: Nf=%d npts=%d ldt=%lld", Nf,
              npts, (long long)ldt);
  hipLaunchKernelGGL(k_gwb_twiddle, dim3(pta_cdiv(ldt, 128), Nf - 2), dim3(128), 0, pta_stream(stream), sqrtC, Nf, npts, i0,
                     inv_dt, T, ldt);
  PTA_LAUNCH_CHECK();
  return PTA_OK;
}

// replay mode: the draws come from a buffer (NumPy's legacy stream, consumed on the host in the
// reference's order).  Two GEMMs over the real and imaginary planes of the interleaved rows.
extern "C" int pta_gwb_idft(const double *w, int64_t ldw, int M, int Nf, const double *T, int64_t ldt, int npts, double *G0,
                            int64_t ldg, int algo, void *stream) {
  PTA_REQUIRE(w && T && G0, PTA_E_ARG, "pta_gwb_idft: NULL argument");
  PTA_REQUIRE(M > 0 && Nf >= 3 && npts > 0 && ldw >= 2 * (int64_t)Nf && ldt >= npts && ldg >= npts, PTA_E_ARG,
              "pta_gwb_idft: M=%d Nf=%d npts=%d", M, Nf, npts);
  const int Kf = Nf - 2;
  hipStream_t s = pta_stream(stream);
  for (int m0 = 0; m0 < M; m0 += 32768) {  // VALU kernel's grid.y limit
    int mc = (M - m0 < 32768) ? (M - m0) : 32768;
    const double *wr = w + (int64_t)m0 * ldw;
    double *g = G0 + (int64_t)m0 * ldg;
    int rc = pta_dgemm_launch(0, mc, npts, Kf, 1.0, wr + 2, ldw, 2, T, ldt, 0.0, g, ldg, 0, 1, 0, 0, 0, algo, s);
    if (rc != PTA_OK) return rc;
    rc = pta_dgemm_launch(0, mc, npts, Kf, 1.0, wr + 3, ldw, 2, T + (int64_t)Kf * ldt, ldt, 1.0, g, ldg, 0, 1, 0, 0, 0, algo, s);
    if (rc != PTA_OK) return rc;
  }
  return PTA_OK;
}

// ---- throughput mode: the draws never exist in memory -------------------------------------------
// Workgroup = 4 waves = 64 rows (row m = realisation r * P + pulsar a) x all output samples.  Wave w
// owns rows [16w, 16w+16) and all NT column tiles: NT x 4 fp64 accumulators per lane (304 VGPRs at
// NT = 38, one wave per SIMD).  Per K-step of 4 frequency bins a lane is A-operand element
// (row l & 15, bin l >> 4) of v_mfma_f64_16x16x4_f64 - i.e. exactly ONE complex draw w[a, k], which
// it generates in registers (Philox + Box-Muller) and feeds to two MFMAs per column tile: Re against
// the cos plane, Im against the sin plane.  The T slab of the K-step (4 rows x 2 planes) is shared by
// the 4 waves through LDS, double buffered; T itself streams from L2 (all workgroups of an XCD walk it
// in lock-step, so HBM sees it once).
#define IR_NT 38             // column tiles of 16 -> up to 608 output samples per pass
#define IR_LD (IR_NT * 16)   // columns of a slab row
#define IR_PITCH (IR_LD + 16) // LDS row pitch == 16 (mod 32) doubles: the 2 k-rows a 32-lane group reads hit disjoint banks
#define IR_SLAB (2 * 4 * IR_LD)    // elements of one K-step slab: [plane][k][col]
#define IR_BUF (2 * 4 * IR_PITCH)  // doubles of one LDS buffer

__global__ __launch_bounds__(256, 1) void k_gwb_idft_rng(uint64_t seed, uint64_t r0, int M, int P, int Nf,
                                                         const double *__restrict__ T, int64_t ldt, int npts, int col0,
                                                         double *__restrict__ G0, int64_t ldg) {
  extern __shared__ __attribute__((aligned(16))) double lds[];  // 2 buffers x [2 planes][4 k][IR_PITCH]
  const int t = threadIdx.x, l = t & 63, wv = t >> 6;
  const int Kf = Nf - 2;
  const int ncol = min(npts - col0, IR_LD);
  const int ntile = (ncol + 15) >> 4;
  const int m = blockIdx.x * 64 + wv * 16 + (l & 15);
  const int mr = min(m, M - 1);
  const uint64_t real = r0 + (uint64_t)(mr / P);
  const uint32_t strm = pta_stream_id(PTA_STREAM_GWB, (uint32_t)(mr % P));
  pta_f64x4 acc[IR_NT];
#pragma unroll
  for (int i = 0; i < IR_NT; ++i) acc[i] = pta_f64x4{0.0, 0.0, 0.0, 0.0};

  const int nstep = (Kf + 3) >> 2;
  // slab loader: element e of the slab = (plane, k, col); 19 elements per thread
  auto load_slab = [&](int ks, double *regs) {
#pragma unroll
    for (int q = 0; q < IR_SLAB / 256; ++q) {
      int e = q * 256 + t;
      int col = e % IR_LD, pk = e / IR_LD;  // pk = plane*4 + k
      int kq = ks * 4 + (pk & 3);
      bool ok = (kq < Kf) && (col < ncol);
      regs[q] = ok ? T[((int64_t)(pk >> 2) * Kf + kq) * ldt + col0 + col] : 0.0;
    }
  };
  auto store_slab = [&](double *buf, const double *regs) {
#pragma unroll
    for (int q = 0; q < IR_SLAB / 256; ++q) {
      int e = q * 256 + t;
      buf[(e / IR_LD) * IR_PITCH + (e % IR_LD)] = regs[q];
    }
  };
  double regs[IR_SLAB / 256];
  load_slab(0, regs);
  store_slab(lds, regs);
  __syncthreads();
  for (int ks = 0; ks < nstep; ++ks) {
    double *cur = lds + (ks & 1) * IR_BUF;
    double *nxt = lds + ((ks + 1) & 1) * IR_BUF;
    if (ks + 1 < nstep) load_slab(ks + 1, regs);  // global loads in flight under the MFMAs below
    double re, im;
    pta_normal_pair(seed, real, strm, (uint32_t)(ks * 4 + (l >> 4) + 1), re, im);  // pair k <-> w[a,k]
    const double *bc = cur + (l >> 4) * IR_PITCH + (l & 15);
    const double *bs = bc + 4 * IR_PITCH;
    // two sweeps over the tiles so that consecutive MFMAs never share an accumulator
#pragma unroll
    for (int i = 0; i < IR_NT; ++i)
      if (i < ntile) acc[i] = pta_mfma_f64(re, bc[i * 16], acc[i]);
#pragma unroll
    for (int i = 0; i < IR_NT; ++i)
      if (i < ntile) acc[i] = pta_mfma_f64(im, bs[i * 16], acc[i]);
    if (ks + 1 < nstep) store_slab(nxt, regs);
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < IR_NT; ++i) {
    if (i < ntile) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int row = blockIdx.x * 64 + wv * 16 + pta_mfma_row(l, r);
        int col = i * 16 + pta_mfma_col(l);
        if (row < M && col < ncol) G0[(int64_t)row * ldg + col0 + col] = acc[i][r];
      }
    }
  }
}

extern "C" int pta_gwb_idft_rng(uint64_t seed, uint64_t r0, int R, int P, int Nf, const double *T, int64_t ldt, int npts,
                                double *G0, int64_t ldg, void *stream) {
  PTA_REQUIRE(T && G0, PTA_E_ARG, "pta_gwb_idft_rng: NULL argument");
  PTA_REQUIRE(R > 0 && P > 0 && P < (1 << 24) && Nf >= 3 && npts > 0 && ldt >= npts && ldg >= npts, PTA_E_ARG,
              "pta_gwb_idft_rng: R=%d P=%d Nf=%d npts=%d", R, P, Nf, npts);
  int64_t M64 = (int64_t)R * P;
  PTA_REQUIRE(M64 < (1LL << 31), PTA_E_ARG, "pta_gwb_idft_rng: R*P too large");
  int M = (int)M64;
  size_t shmem = 2 * IR_BUF * sizeof(double);
  static bool attr_set = false;
  if (!attr_set) {
    PTA_HIP(hipFuncSetAttribute((const void *)k_gwb_idft_rng, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    attr_set = true;
  }
  for (int col0 = 0; col0 < npts; col0 += IR_LD) {
    hipLaunchKernelGGL(k_gwb_idft_rng, dim3(pta_cdiv(M, 64)), dim3(256), shmem, pta_stream(stream), seed, r0, M, P, Nf, T, ldt,
                       npts, col0, G0, ldg);
    PTA_LAUNCH_CHECK();
  }
  return PTA_OK;
}

// ---- mix: G[r] = Mchol . G0[r]  (red_noise.py:268, moved behind the DFT) -------------------------
extern "C" int pta_gwb_mix(const double *Mchol, int P, const double *G0, int R, int npts, int64_t ldg, double *G, void *stream) {
  PTA_REQUIRE(Mchol && G0 && G, PTA_E_ARG, "pta_gwb_mix: NULL argument");
  PTA_REQUIRE(P > 0 && R > 0 && npts > 0 && ldg >= npts, PTA_E_ARG, "pta_gwb_mix: P=%d R=%d npts=%d", P, R, npts);
  const int64_t sr = (int64_t)P * ldg;
  for (int rb = 0; rb < R; rb += 32768) {
    int rc_ = (R - rb < 32768) ? (R - rb) : 32768;
    int rc = pta_dgemm_launch(0, P, npts, P, 1.0, Mchol, P, 1, G0 + rb * sr, ldg, 0.0, G + rb * sr, ldg, 0, rc_, 0, sr, sr,
                              pta_get_gemm_algo(), pta_stream(stream));
    if (rc != PTA_OK) return rc;
  }
  return PTA_OK;
}

// ---- interpolation onto the TOAs (red_noise.py:286-287) ------------------------------------------
__global__ void k_gwb_bracket(const double *__restrict__ ut, int npts, const double *__restrict__ toa_s, int N,
                              int32_t *__restrict__ jlo) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  double x = toa_s[i];
  double step = (ut[npts - 1] - ut[0]) / (double)(npts - 1);
  int j = (int)floor((x - ut[0]) / step);
  j = max(0, min(j, npts - 2));
  while (j > 0 && ut[j] > x) --j;               // last j with ut[j] <= x ...
  while (j < npts - 2 && ut[j + 1] <= x) ++j;   // ... clamped like numpy.interp
  jlo[i] = j;
}

extern "C" int pta_gwb_bracket(const double *ut, int npts, const double *toa_s, int N, int32_t *jlo, void *stream) {
  PTA_REQUIRE(ut && toa_s && jlo, PTA_E_ARG, "pta_gwb_bracket: NULL argument");
  PTA_REQUIRE(npts >= 2 && N > 0, PTA_E_ARG, "pta_gwb_bracket: npts=%d N=%d", npts, N);
  hipLaunchKernelGGL(k_gwb_bracket, dim3(pta_cdiv(N, 256)), dim3(256), 0, pta_stream(stream), ut, npts, toa_s, N, jlo);
  PTA_LAUNCH_CHECK();
  return PTA_OK;
}

__device__ __forceinline__ double pta_lerp(const double *__restrict__ g, const double *__restrict__ ut, int j, double x) {
  double slope = (g[j + 1] - g[j]) / (ut[j + 1] - ut[j]);  // numpy.interp: slope*(x - xp[j]) + fp[j]
  return slope * (x - ut[j]) + g[j];
}

__global__ void k_gwb_interp(const double *__restrict__ G, int64_t ldg, int P, const double *__restrict__ ut,
                             const double *__restrict__ toa_s, const int32_t *__restrict__ psr_of_toa,
                             const int32_t *__restrict__ jlo, int N, double scale, double *__restrict__ out, int64_t ld_out,
                             int accumulate) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  int r = blockIdx.y;
  if (i >= N) return;
  const double *g = G + ((int64_t)r * P + psr_of_toa[i]) * ldg;
  double v = pta_lerp(g, ut, jlo[i], toa_s[i]);
  if (scale != 1.0) v = v * scale;
  int64_t o = (int64_t)r * ld_out + i;
  out[o] = accumulate ? out[o] + v : v;
}

extern "C" int pta_gwb_interp(const double *G, int64_t ldg, int P, int npts, const double *ut, const double *toa_s,
                              const int32_t *psr_of_toa, const int32_t *jlo, int N, int R, double scale, double *out,
                              int64_t ld_out, int accumulate, void *stream) {
  PTA_REQUIRE(G && ut && toa_s && psr_of_toa && jlo && out, PTA_E_ARG, "pta_gwb_interp: NULL argument");
  PTA_REQUIRE(P > 0 && npts >= 2 && N > 0 && R > 0 && R <= 65535 && ldg >= npts && ld_out >= N, PTA_E_ARG,
              "pta_gwb_interp: P=%d npts=%d N=%d R=%d", P, npts, N, R);
  hipLaunchKernelGGL(k_gwb_interp, dim3(pta_cdiv(N, 256), R), dim3(256), 0, pta_stream(stream), G, ldg, P, ut, toa_s,
                     psr_of_toa, jlo, N, scale, out, ld_out, accumulate);
  PTA_LAUNCH_CHECK();
  return PTA_OK;
}
