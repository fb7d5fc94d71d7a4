// GWB frequency -> time stage as a chirp-z (Bluestein) transform: one workgroup per (realisation, pulsar) row,
// two in-LDS 4096-point fp64 FFTs, draws generated in registers.  See pta_fft.h for the algebra.
// Replaces, per row, the Hermitian pack + np.fft.ifft + crop of red_noise.py:275-285 (the sqrt(C) scaling of :270 is
// folded into the pre-chirp, the 1/dt and ifft normalisation into the post-chirp, the M @ w mix of :268 follows on
// the 600-sample grid in pta_gwb_mix).
#include "pta_common.h"
#include "pta_fft.h"

// setup: twiddles, pre-chirp (with sqrtC), post-chirp (with 2/(n dt)), and the spectrum of the convolution chirp
__global__ __launch_bounds__(PTA_FFT_THREADS) void k_czt_setup(const double *__restrict__ sqrtC, int Nf, int npts, int i0,
                                                               double inv_dt, double *__restrict__ pre,
                                                               double *__restrict__ FB, double *__restrict__ tw,
                                                               double *__restrict__ post) {
  __shared__ double re[PTA_FFT_PLANE], im[PTA_FFT_PLANE];
  const int tid = threadIdx.x;
  const int Kf = Nf - 2;
  const int64_t n = 2 * (int64_t)Nf - 2;
  for (int m = tid; m < PTA_FFT_N; m += PTA_FFT_THREADS) {
    double s, c;
    sincospi((double)(2 * m) / (double)PTA_FFT_N, &s, &c);
    tw[2 * m] = c;
    tw[2 * m + 1] = -s;  // e^{-2 pi i m / 4096}
    // pre-chirp sqrtC_k e^{+ pi i k^2 / n}, k = m + 1
    double pr = 0.0, pi_ = 0.0;
    if (m < Kf) {
      int64_t k = m + 1, q = (k * k) % (2 * n);
      sincospi((double)q / (double)n, &s, &c);
      pr = sqrtC[k] * c;
      pi_ = sqrtC[k] * s;
    }
    pre[2 * m] = pr;
    pre[2 * m + 1] = pi_;
    // convolution chirp b_u = e^{- pi i u^2 / n} on u = j - k in [i0 - Kf, i0 + npts - 2], wrapped mod 4096
    int64_t u = m;
    bool used = (m <= i0 + npts - 2);
    if (!used && m >= PTA_FFT_N - (Kf - i0)) {
      u = (int64_t)m - PTA_FFT_N;
      used = true;
    }
    double br = 0.0, bi = 0.0;
    if (used) {
      int64_t q = (u * u) % (2 * n);
      sincospi((double)q / (double)n, &s, &c);
      br = c;
      bi = -s;
    }
    re[PTA_FFT_PHYS(m)] = br;
    im[PTA_FFT_PHYS(m)] = bi;
  }
  for (int jj = tid; jj < npts; jj += PTA_FFT_THREADS) {  // post-chirp (2/(n dt)) e^{+ pi i j^2 / n}
    int64_t j = i0 + jj, q = (j * j) % (2 * n);
    double s, c;
    sincospi((double)q / (double)n, &s, &c);
    double amp = 2.0 * inv_dt / (double)n;
    post[2 * jj] = amp * c;
    post[2 * jj + 1] = amp * s;
  }
  __threadfence_block();
  __syncthreads();
  // (the passes read the twiddles this same block just wrote)
  pta_fft_pass<false, 9>(re, im, tw, tid);
  __syncthreads();
  pta_fft_pass<false, 6>(re, im, tw, tid);
  __syncthreads();
  pta_fft_pass<false, 3>(re, im, tw, tid);
  __syncthreads();
  pta_fft_pass<false, 0>(re, im, tw, tid);
  __syncthreads();
  for (int m = tid; m < PTA_FFT_N; m += PTA_FFT_THREADS) {  // logical (digit-reversed) order, 1/L of the inverse folded in
    // stored butterfly-transposed: element q of the s = 1 butterfly b (logical index 8 b + q) at [q][b], so that the eight
    // loads of the fused kernel are 1 KB-contiguous per wave instead of 64 lanes x 128-byte stride
    const int slot = (m & 7) * (PTA_FFT_N / 8) + (m >> 3);
    FB[2 * slot] = re[PTA_FFT_PHYS(m)] * (1.0 / PTA_FFT_N);
    FB[2 * slot + 1] = im[PTA_FFT_PHYS(m)] * (1.0 / PTA_FFT_N);
  }
}

static bool czt_fits(int Nf, int npts, int i0) { return Nf >= 3 && (Nf - 2) + npts - 2 < PTA_FFT_N && i0 >= 1 && (Nf - 2) <= PTA_FFT_N; }

extern "C" int pta_gwb_czt_setup(const double *sqrtC, int Nf, int npts, int i0, double inv_dt, double *pre, double *FB, double *tw,
                                 double *post, void *stream) {
  PTA_REQUIRE(sqrtC && pre && FB && tw && post, PTA_E_ARG, "pta_gwb_czt_setup: NULL argument");
  PTA_REQUIRE(czt_fits(Nf, npts, i0), PTA_E_ARG, "pta_gwb_czt_setup: Nf=%d npts=%d does not fit one 4096-point convolution", Nf, npts);
  hipLaunchKernelGGL(k_czt_setup, dim3(1), dim3(PTA_FFT_THREADS), 0, pta_stream(stream), sqrtC, Nf, npts, i0, inv_dt, pre, FB, tw, post);
  PTA_LAUNCH_CHECK();
  return PTA_OK;
}

extern "C" int pta_gwb_czt_fits(int Nf, int npts, int i0) { return czt_fits(Nf, npts, i0) ? 1 : 0; }

// One workgroup per (realisation, pulsar) row.  FUSE selects how much stays in registers between LDS exchanges:
//   bit 0: forward s=1 butterfly, chirp-spectrum product and inverse s=1 butterfly in one go (two exchanges fewer);
//   bit 1: the last butterfly writes the npts-sample window straight to memory;
//   bit 2: the first butterfly starts from registers (draws generated per thread for its own 512-strided elements);
//   bit 3: one twiddle load per butterfly, the rest by squaring / products (pta_fft_twiddles<.., 1>).
// FUSE = 15 is the production kernel: 6 LDS exchanges instead of 10, and - what mattered most on MI355X - 6 instead of 56
// global twiddle loads per thread (the q-strided twiddle gathers, not LDS or the barriers, bounded the first version:
// 3.2 -> 2.0 ms per 65 280 rows; rocprofv3 SQ_INSTS_VALU now accounts for ~95 % of the time).  FUSE = 0 (every stage through
// LDS, table twiddles) is kept as the cross-check.
template <bool RNG, bool FAST, int FUSE>
__global__ __launch_bounds__(PTA_FFT_THREADS, 4) void k_gwb_czt(uint64_t seed, uint64_t r0, const double *__restrict__ w, int64_t ldw,
                                                                  int M, int P, int Nf, int npts, int i0,
                                                                  const double *__restrict__ pre, const double *__restrict__ FB,
                                                                  const double *__restrict__ tw, const double *__restrict__ post,
                                                                  double *__restrict__ G0, int64_t ldg) {
  __shared__ double re[PTA_FFT_PLANE], im[PTA_FFT_PLANE];
  const int tid = threadIdx.x;
  const int row = blockIdx.x;
  const int Kf = Nf - 2;
  if (RNG) {
    pta_rng_stage_tables();  // Box-Muller tables -> LDS (pta_rng.h)
    __syncthreads();
  }
  const uint64_t real = r0 + (uint64_t)(row / P);
  const uint32_t strm = pta_stream_id(PTA_STREAM_GWB, (uint32_t)(row % P));
  const pta_cplx *pre2 = reinterpret_cast<const pta_cplx *>(pre);
  constexpr int TW = (FUSE & 8) ? 1 : 3;
  // the inverse passes use the same twiddles as the forward ones; through an opaque copy of the pointer the compiler reloads
  // them (L1/L2 hits) instead of carrying 28 registers per pass across the whole kernel in scratch
  const double *twi = tw;
  asm volatile("" : "+s"(twi));
  if (FUSE & 4) {
    int b, o;
    pta_cplx v[8], tv[8];
    pta_fft_map<9>(tid, b, o);
    pta_fft_twiddles<9, TW>(tw, o, tv);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int t = o + 512 * q;
      v[q] = {0.0, 0.0};
      if (t < Kf) {  // uniform per q except in the one 512-block that straddles Kf
        double wr, wi;
        if (RNG) {
          pta_normal_pair(seed, real, strm, (uint32_t)(t + 1), wr, wi, FAST ? 1 : 0);
        } else {
          wr = w[(int64_t)row * ldw + 2 * (t + 1)];
          wi = w[(int64_t)row * ldw + 2 * (t + 1) + 1];
        }
        const pta_cplx pc = pre2[t];
        v[q] = {wr * pc.re - wi * pc.im, wr * pc.im + wi * pc.re};
      }
    }
    pta_fft_core<false, 9>(v, tv);
    pta_fft_store<9>(re, im, b, o, v);
  } else {
#pragma unroll 1
    for (int i = 0; i < PTA_FFT_N / PTA_FFT_THREADS; ++i) {
      const int t = tid + PTA_FFT_THREADS * i;
      double ar = 0.0, ai = 0.0;
      if (t < Kf) {
        double wr, wi;
        if (RNG) {
          pta_normal_pair(seed, real, strm, (uint32_t)(t + 1), wr, wi, FAST ? 1 : 0);
        } else {
          wr = w[(int64_t)row * ldw + 2 * (t + 1)];
          wi = w[(int64_t)row * ldw + 2 * (t + 1) + 1];
        }
        const pta_cplx pc = pre2[t];
        ar = wr * pc.re - wi * pc.im;
        ai = wr * pc.im + wi * pc.re;
      }
      re[PTA_FFT_PHYS(t)] = ar;
      im[PTA_FFT_PHYS(t)] = ai;
    }
    __syncthreads();
    pta_fft_pass<false, 9, TW>(re, im, tw, tid);
  }
  __syncthreads();
  pta_fft_pass<false, 6, TW>(re, im, tw, tid);
  __syncthreads();
  pta_fft_pass<false, 3, TW>(re, im, tw, tid);
  __syncthreads();
  if (FUSE & 1) {
    int b, o;
    pta_cplx v[8], fv[8];
    pta_fft_map<0>(tid, b, o);
    const pta_cplx *fb = reinterpret_cast<const pta_cplx *>(FB) + b;
#pragma unroll
    for (int q = 0; q < 8; ++q) fv[q] = fb[q * (PTA_FFT_N / 8)];
    pta_fft_load<0>(re, im, b, o, v);
    pta_dft8<false>(v);
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = pta_cmul(v[q], fv[q]);
    pta_dft8<true>(v);
    pta_fft_store<0>(re, im, b, o, v);
  } else {
    pta_fft_pass<false, 0, TW>(re, im, tw, tid);
    __syncthreads();
#pragma unroll 4
    for (int i = 0; i < PTA_FFT_N / PTA_FFT_THREADS; ++i) {
      const int t = tid + PTA_FFT_THREADS * i;
      const int p = PTA_FFT_PHYS(t);
      const int slot = (t & 7) * (PTA_FFT_N / 8) + (t >> 3);
      const double xr = re[p], xi = im[p], fr = FB[2 * slot], fi = FB[2 * slot + 1];
      re[p] = xr * fr - xi * fi;
      im[p] = xr * fi + xi * fr;
    }
    __syncthreads();
    pta_fft_pass<true, 0, TW>(re, im, twi, tid);
  }
  __syncthreads();
  pta_fft_pass<true, 3, TW>(re, im, twi, tid);
  __syncthreads();
  pta_fft_pass<true, 6, TW>(re, im, twi, tid);
  __syncthreads();
  const pta_cplx *post2 = reinterpret_cast<const pta_cplx *>(post);
  if (FUSE & 2) {
    int b, o;
    pta_cplx v[8], tv[8];
    pta_fft_map<9>(tid, b, o);
    pta_fft_twiddles<9, TW>(twi, o, tv);
    pta_fft_load<9>(re, im, b, o, v);
    pta_fft_core<true, 9>(v, tv);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int jj = o + 512 * q - (i0 - 1);
      if (jj >= 0 && jj < npts) {
        const pta_cplx pp = post2[jj];
        G0[(int64_t)row * ldg + jj] = v[q].re * pp.re - v[q].im * pp.im;
      }
    }
  } else {
    pta_fft_pass<true, 9, TW>(re, im, twi, tid);
    __syncthreads();
    for (int jj = tid; jj < npts; jj += PTA_FFT_THREADS) {
      const int p = PTA_FFT_PHYS(i0 - 1 + jj);
      G0[(int64_t)row * ldg + jj] = re[p] * post2[jj].re - im[p] * post2[jj].im;
    }
  }
}

// w == NULL: draws generated on chip (throughput mode); else w[M x ldw] interleaved (re, im) rows (replay mode)
extern "C" int pta_gwb_czt(uint64_t seed, uint64_t r0, const double *w, int64_t ldw, int R, int P, int Nf, int npts, int i0,
                           const double *pre, const double *FB, const double *tw, const double *post, double *G0, int64_t ldg,
                           int variant, int rng_fast, void *stream) {
  PTA_REQUIRE(pre && FB && tw && post && G0, PTA_E_ARG, "pta_gwb_czt: NULL argument");
  PTA_REQUIRE(R > 0 && P > 0 && P < (1 << 24) && npts > 0 && ldg >= npts, PTA_E_ARG, "pta_gwb_czt: R=%d P=%d npts=%d", R, P, npts);
  PTA_REQUIRE(czt_fits(Nf, npts, i0), PTA_E_ARG, "pta_gwb_czt: Nf=%d npts=%d does not fit one 4096-point convolution", Nf, npts);
  PTA_REQUIRE(!w || ldw == 0 || ldw >= 2 * (int64_t)Nf, PTA_E_ARG, "pta_gwb_czt: ldw too small");  // 0 = one row of draws for all rows
  int64_t M64 = (int64_t)R * P;
  PTA_REQUIRE(M64 < (1LL << 31), PTA_E_ARG, "pta_gwb_czt: R*P too large");
  const int M = (int)M64;
  // variant: 0 = fully fused kernel (default), 1 = plain cross-check (FUSE = 0), 10 + FUSE = any ladder step
  const int fastm = rng_fast ? 1 : 0;
  const int fuse = variant == 0 ? 15 : (variant == 1 ? 0 : variant - 10);
#define PTA_CZT_X(RNGV, FASTV, FUSEV)                                                                                                  \
  hipLaunchKernelGGL((k_gwb_czt<RNGV, FASTV, FUSEV>), dim3(M), dim3(PTA_FFT_THREADS), 0, pta_stream(stream), seed, r0, w, ldw, M, P, \
                     Nf, npts, i0, pre, FB, tw, post, G0, ldg)
#define PTA_CZT_XF(FUSEV)              \
  if (w)                               \
    PTA_CZT_X(false, false, FUSEV);    \
  else if (fastm)                      \
    PTA_CZT_X(true, true, FUSEV);      \
  else                                 \
    PTA_CZT_X(true, false, FUSEV)
  switch (fuse) {
    case 15: PTA_CZT_XF(15); break;
    case 0: PTA_CZT_XF(0); break;
    case 1: PTA_CZT_XF(1); break;
    case 3: PTA_CZT_XF(3); break;
    case 7: PTA_CZT_XF(7); break;
    case 8: PTA_CZT_XF(8); break;
    default: pta_set_error("pta_gwb_czt: unknown variant %d", variant); return PTA_E_ARG;
  }
#undef PTA_CZT_XF
#undef PTA_CZT_X
  PTA_LAUNCH_CHECK();
  return PTA_OK;
}
