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
// Two ideas on top of "DFT as GEMM":
//
// (1) Window symmetry halves the flops.  With c the centre of the output window and j = c + j',
//         x[c + j'] = E(j') - O(j'),   x[c - j'] = E(j') + O(j'),
//         E(j') = sum_k amp_k a'_k cos(2 pi j' k / n),   O(j') = sum_k amp_k b'_k sin(2 pi j' k / n),
//     where (a'_k + i b'_k) = (Re w_k + i Im w_k) e^{2 pi i c k / n} is the draw rotated by a fixed,
//     realisation-independent phase.  So npts outputs cost two [rows x Kf] . [Kf x npts/2] products.
//
// (2) The A operand never touches memory.  Workgroup = 4 waves = 64 rows (row m = realisation r * P +
//     pulsar a) x NT column tiles of the half window.  Per K-step of 4 frequency bins a lane is A-operand
//     element (row l & 15, bin l >> 4) of v_mfma_f64_16x16x4_f64 - exactly ONE complex draw w[a, k], which
//     it generates in registers (Philox + Box-Muller), rotates, and feeds to the E tiles (Re') and the O
//     tiles (Im').  The twiddle slab of the K-step (2 planes x 4 bins x NT*16 columns) is pre-laid-out in
//     global memory exactly as it sits in LDS, shared by the 4 waves, double buffered; all workgroups of an
//     XCD walk it in lock-step, so it streams from L2 and HBM sees it once.
template <int NT>
struct pta_sym_cfg {
  static constexpr int COLS = NT * 16;
  // LDS/global row pitch == 16 (mod 32) doubles: the two bins a 32-lane group reads land on disjoint banks
  static constexpr int PITCH = (NT % 2) ? COLS : COLS + 16;
  static constexpr int SLAB = 8 * PITCH;               // doubles per K-step: [plane 2][bin 4][PITCH]
  static constexpr int NLD = (SLAB / 2 + 255) / 256;   // double2 loads per thread per slab
  static constexpr int SLAB_PAD = NLD * 512;           // slab stride, padded so that the staging copy needs no guards
};

// Tsym[chunk][ks][plane][kk][PITCH] and rot[(ks*4+kk)*2 + {0,1}] = (cos, sin)(2 pi c k / n), zero padded.
template <int NT>
__global__ void k_gwb_twiddle_sym(const double *__restrict__ sqrtC, int Nf, int npts, int i0, double inv_dt,
                                  double *__restrict__ Tsym, double *__restrict__ rot) {
  using C = pta_sym_cfg<NT>;
  const int Kf = Nf - 2, nstep = (Kf + 3) >> 2;
  const int half = (npts + 1) >> 1;
  const int col = blockIdx.x * blockDim.x + threadIdx.x;  // column inside the chunk, < PITCH
  const int kq = blockIdx.y;                               // padded bin index, < nstep*4
  const int chunk = blockIdx.z;
  if (col >= C::PITCH) return;
  const int64_t n = 2 * (int64_t)Nf - 2;
  const int p = chunk * C::COLS + col;
  double c = 0.0, s = 0.0;
  if (kq < Kf && col < C::COLS && p < half) {
    int64_t twoj = (npts & 1) ? 2 * (int64_t)p : 2 * (int64_t)p + 1;  // 2 j'
    int64_t m2 = (twoj * (int64_t)(kq + 1)) % (2 * n);
    double sn, cs;
    sincospi((double)m2 / (double)n, &sn, &cs);
    double amp = 2.0 * inv_dt / (double)n * sqrtC[kq + 1];
    c = amp * cs;
    s = amp * sn;
  }
  const int ks = kq >> 2, kk = kq & 3;
  double *slab = Tsym + ((int64_t)chunk * nstep + ks) * C::SLAB_PAD;
  slab[(0 * 4 + kk) * C::PITCH + col] = c;
  slab[(1 * 4 + kk) * C::PITCH + col] = s;
  if (chunk == 0 && col == 0) {
    double sn = 0.0, cs = 0.0;
    if (kq < Kf) {
      int64_t m2 = ((2 * (int64_t)i0 + npts - 1) * (int64_t)(kq + 1)) % (2 * n);  // 2 c k mod 2n
      sincospi((double)m2 / (double)n, &sn, &cs);
    }
    rot[2 * kq] = cs;
    rot[2 * kq + 1] = sn;
  }
}

template <int NT, int MINW>
__global__ __launch_bounds__(256, MINW) void k_gwb_idft_sym_rng(uint64_t seed, uint64_t r0, int M, int P, int Nf,
                                                                const double *__restrict__ Tsym,
                                                                const double *__restrict__ rot, int npts,
                                                                double *__restrict__ G0, int64_t ldg, int fast) {
  using C = pta_sym_cfg<NT>;
  extern __shared__ __attribute__((aligned(16))) double lds[];  // 2 x SLAB_PAD
  const int t = threadIdx.x, l = t & 63, wv = t >> 6;
  const int Kf = Nf - 2, nstep = (Kf + 3) >> 2;
  const int chunk = blockIdx.y;
  const int m = blockIdx.x * 64 + wv * 16 + (l & 15);
  const int mr = min(m, M - 1);
  const uint64_t real = r0 + (uint64_t)(mr / P);
  const uint32_t strm = pta_stream_id(PTA_STREAM_GWB, (uint32_t)(mr % P));
  pta_f64x4 accE[NT], accO[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    accE[i] = pta_f64x4{0.0, 0.0, 0.0, 0.0};
    accO[i] = pta_f64x4{0.0, 0.0, 0.0, 0.0};
  }
  const double2 *src = reinterpret_cast<const double2 *>(Tsym + (int64_t)chunk * nstep * C::SLAB_PAD) + t;
  double2 *lds2 = reinterpret_cast<double2 *>(lds) + t;
  double2 regs[C::NLD];
#pragma unroll
  for (int q = 0; q < C::NLD; ++q) lds2[q * 256] = src[q * 256];
  pta_rng_stage_tables();  // Box-Muller tables -> LDS (pta_rng.h)
  __syncthreads();
  const int kk = l >> 4;
  for (int ks = 0; ks < nstep; ++ks) {
    const double *cur = lds + (ks & 1) * C::SLAB_PAD;
    double2 *nxt = lds2 + ((ks + 1) & 1) * (C::SLAB_PAD / 2);
    // next slab: global loads fly under the MFMAs below (the last step harmlessly re-stages its own slab,
    // which keeps the staging registers free of control flow)
    const int ksn = min(ks + 1, nstep - 1);
#pragma unroll
    for (int q = 0; q < C::NLD; ++q) regs[q] = src[(int64_t)ksn * (C::SLAB_PAD / 2) + q * 256];
    double re, im;
    pta_normal_pair(seed, real, strm, (uint32_t)(ks * 4 + kk + 1), re, im, fast);  // pair k <-> w[a,k] (red_noise.py:240)
    const double2 cs = reinterpret_cast<const double2 *>(rot)[ks * 4 + kk];
    const double ar = re * cs.x - im * cs.y;  // rotated draw
    const double br = re * cs.y + im * cs.x;
    const double *bc = cur + kk * C::PITCH + (l & 15);
    const double *bs = bc + 4 * C::PITCH;
#pragma unroll
    for (int i = 0; i < NT; ++i) accE[i] = pta_mfma_f64(ar, bc[i * 16], accE[i]);
#pragma unroll
    for (int i = 0; i < NT; ++i) accO[i] = pta_mfma_f64(br, bs[i * 16], accO[i]);
#pragma unroll
    for (int q = 0; q < C::NLD; ++q) nxt[q * 256] = regs[q];
    __syncthreads();
  }
  const int half = (npts + 1) >> 1;
  const int hi0 = npts >> 1;                         // even: c + 1/2 ; odd: the centre sample
  const int lo0 = (npts & 1) ? hi0 : hi0 - 1;
#pragma unroll
  for (int i = 0; i < NT; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int row = blockIdx.x * 64 + wv * 16 + pta_mfma_row(l, r);
      int p = chunk * C::COLS + i * 16 + pta_mfma_col(l);
      if (row < M && p < half) {
        double e = accE[i][r], o = accO[i][r];
        G0[(int64_t)row * ldg + hi0 + p] = e - o;
        G0[(int64_t)row * ldg + lo0 - p] = e + o;
      }
    }
  }
}

// column tiling of a workgroup: variant 0: NT = 19 (whole half window per workgroup, 1 wave/SIMD), 1 (recommended): NT = 10
// (2 chunks, 2 waves/SIMD), 2: NT = 7 (3 chunks).  The Tsym layout depends on it, so the same value goes to all three calls.
static int sym_nt(int variant) { return variant == 1 ? 10 : (variant == 2 ? 7 : 19); }
static int sym_pitch(int nt) { return (nt % 2) ? nt * 16 : nt * 16 + 16; }
static int sym_slab_pad(int nt) { return ((8 * sym_pitch(nt) / 2 + 255) / 256) * 512; }

extern "C" int64_t pta_gwb_twiddle_sym_size(int Nf, int npts, int variant, int64_t *rot_doubles) {
  const int nt = sym_nt(variant);
  const int nstep = (Nf - 2 + 3) >> 2;
  const int half = (npts + 1) >> 1;
  const int nchunk = (half + nt * 16 - 1) / (nt * 16);
  if (rot_doubles) *rot_doubles = (int64_t)nstep * 8;
  return (int64_t)nchunk * nstep * sym_slab_pad(nt);
}

extern "C" int pta_gwb_twiddle_sym(const double *sqrtC, int Nf, int npts, int i0, double inv_dt, double *Tsym, double *rot,
                                   int variant, void *stream) {
  PTA_REQUIRE(sqrtC && Tsym && rot, PTA_E_ARG, "pta_gwb_twiddle_sym: NULL argument");
  PTA_REQUIRE(Nf >= 3 && npts > 0 && i0 >= 0, PTA_E_ARG, "pta_gwb_twiddle_sym: Nf=%d npts=%d", Nf, npts);
  const int nt = sym_nt(variant);
  const int nstep = (Nf - 2 + 3) >> 2;
  const int half = (npts + 1) >> 1;
  const int nchunk = (half + nt * 16 - 1) / (nt * 16);
  PTA_REQUIRE(nstep * 4 <= 65535 && nchunk <= 65535, PTA_E_ARG, "pta_gwb_twiddle_sym: problem too large");
  dim3 g(pta_cdiv(sym_pitch(nt), 64), nstep * 4, nchunk);
  if (nt == 19)
    hipLaunchKernelGGL(k_gwb_twiddle_sym<19>, g, dim3(64), 0, pta_stream(stream), sqrtC, Nf, npts, i0, inv_dt, Tsym, rot);
  else if (nt == 10)
    hipLaunchKernelGGL(k_gwb_twiddle_sym<10>, g, dim3(64), 0, pta_stream(stream), sqrtC, Nf, npts, i0, inv_dt, Tsym, rot);
  else
    hipLaunchKernelGGL(k_gwb_twiddle_sym<7>, g, dim3(64), 0, pta_stream(stream), sqrtC, Nf, npts, i0, inv_dt, Tsym, rot);
  PTA_LAUNCH_CHECK();
  return PTA_OK;
}

extern "C" int pta_gwb_idft_rng(uint64_t seed, uint64_t r0, int R, int P, int Nf, const double *Tsym, const double *rot,
                                int npts, double *G0, int64_t ldg, int variant, int rng_fast, void *stream) {
  PTA_REQUIRE(Tsym && rot && G0, PTA_E_ARG, "pta_gwb_idft_rng: NULL argument");
  PTA_REQUIRE(R > 0 && P > 0 && P < (1 << 24) && Nf >= 3 && npts > 0 && ldg >= npts, PTA_E_ARG,
              "pta_gwb_idft_rng: R=%d P=%d Nf=%d npts=%d", R, P, Nf, npts);
  int64_t M64 = (int64_t)R * P;
  PTA_REQUIRE(M64 < (1LL << 31), PTA_E_ARG, "pta_gwb_idft_rng: R*P too large");
  const int M = (int)M64;
  const int nt = sym_nt(variant);
  rng_fast = rng_fast ? 1 : 0;
  const int half = (npts + 1) >> 1;
  const int nchunk = (half + nt * 16 - 1) / (nt * 16);
  const size_t shmem = 2 * (size_t)sym_slab_pad(nt) * sizeof(double);
  dim3 g(pta_cdiv(M, 64), nchunk);
  if (nt == 19) {
    auto kern = k_gwb_idft_sym_rng<19, 1>;
    PTA_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    hipLaunchKernelGGL(kern, g, dim3(256), shmem, pta_stream(stream), seed, r0, M, P, Nf, Tsym, rot, npts, G0, ldg, rng_fast);
  } else if (nt == 10) {
    auto kern = k_gwb_idft_sym_rng<10, 2>;
    PTA_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    hipLaunchKernelGGL(kern, g, dim3(256), shmem, pta_stream(stream), seed, r0, M, P, Nf, Tsym, rot, npts, G0, ldg, rng_fast);
  } else {
    auto kern = k_gwb_idft_sym_rng<7, 2>;
    PTA_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    hipLaunchKernelGGL(kern, g, dim3(256), shmem, pta_stream(stream), seed, r0, M, P, Nf, Tsym, rot, npts, G0, ldg, rng_fast);
  }
  PTA_LAUNCH_CHECK();
  return PTA_OK;
}

// ---- mix: G[r] = Mchol . G0[r]  (red_noise.py:268, moved behind the DFT) -------------------------
// Array-sized specialisation (P <= 80): the generic 64x64-tile GEMM spends half its matrix-core work and a second read of
// G0 on the 4 rows that 68 pulsars spill into a second tile.  Here Mchol sits in LDS K-major (lane <-> pulsar, conflict
// free), every wave owns 16 grid samples and all ceil(P/16) pulsar tiles, G0 rows stream in as the B operand straight from
// global memory (128-byte segments) and each workgroup amortises the LDS fill over `rpw` realisations.  HBM bound:
// 16 bytes per (realisation, pulsar, sample).
#define MIX_JT 64
template <int NTA>
__global__ __launch_bounds__(256) void k_gwb_mix_small(const double *__restrict__ Mchol, int P, const double *__restrict__ G0, int R,
                                                       int npts, int64_t ldg, double *__restrict__ G, int rpw) {
  extern __shared__ double Ms[];  // [KP][MP]: Ms[b * MP + a] = Mchol[a][b], zero padded
  constexpr int MP = NTA * 16;
  const int KP = (P + 3) & ~3;
  const int t = threadIdx.x, l = t & 63, wv = t >> 6;
  for (int idx = t; idx < KP * MP; idx += 256) {
    const int b = idx / MP, a = idx - b * MP;
    Ms[idx] = (a < P && b < P) ? Mchol[(int64_t)a * P + b] : 0.0;
  }
  __syncthreads();
  const int col = l & 15, quad = l >> 4;
  const int jj = blockIdx.x * MIX_JT + wv * 16 + col;
  const bool jin = jj < npts;
  for (int rr = 0; rr < rpw; ++rr) {
    const int r = blockIdx.y * rpw + rr;
    if (r >= R) break;
    const double *__restrict__ g0 = G0 + (int64_t)r * P * ldg;
    pta_f64x4 acc[NTA];
#pragma unroll
    for (int i = 0; i < NTA; ++i) acc[i] = pta_f64x4{0.0, 0.0, 0.0, 0.0};
    double bv[NTA * 4];  // all K-steps' G0 samples of this realisation in flight at once (KP / 4 <= NTA * 4)
#pragma unroll
    for (int s4 = 0; s4 < NTA * 4; ++s4) {
      const int b = 4 * s4 + quad;
      bv[s4] = (b < P && jin) ? g0[(int64_t)b * ldg + jj] : 0.0;
    }
#pragma unroll
    for (int s4 = 0; s4 < NTA * 4; ++s4) {
      if (4 * s4 < KP) {  // uniform
        const int b = 4 * s4 + quad;
#pragma unroll
        for (int i = 0; i < NTA; ++i) acc[i] = pta_mfma_f64(Ms[b * MP + i * 16 + col], bv[s4], acc[i]);
      }
    }
    if (jin) {
#pragma unroll
      for (int i = 0; i < NTA; ++i)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int a = i * 16 + pta_mfma_row(l, rg);
          if (a < P) G[((int64_t)r * P + a) * ldg + jj] = acc[i][rg];
        }
    }
  }
}

// variant: 0 = specialised kernel when P <= 80 (else the generic batched MFMA GEMM), 1 = always the generic MFMA GEMM,
// 2 = the generic VALU reference GEMM
extern "C" int pta_gwb_mix(const double *Mchol, int P, const double *G0, int R, int npts, int64_t ldg, double *G, int variant,
                           void *stream) {
  PTA_REQUIRE(Mchol && G0 && G, PTA_E_ARG, "pta_gwb_mix: NULL argument");
  PTA_REQUIRE(P > 0 && R > 0 && npts > 0 && ldg >= npts, PTA_E_ARG, "pta_gwb_mix: P=%d R=%d npts=%d", P, R, npts);
  const int64_t sr = (int64_t)P * ldg;
  if (variant == 0 && P <= 80) {
    const int nta = (P + 15) / 16, kp = (P + 3) & ~3;
    const int rpw = 4;
    const size_t shmem = (size_t)kp * nta * 16 * sizeof(double);
    dim3 g(pta_cdiv(npts, MIX_JT), pta_cdiv(R, rpw));
    PTA_REQUIRE(g.y <= 65535u, PTA_E_ARG, "pta_gwb_mix: R=%d too large for one launch", R);
#define PTA_MIX(N) hipLaunchKernelGGL(k_gwb_mix_small<N>, g, dim3(256), shmem, pta_stream(stream), Mchol, P, G0, R, npts, ldg, G, rpw)
    switch (nta) {
      case 1: PTA_MIX(1); break;
      case 2: PTA_MIX(2); break;
      case 3: PTA_MIX(3); break;
      case 4: PTA_MIX(4); break;
      default: PTA_MIX(5); break;
    }
#undef PTA_MIX
    PTA_LAUNCH_CHECK();
    return PTA_OK;
  }
  for (int rb = 0; rb < R; rb += 32768) {
    int rc_ = (R - rb < 32768) ? (R - rb) : 32768;
    int rc = pta_dgemm_launch(0, P, npts, P, 1.0, Mchol, P, 1, G0 + rb * sr, ldg, 0.0, G + rb * sr, ldg, 0, rc_, 0, sr, sr,
                              variant == 2 ? 0 : 1, pta_stream(stream));
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

__global__ void k_gwb_weights(const double *__restrict__ ut, int npts, const double *__restrict__ toa_s,
                              const int32_t *__restrict__ jlo, int N, double *__restrict__ w) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  int j = max(0, min(jlo[i], npts - 2));
  w[i] = (toa_s[i] - ut[j]) / (ut[j + 1] - ut[j]);
}

extern "C" int pta_gwb_weights(const double *ut, int npts, const double *toa_s, const int32_t *jlo, int N, double *w, void *stream) {
  PTA_REQUIRE(ut && toa_s && jlo && w, PTA_E_ARG, "pta_gwb_weights: NULL argument");
  PTA_REQUIRE(npts >= 2 && N > 0, PTA_E_ARG, "pta_gwb_weights: npts=%d N=%d", npts, N);
  hipLaunchKernelGGL(k_gwb_weights, dim3(pta_cdiv(N, 256)), dim3(256), 0, pta_stream(stream), ut, npts, toa_s, jlo, N, w);
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
