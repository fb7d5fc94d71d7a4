// Throughput mode: one fused pass writes R whole-array realisations,
//     out[r, i] = RN + GWB + EFAC/EQUAD + ECORR + deterministic,
// with every Gaussian deviate produced in registers (pta_rng.h).  This is the kernel the
// realisations/sec metric is quoted on; its algorithmic traffic is the 8 bytes per (realisation, TOA)
// it writes - everything it reads is realisation independent (Ft, noise vectors, epoch
// map) or tiny per realisation (60 RN coefficients and a 600-sample GWB row per pulsar).
#include "pta_common.h"
#include "pta_rng.h"
#include "pta_mfma.h"

// coef[(r*P + a)*K + c] = amp[a*K + c] * z,  z = deviate c of stream (RN, a)   (red_noise.py:126-127)
__global__ void k_engine_rn_coef(uint64_t seed, uint64_t r0, int R, int P, int K, const double *__restrict__ amp,
                                 double *__restrict__ coef, int fast) {
  pta_rng_stage_tables();  // Box-Muller tables -> LDS (pta_rng.h)
  __syncthreads();
  int idx = blockIdx.x * blockDim.x + threadIdx.x;  // (r, a, pair)
  int hp = K / 2;
  int total = R * P * hp;
  if (idx >= total) return;
  int p = idx % hp, ra = idx / hp;
  int a = ra % P, r = ra / P;
  double z0, z1;
  pta_normal_pair(seed, r0 + (uint64_t)r, pta_stream_id(PTA_STREAM_RN, (uint32_t)a), (uint32_t)p, z0, z1, fast);
  int64_t o = (int64_t)ra * K + 2 * p;
  coef[o] = amp[(int64_t)a * K + 2 * p] * z0;
  coef[o + 1] = amp[(int64_t)a * K + 2 * p + 1] * z1;
}

extern "C" int pta_engine_rn_coef(uint64_t seed, uint64_t r0, int R, int P, int K, const double *amp, double *coef, int rng_fast,
                                  void *stream) {
  PTA_REQUIRE(amp && coef, PTA_E_ARG, "pta_engine_rn_coef: NULL argument");
  PTA_REQUIRE(R > 0 && P > 0 && K > 0 && (K % 2) == 0, PTA_E_ARG, "pta_engine_rn_coef: R=%d P=%d K=%d (K must be even)", R, P, K);
  int64_t total = (int64_t)R * P * (K / 2);
  PTA_REQUIRE(total < (1LL << 31), PTA_E_ARG, "pta_engine_rn_coef: problem too large");
  hipLaunchKernelGGL(k_engine_rn_coef, dim3(pta_cdiv(total, 256)), dim3(256), 0, pta_stream(stream), seed, r0, R, P, K, amp, coef, rng_fast ? 1 : 0);
  PTA_LAUNCH_CHECK();
  return PTA_OK;
}

// One workgroup = one tile of <= 256 consecutive TOAs of ONE pulsar x ENG_RB realisations; one thread = one TOA.
//  * the pulsar and the realisations are workgroup-uniform, so the 60 red-noise coefficients and the GWB row
//    base come through the scalar unit (s_load) and feed v_fma_f64 directly - no per-lane gathers;
//  * the per-TOA vectors (Ft column, noise levels, bracket) are read once and reused for ENG_RB realisations;
//  * ECORR deviates are shared by the TOAs of an epoch, and Box-Muller yields them in pairs: the tile draws the
//    contiguous pair range its epochs cover once into LDS instead of one pair per TOA (halves that RNG work).
#define ENG_RB 8

template <int MINW>
__global__ __launch_bounds__(PTA_ENGINE_TILE, MINW) void k_engine_synth(pta_engine_plan pl, uint64_t seed, uint64_t r0, int R,
                                                                  double *__restrict__ out, int64_t ld_out, int fast) {
  __shared__ double zec[ENG_RB][2 * PTA_ENGINE_EPMAX];
  // realisation groups are the FAST grid axis: the workgroups that share a tile's Ft columns / noise vectors run
  // back to back and hit L2 (with tiles fastest every sweep re-read the whole 163 MB design matrix from HBM:
  // rocprofv3 FETCH_SIZE 11.6 GB per launch against 2.6 GB written)
  pta_rng_stage_tables();  // Box-Muller tables -> LDS (pta_rng.h)
  __syncthreads();
  const int tile = blockIdx.y;
  const int rb = blockIdx.x * ENG_RB;
  const int a = pl.tile_psr[tile];
  const int start = pl.tile_start[tile];
  const int count = pl.tile_count[tile];
  const int t = threadIdx.x;
  const int P = pl.n_psr;
  const bool has_ec = pl.ecorr_toa != nullptr;
  const int epn = has_ec ? pl.tile_epn[tile] : 0;
  const int ep0 = has_ec ? pl.tile_ep0[tile] : 0;
  if (epn > 0) {
    const uint32_t strm = pta_stream_id(PTA_STREAM_ECORR, (uint32_t)a);
    for (int idx = t; idx < epn * ENG_RB; idx += PTA_ENGINE_TILE) {
      int q = idx / epn, p = idx - q * epn;
      double z0, z1;
      pta_normal_pair(seed, r0 + (uint64_t)(rb + q), strm, (uint32_t)(ep0 + p), z0, z1, fast);
      zec[q][2 * p] = z0;
      zec[q][2 * p + 1] = z1;
    }
    __syncthreads();
  }
  if (t >= count) return;
  const int i = start + t;
  double v[ENG_RB];
#pragma unroll
  for (int q = 0; q < ENG_RB; ++q) v[q] = 0.0;

  if (pl.rn_k > 0) {  // red noise: dt = F @ y (red_noise.py:128); coefficients are wave-uniform
    const int K = pl.rn_k;
    const double *__restrict__ Fcol = pl.Ft + i;
    const double *__restrict__ cf = pl.rn_coef + ((int64_t)rb * P + a) * K;
    const int64_t rstride = (int64_t)P * K;
    for (int c = 0; c < K; ++c) {
      double fv = Fcol[(int64_t)c * pl.ldf];
#pragma unroll
      for (int q = 0; q < ENG_RB; ++q) {
        int qq = (rb + q < R) ? q : 0;
        v[q] = fma(fv, cf[qq * rstride + c], v[q]);
      }
    }
  }
  if (pl.gw_npts > 0) {  // GWB: interpolate the mixed grid series (red_noise.py:286-287)
    const int j = pl.gw_jlo[i];
    const double wgt = pl.gw_w[i];
    const double *__restrict__ gbase = pl.gw_G + ((int64_t)rb * P + a) * pl.gw_npts;
    const int64_t rstride = (int64_t)P * pl.gw_npts;
#pragma unroll
    for (int q = 0; q < ENG_RB; ++q) {
      int qq = (rb + q < R) ? q : 0;
      const double *g = gbase + qq * rstride;
      v[q] = v[q] + ((g[j + 1] - g[j]) * wgt + g[j]);
    }
  }
  if (pl.wn_a) {  // EFAC/EQUAD: (efac sigma) z1 + (efac equad | equad) z2 (white_noise.py:105-109)
    const double wa = pl.wn_a[i], wb = pl.wn_b[i];
    const uint32_t strm = pta_stream_id(PTA_STREAM_WN, (uint32_t)a);
    const uint32_t pair = (uint32_t)pl.idx_in_psr[i];
#pragma unroll
    for (int q = 0; q < ENG_RB; ++q) {
      double z1, z2;
      pta_normal_pair(seed, r0 + (uint64_t)(rb + q), strm, pair, z1, z2, fast);
      v[q] = v[q] + (wa * z1 + wb * z2);
    }
  }
  if (has_ec) {  // ECORR: ecorr[e(i)] z[e(i)] (white_noise.py:182)
    const double ec = pl.ecorr_toa[i];
    const int e = pl.epoch_of[i];
    if (epn > 0) {
      const int o = e - 2 * ep0;
#pragma unroll
      for (int q = 0; q < ENG_RB; ++q) v[q] = v[q] + ec * zec[q][o];
    } else if (ec != 0.0) {
      const uint32_t strm = pta_stream_id(PTA_STREAM_ECORR, (uint32_t)a);
#pragma unroll
      for (int q = 0; q < ENG_RB; ++q) v[q] = v[q] + ec * pta_normal_single(seed, r0 + (uint64_t)(rb + q), strm, (uint32_t)e, fast);
    }
  }
  const double det = pl.det ? pl.det[i] : 0.0;
#pragma unroll
  for (int q = 0; q < ENG_RB; ++q)
    if (rb + q < R) out[(int64_t)(rb + q) * ld_out + i] = v[q] + det;
}

// MFMA variant (default).  The ablation of the VALU kernel above (profiles/) shows the red-noise F @ y loop as its largest
// single term (1.5 of 4.1 ms at 68 x 5000, R = 960): 60 dependent scalar-load + FMA steps per TOA.  F @ y for a tile IS a small
// dense product - [16 realisations x 60] . [60 x 256 TOAs] - so it goes to the otherwise idle matrix cores:
// one wave = 16 realisations x 64 TOAs = 4 tiles of v_mfma_f64_16x16x4_f64, 15 K-steps; lane l supplies the coefficient of
// realisation (l & 15), bin (l >> 4) and the design-matrix entry of bin (l >> 4), TOA (l & 15) of each tile, and receives the
// sums for realisations (l >> 4) + 4 g, TOA (l & 15): exactly the (4 TOAs x 4 realisations) it then finishes on the VALU
// (GWB interpolation, EFAC/EQUAD and ECORR deviates, deterministic term) and stores as 128-byte row segments.
// two doubles loaded as one 16-byte access from an address that is only 8-byte aligned (global loads need dword alignment)
typedef double pta_f64x2_a8 __attribute__((ext_vector_type(2), aligned(8)));
#define ENG_MR 16  // realisations per workgroup (MFMA M)
#define ENG_ZPITCH (2 * PTA_ENGINE_EPMAX + 8)  // = 16 mod 32 doubles: the two realisation rows a half-wave reads sit on disjoint banks

// (SINGLE keeps four deviates alive from a lane's even TOA to its odd one: 8 VGPRs more than the 128 that four workgroups per CU allow -
// it is compiled for three, where the occupancy curve of the kernel is already flat: 4.9 against 4.8 ms at 3 / 4 workgroups per CU)
template <bool FAST, bool SINGLE>
__global__ __launch_bounds__(PTA_ENGINE_TILE, SINGLE ? 3 : 4) void k_engine_synth_mfma(pta_engine_plan pl, uint64_t seed, uint64_t r0, int R,
                                                                           double *__restrict__ out, int64_t ld_out, int xcd_aware) {
  constexpr int fast = FAST ? 1 : 0;  // template parameters: the RNG-math modes and the single-deviate white noise are separate kernels
                                      // (and profile rows) - folded into one, the extra live values spill the default path
  __shared__ double zec[ENG_MR][ENG_ZPITCH];
  // 1-D launch, XCD-aware: hardware deals consecutive workgroups round-robin to the 8 XCDs, so workgroup `lin` is given the
  // work item (lin % 8) * chunk + lin / 8 - each XCD walks its own contiguous range of (tile, realisation group) items and a
  // tile's design-matrix slab and per-TOA vectors live in ONE L2 instead of eight.
  // (Measured, not kept - VERDICT r2 #3ii: the same kernel PERSISTENT, 4 workgroups per CU walking their XCD's items so that the
  // Box-Muller tables are staged once per workgroup: 5.56 against 4.81 ms per step.  The plan's ~25 pointers then have to stay live
  // across items - 94 SGPR spills - and the per-item body loses 12 VGPRs to the loop state at a 128-VGPR budget.)
  const int nrg = (R + ENG_MR - 1) / ENG_MR;
  const int64_t total = (int64_t)nrg * pl.n_tiles, chunk = (total + 7) >> 3;
  const int64_t item = xcd_aware ? (int64_t)(blockIdx.x & 7) * chunk + (blockIdx.x >> 3) : (int64_t)blockIdx.x;
  if (item >= total) return;  // workgroup-uniform
  pta_rng_stage_tables();  // Box-Muller tables -> LDS (pta_rng.h)
  __syncthreads();
  const int t = threadIdx.x, l = t & 63, wv = t >> 6;
  const int P = pl.n_psr;
  const bool has_ec = pl.ecorr_toa != nullptr;
  constexpr bool wn_single = SINGLE;            // opt-in: ONE deviate per TOA, amplitude sqrt((efac sigma)^2 + (efac equad | equad)^2)
  const int col = l & 15, quad = l >> 4;
  const int tbase = wv * 64 + col;  // TOA (inside the tile) of MFMA tile 0; tile j adds 16 j
  const int tile = (int)(item / nrg);
  const int rb = (int)(item - (int64_t)tile * nrg) * ENG_MR;
  const int a = pl.tile_psr[tile];
  const int start = pl.tile_start[tile];
  const int count = pl.tile_count[tile];
  const int epn = has_ec ? pl.tile_epn[tile] : 0;
  const int ep0 = has_ec ? pl.tile_ep0[tile] : 0;
  if (epn > 0) {
    const uint32_t strm = pta_stream_id(PTA_STREAM_ECORR, (uint32_t)a);
    for (int idx = t; idx < epn * ENG_MR; idx += PTA_ENGINE_TILE) {
      int q = idx / epn, p = idx - q * epn;
      double z0, z1;
      pta_normal_pair(seed, r0 + (uint64_t)(rb + q), strm, (uint32_t)(ep0 + p), z0, z1, fast);
      zec[q][2 * p] = z0;
      zec[q][2 * p + 1] = z1;
    }
    __syncthreads();
  }
  // No predicated stores anywhere below: a lane beyond the tile's count works on the tile's LAST TOA and a realisation row beyond
  // R on realisation R - 1 - same counters, same operands, bit-identical value - and stores it again to the same address.  (Stores
  // count in vmcnt on gfx950 and a predicated store is its own exec-masked block: behind a pending load the compiler waits
  // vmcnt(0) - for the PREVIOUS STORE's completion - in every such block, 12 store round trips per lane.)
  int rq[4];  // this lane's four realisation rows inside the group, clamped
#pragma unroll
  for (int g = 0; g < 4; ++g) rq[g] = min(quad + 4 * g, R - 1 - rb);
  int jl4[4];  // GWB bracket indices of this lane's four TOAs: requested now, needed (as addresses) after the red-noise product
#pragma unroll
  for (int j = 0; j < 4; ++j) jl4[j] = pl.gw_npts > 0 ? pl.gw_jlo[start + min(tbase + 16 * j, count - 1)] : 0;
  pta_f64x4 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = pta_f64x4{0.0, 0.0, 0.0, 0.0};
  if (pl.rn_k > 0) {
    const int K = pl.rn_k;
    const int ra = min(rb + col, R - 1);
    const double *__restrict__ cf = pl.rn_coef + ((int64_t)ra * P + a) * K;
    const double *__restrict__ Fb = pl.Ft + start;
    // branch-free body - every load is unconditional, from a clamped (always valid) address: a bin beyond K enters with a zero
    // coefficient, a TOA beyond the tile's count lands in an accumulator column the epilogue never stores - so that the compiler
    // can unroll the loop and keep several K-steps' loads in flight (with predicated loads it emitted one exec-masked block and
    // one s_waitcnt vmcnt(0) per step: 15 serialised L2 round trips per workgroup)
    int tcl[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) tcl[j] = min(tbase + 16 * j, count - 1);
    // three K-steps in flight: the operands of steps k0 + 4 and k0 + 8 are being loaded while step k0 runs on the matrix cores
    // (rotation by name, not by register moves; loads past K re-read bin K - 1 and are dropped)
    auto ld = [&](int k0, double &av, double (&bv)[4]) {
      const int k = k0 + quad, kc = min(k, K - 1);
      const double a_ld = cf[kc];
      av = (k < K) ? a_ld : 0.0;
      const double *__restrict__ Fk = Fb + (int64_t)kc * pl.ldf;
#pragma unroll
      for (int j = 0; j < 4; ++j) bv[j] = Fk[tcl[j]];
    };
    auto mm = [&](double av, const double (&bv)[4]) {
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = pta_mfma_f64(av, bv[j], acc[j]);
    };
    double a0, a1, a2, b0[4], b1[4], b2[4];
    ld(0, a0, b0);
    ld(4, a1, b1);
    for (int k0 = 0; k0 < K; k0 += 12) {
      ld(k0 + 8, a2, b2);
      mm(a0, b0);
      ld(k0 + 12, a0, b0);
      mm(a1, b1);  // steps past K carry a zero coefficient: no exits inside the rotation
      ld(k0 + 16, a1, b1);
      mm(a2, b2);
    }
  }
  const uint32_t strm_wn = pta_stream_id(PTA_STREAM_WN, (uint32_t)a);
  const uint32_t strm_ec = pta_stream_id(PTA_STREAM_ECORR, (uint32_t)a);
  // single-deviate white noise (opt-in): TOA idx (index inside its pulsar; tiles start at multiples of 256) takes branch (idx >> 4) & 1
  // of the Box-Muller pair idx & ~16 - for a lane these are its TOAs j (branch 0) and j + 1 (branch 1), so a FULL tile draws one pair
  // per two TOAs; a tile cut short by the end of the pulsar (clamped lanes) evaluates the pair of every TOA on its own
  const bool wn_share = wn_single && count == PTA_ENGINE_TILE;  // workgroup-uniform
  double zs[4] = {0.0, 0.0, 0.0, 0.0};
  // Loads and stores share the in-order vmcnt on gfx950: waiting for a load also waits for every store issued before it.  The four
  // stores of iteration j are therefore issued one iteration late, BEHIND the operand loads of iteration j + 1 - they then drain
  // under that iteration's Box-Muller chains instead of in front of its first use of a loaded value.
  double pend[4] = {0.0, 0.0, 0.0, 0.0};
  int pend_i = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int i = start + min(tbase + 16 * j, count - 1);
    // every per-TOA operand of this j is requested up front: a load issued between the Box-Muller chains is waited for on the
    // spot
    const bool has_gw = pl.gw_npts > 0, has_wn = pl.wn_a != nullptr || wn_single;
    const double wgt = has_gw ? pl.gw_w[i] : 0.0;
    pta_f64x2_a8 y[4];
    if (has_gw) {  // GWB: both bracket samples of the mixed grid series in one 16-byte load (red_noise.py:286-287)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        y[g] = *reinterpret_cast<const pta_f64x2_a8 *>(pl.gw_G + ((int64_t)(rb + rq[g]) * P + a) * pl.gw_npts + jl4[j]);
    }
    const double wa = has_wn ? (wn_single ? pl.wn_c[i] : pl.wn_a[i]) : 0.0, wb = (has_wn && !wn_single) ? pl.wn_b[i] : 0.0;
    const uint32_t pair = (uint32_t)pl.idx_in_psr[i];
    const double ec = has_ec ? pl.ecorr_toa[i] : 0.0;
    const int e = has_ec ? pl.epoch_of[i] : 0;
    const double det = pl.det ? pl.det[i] : 0.0;
    if (j > 0) {
#pragma unroll
      for (int g = 0; g < 4; ++g) out[(int64_t)(rb + rq[g]) * ld_out + pend_i] = pend[g];
    }
    double v[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) v[g] = acc[j][g];
    if (has_gw) {
#pragma unroll
      for (int g = 0; g < 4; ++g) v[g] = v[g] + ((y[g].y - y[g].x) * wgt + y[g].x);
    }
    if (has_wn && !wn_single) {  // EFAC/EQUAD, the reference's two deviates per TOA (white_noise.py:105-109)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        double z1, z2;
        pta_normal_pair(seed, r0 + (uint64_t)(rb + rq[g]), strm_wn, pair, z1, z2, fast);
        v[g] = v[g] + (wa * z1 + wb * z2);
      }
    } else if (wn_single) {
      // pair and branch of this TOA, per lane; a lane's TOAs j (even) and j + 1 share a pair unless the tile is cut short by the end of
      // the pulsar and the lane is clamped to its last TOA - then (a wave-level test: rare, the last tile of a pulsar only) the odd
      // step evaluates its own pair
      const uint32_t pid = pair & ~16u;
      const bool br = (pair & 16u) != 0;
      // (the odd step of a short tile used to reuse zs whenever no lane's pair id differed from the even step's - true also
      // when both steps are clamped to the pulsar's LAST TOA, whose own branch was then replaced by its partner's.  A short tile now
      // evaluates every step: workgroup-uniform, the last tile of a pulsar only.)
      const bool fresh = (j & 1) == 0 || !wn_share;
      if (fresh) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          double z0, z1;
          pta_normal_pair(seed, r0 + (uint64_t)(rb + rq[g]), strm_wn, pid, z0, z1, fast);
          v[g] = v[g] + wa * (br ? z1 : z0);
          zs[g] = br ? z0 : z1;   // the other branch, for the partner TOA
        }
      } else {
#pragma unroll
        for (int g = 0; g < 4; ++g) v[g] = v[g] + wa * zs[g];
      }
    }
    if (has_ec) {  // ECORR (white_noise.py:182)
      if (epn > 0) {
        const int o = e - 2 * ep0;
#pragma unroll
        for (int g = 0; g < 4; ++g) v[g] = v[g] + ec * zec[rq[g]][o];
      }
#ifndef PTA_ISA_TABLE_MAIN_PATH_ONLY  // scripts/isa_instruction_table.py counts the staged-ECORR path only (the one the bench workload takes)
      else if (ec != 0.0) {
#pragma unroll
        for (int g = 0; g < 4; ++g) v[g] = v[g] + ec * pta_normal_single(seed, r0 + (uint64_t)(rb + rq[g]), strm_ec, (uint32_t)e, fast);
      }
#endif
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) pend[g] = v[g] + det;
    pend_i = i;
  }
#pragma unroll
  for (int g = 0; g < 4; ++g) out[(int64_t)(rb + rq[g]) * ld_out + pend_i] = pend[g];
}

extern "C" int pta_engine_synth(const pta_engine_plan *plan_host, uint64_t seed, uint64_t r0, int R, double *out, int64_t ld_out,
                                void *stream) {
  PTA_REQUIRE(plan_host && out, PTA_E_ARG, "pta_engine_synth: NULL argument");
  const pta_engine_plan &p = *plan_host;
  PTA_REQUIRE(p.n_toa > 0 && p.n_psr > 0 && R > 0 && ld_out >= p.n_toa, PTA_E_ARG, "pta_engine_synth: n_toa=%d n_psr=%d R=%d", p.n_toa,
              p.n_psr, R);
  PTA_REQUIRE(p.n_tiles > 0 && p.tile_psr && p.tile_start && p.tile_count && p.tile_ep0 && p.tile_epn && p.idx_in_psr, PTA_E_ARG,
              "pta_engine_synth: tile table / idx_in_psr missing");
  PTA_REQUIRE(p.rn_k == 0 || (p.Ft && p.rn_coef && p.ldf >= p.n_toa), PTA_E_ARG, "pta_engine_synth: red-noise inputs missing");
  PTA_REQUIRE(p.gw_npts == 0 || (p.gw_G && p.gw_jlo && p.gw_w && p.gw_npts >= 2), PTA_E_ARG,
              "pta_engine_synth: GWB inputs missing");
  PTA_REQUIRE(!p.wn_a || p.wn_b, PTA_E_ARG, "pta_engine_synth: wn_b missing");
  PTA_REQUIRE(!p.wn_c || !p.wn_a, PTA_E_ARG, "pta_engine_synth: wn_c (single-deviate white noise) replaces wn_a / wn_b - pass one or the other");
  PTA_REQUIRE(!p.ecorr_toa || p.epoch_of, PTA_E_ARG, "pta_engine_synth: epoch_of missing");
  // 0 = MFMA kernel (default); 1 = same, linear workgroup order; 4 / 6 / 8 = all-VALU kernel; 100 + k = MFMA kernel with k KB of
  // unused dynamic LDS per workgroup - the occupancy probe of round 3 (script since removed; result in DESIGN.md §4.1): the 34 KB ECORR staging buffer allows 4
  // workgroups per CU (+12 KB: 3, +20 KB: 2).  Measured step time 10.3 / 7.2 / 5.9 / 5.5 ms at 1 / 2 / 3 / 4 workgroups; a two-pass
  // staging variant (17 KB, 6 per CU, 3 barriers) measured 6.0 ms at 6 AND when padded back to 4: occupancy is saturated at 4.
  const int variant = p.synth_variant;
  const int rng_fast = p.rng_fast ? 1 : 0;
  // 0 = MFMA kernel (default); 1 = same, linear workgroup order (A/B of the XCD mapping); 4 / 6 / 8 = all-VALU kernel; 100 + k = MFMA
  // kernel with k KB of unused dynamic LDS per workgroup - the occupancy probe of round 3 (script since removed; result in DESIGN.md §4.1): the 34 KB ECORR
  // staging buffer allows 4 workgroups per CU (+12 KB: 3, +20 KB: 2)
  if (variant == 0 || variant == 1 || (variant >= 100 && variant <= 164)) {
    const int xcd = variant == 1 ? 0 : 1;
    const unsigned pad = variant >= 100 ? (unsigned)(variant - 100) * 1024u : 0u;
    const int64_t total = (int64_t)pta_cdiv(R, ENG_MR) * p.n_tiles, nwg = ((total + 7) >> 3) << 3;
    PTA_REQUIRE(nwg < (1LL << 31), PTA_E_ARG, "pta_engine_synth: %lld workgroups exceed one launch", (long long)nwg);
    auto kern = p.wn_c ? (rng_fast ? k_engine_synth_mfma<true, true> : k_engine_synth_mfma<false, true>)
                       : (rng_fast ? k_engine_synth_mfma<true, false> : k_engine_synth_mfma<false, false>);
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(PTA_ENGINE_TILE), pad, pta_stream(stream), p, seed, r0, R, out, ld_out, xcd);
    PTA_LAUNCH_CHECK();
    return PTA_OK;
  }
  PTA_REQUIRE(!p.wn_c, PTA_E_ARG, "pta_engine_synth: the single-deviate white noise exists in the MFMA kernel only");
  PTA_REQUIRE(p.n_tiles <= 65535, PTA_E_ARG, "pta_engine_synth: %d tiles exceed the 2-D launch of the all-VALU kernel", p.n_tiles);
  dim3 g(pta_cdiv(R, ENG_RB), p.n_tiles), b(PTA_ENGINE_TILE);
  // register budget per lane (waves per SIMD the compiler must allow): the kernel alternates long Box-Muller chains
  // with a load-fed FMA loop, so occupancy matters more than keeping all eight chains' temporaries in registers
  if (variant >= 8)
    hipLaunchKernelGGL(k_engine_synth<8>, g, b, 0, pta_stream(stream), p, seed, r0, R, out, ld_out, rng_fast);
  else if (variant >= 6)
    hipLaunchKernelGGL(k_engine_synth<6>, g, b, 0, pta_stream(stream), p, seed, r0, R, out, ld_out, rng_fast);
  else
    hipLaunchKernelGGL(k_engine_synth<4>, g, b, 0, pta_stream(stream), p, seed, r0, R, out, ld_out, rng_fast);
  PTA_LAUNCH_CHECK();
  return PTA_OK;
}

extern "C" int pta_gwb_czt(uint64_t seed, uint64_t r0, const double *w, int64_t ldw, int R, int P, int Nf, int npts, int i0,
                           const double *pre, const double *FB, const double *tw, const double *post, double *G0, int64_t ldg,
                           int variant, int rng_fast, void *stream);
extern "C" int pta_gwb_idft_rng(uint64_t seed, uint64_t r0, int R, int P, int Nf, const double *Tsym, const double *rot, int npts,
                                double *G0, int64_t ldg, int variant, int rng_fast, void *stream);
extern "C" int pta_gwb_mix(const double *Mchol, int P, const double *G0, int R, int npts, int64_t ldg, double *G, int variant,
                           void *stream);

extern "C" int pta_engine_generate(const pta_engine_plan *plan_host, const pta_engine_tables *tables_host, uint64_t seed, uint64_t r0,
                                   int R, double *out, int64_t ld_out, void *stream) {
  PTA_REQUIRE(plan_host && tables_host && out, PTA_E_ARG, "pta_engine_generate: NULL argument");
  pta_engine_plan p = *plan_host;
  const pta_engine_tables &tb = *tables_host;
  int rc;
  if (p.rn_k > 0) {
    PTA_REQUIRE(tb.rn_amp && tb.ws_coef, PTA_E_ARG, "pta_engine_generate: red-noise amplitudes / workspace missing");
    rc = pta_engine_rn_coef(seed, r0, R, p.n_psr, p.rn_k, tb.rn_amp, tb.ws_coef, p.rng_fast, stream);
    if (rc != PTA_OK) return rc;
    p.rn_coef = tb.ws_coef;
  }
  // (Measured, not kept - VERDICT r2 #3i: the batch as two halves with the HBM-bound ORF mix of one half on a side stream beside the
  // VALU-bound transform / synthesis of the other: 4.806 against 4.814 ms per step - the mix's workgroups only get the slots the
  // other kernel's retiring workgroups free, and the halves run at slightly lower efficiency.  pta_engine_rn_coef is 0.016 ms.)
  if (p.gw_npts > 0) {
    PTA_REQUIRE(tb.Mchol && tb.ws_G0 && tb.ws_G, PTA_E_ARG, "pta_engine_generate: GWB factor / workspace missing");
    if (tb.use_czt) {
      PTA_REQUIRE(tb.czt_pre && tb.czt_FB && tb.czt_tw && tb.czt_post, PTA_E_ARG, "pta_engine_generate: chirp-z tables missing");
      rc = pta_gwb_czt(seed, r0, nullptr, 0, R, p.n_psr, tb.gw_nf, p.gw_npts, tb.gw_i0, tb.czt_pre, tb.czt_FB, tb.czt_tw, tb.czt_post,
                       tb.ws_G0, p.gw_npts, tb.czt_variant, p.rng_fast, stream);
    } else {
      PTA_REQUIRE(tb.Tsym && tb.rot, PTA_E_ARG, "pta_engine_generate: DFT-GEMM tables missing");
      rc = pta_gwb_idft_rng(seed, r0, R, p.n_psr, tb.gw_nf, tb.Tsym, tb.rot, p.gw_npts, tb.ws_G0, p.gw_npts, tb.idft_variant, p.rng_fast, stream);
    }
    if (rc != PTA_OK) return rc;
    rc = pta_gwb_mix(tb.Mchol, p.n_psr, tb.ws_G0, R, p.gw_npts, p.gw_npts, tb.ws_G, tb.mix_variant, stream);
    if (rc != PTA_OK) return rc;
    p.gw_G = tb.ws_G;
  }
  return pta_engine_synth(&p, seed, r0, R, out, ld_out, stream);
}
