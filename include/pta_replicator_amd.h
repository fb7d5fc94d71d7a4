/*
 * pta_replicator_amd — C ABI of the MI355X (gfx950) stochastic-injection hot path.
 *
 * The reference (bencebecsy/pta_replicator) is pure Python with no FFI layer of its own
 * (SURVEY.md §8b); this header therefore defines the boundary a maintainer would bind with
 * ctypes (INTEGRATION.md shows the stub).  Each entry point cites the reference code it
 * replaces as file:line under /root/reference/pta_replicator/.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes; no torch / C++ types.
 *  - Every `double*` / `int32_t*` is a DEVICE pointer to contiguous memory owned by the
 *    caller (e.g. torch.Tensor.data_ptr()) unless the parameter name ends in `_host`.
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream). Calls are
 *    asynchronous on that stream; no allocation, no synchronisation inside.
 *  - Return 0 on success, a negative PTA_E_* code otherwise; pta_last_error() gives the
 *    thread-local message. No exceptions cross the boundary.
 *  - Batches: R realisations are the leading axis; `ld_*` are row strides in elements.
 *    `accumulate` = 0 overwrites `out`, 1 adds into it.
 *  - Everything is IEEE binary64, like the reference (it casts PINT's longdouble columns to
 *    float64 before any arithmetic: red_noise.py:123, :287).
 *  - No process-wide state: every option that changes what a call computes or how (Gaussian-transform mode, kernel
 *    variants) is an argument or a field of the plan struct passed to that call, so concurrent callers on different
 *    streams / threads cannot influence each other (ABI 2; ABI 1 had pta_set_* switches).
 */
#ifndef PTA_REPLICATOR_AMD_H
#define PTA_REPLICATOR_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PTA_ABI_VERSION 8

#define PTA_OK 0
#define PTA_E_ARG (-1)     /* bad argument (sizes, NULL pointers, unsupported lmax ...) */
#define PTA_E_HIP (-2)     /* HIP runtime error; message carries hipGetErrorString */
#define PTA_E_NOTPD (-3)   /* reserved: matrix not positive definite (see `info` of pta_potrf_batched) */

/* ---------------------------------------------------------------- library ---------- */
int pta_abi_version(void);
const char *pta_last_error(void);
/* number of CUs / wavefront size / gcnArchName of the current HIP device */
int pta_device_info(int *cu_count, int *wavefront, char *arch, int arch_len);

/* ---------------------------------------------------------------- RNG -------------- */
/* Throughput-mode draws. The reference consumes NumPy's global legacy stream
 * (red_noise.py:119,127,176,238-240; white_noise.py:80,105-109,155,182); on the device every
 * deviate is Philox-4x32-10(key = seed, counter = (pair, stream, realisation)) + Box-Muller.
 * stream ids: (kind << 24) | pulsar, kind = 1 GWB, 2 RN, 3 WN, 4 ECORR, 5 TD (N_a x N_a factor), 6 TDGW (GWB grid factor). */

/* `rng_fast` (argument or plan field of every call that draws on chip) selects the Gaussian transform: 0 (default) = fp64
 * Box-Muller with < 1 ulp log / sincos, 1 = "fast RNG math": the same uniforms through the hardware fp32 log / sqrt / sin /
 * cos (deviates accurate to ~1e-6).  It is part of what defines realisation r: all ranks of a job must use the same value
 * (pta_replicator_amd.distributed checks).                                                                              */

/* Known-answer access to the raw generator: out[i] = 4 x u32 of
 * Philox4x32-10(counter = ctr[i][0..3], key = key[0..1]).  ctr/key/out are device u32.   */
int pta_rng_philox_raw(const uint32_t *ctr, const uint32_t *key, int n, uint32_t *out, void *stream);

/* Dump the normals the fused kernels would use, in the layout the replay kernels read:
 *   interleave = 1: z0[r*ld + 2p] = first, z0[r*ld + 2p + 1] = second deviate of pair p (z1 unused)
 *   interleave = 0: z0[r*ld + p] = first, z1[r*ld + p] = second deviate of pair p
 * for pairs p in [0, npairs), realisations r0 .. r0+R-1.                                   */
int pta_rng_fill_normal(uint64_t seed, uint64_t r0, int R, uint32_t stream_id, int npairs, int interleave,
                        double *z0, double *z1, int64_t ld, int rng_fast, void *stream);
/* The same fill for all blocks of a pta_td_plan in ONE launch: z[r * ld + blk_zoff[b] + j] = deviate j of stream (stream_kind, b),
 * j < blk_n[b] rounded up to even (pairs are written whole), r < R.  blk_n / blk_zoff: device arrays (blk_zoff even), max_n = the
 * largest blk_n; ld even, z 16-byte aligned.                                                                                    */
int pta_rng_fill_normal_blocks(uint64_t seed, uint64_t r0, int R, uint32_t stream_kind, int n_blocks, const int32_t *blk_n,
                               const int32_t *blk_zoff, int max_n, double *z, int64_t ld, int rng_fast, void *stream);

/* ---------------------------------------------------------------- red noise -------- */
/* Fourier design matrix, transposed: Ft[c*N + i] = column c of F for TOA i.
 * Replaces create_fourier_design_matrix_red (red_noise.py:36-103).
 *   arg = ((2 pi) * (t[i] - t_ref)) * f[k] + phase[k]        (phase may be NULL = 0)
 *   cos_first = 0 (default convention, t_ref = 0): c = 2k sin, c = 2k+1 cos   (:98-101)
 *   cos_first = 1 (libstempo convention, t_ref = t[0]): c = 2k cos, c = 2k+1 sin (:92-96) */
int pta_rn_basis(const double *t, int N, double t_ref, const double *f, const double *phase, int nmodes,
                 int cos_first, double *Ft, int64_t ldf, void *stream);

/* out[r*ld_out + i] (+)= sum_c Ft[c*ldf + i] * coef[r*ld_coef + c],  c < K.
 * Replaces dt = F @ (sqrt(prior) * randn) (red_noise.py:127-128); coef = sqrt(prior)*z.    */
int pta_rn_synth(const double *Ft, int64_t ldf, int N, int K, const double *coef, int64_t ld_coef, int R, double *out,
                 int64_t ld_out, int accumulate, void *stream);

/* ---------------------------------------------------------------- white noise ------ */
/* out[r,i] (+)= (efac[i]*sigma[i])*z1[r,i] + (tnequad ? equad[i] : efac[i]*equad[i])*z2[r,i]
 * Replaces add_measurement_noise's arithmetic (white_noise.py:105-109).                    */
int pta_wn(const double *sigma, const double *efac, const double *equad, int N, int tnequad, const double *z1,
           const double *z2, int64_t ld_z, int R, double *out, int64_t ld_out, int accumulate, void *stream);

/* Greedy epoch bucketing on the HOST (realisation independent). Replaces quantize_fast
 * (white_noise.py:7-44) without the dense U: epoch_of_host[i] = column of U holding TOA i;
 * first_index_host[e] = first (earliest) TOA of epoch e, whose flag labels the epoch (:35).
 * order_host = np.argsort(times) to reproduce the reference's tie order exactly, or NULL for a
 * stable sort.  first_index_host must have room for N entries.                             */
int pta_quantize_epochs(const double *times_host, int N, double dt, const int64_t *order_host,
                        int32_t *epoch_of_host, int32_t *first_index_host, int *n_epochs);

/* HOST helper of the CW-catalogue path: out_host[i] = x_host[3i .. 3i+2] . y_host[0..2], summed as fma(x2, y2, fma(x1, y1, x0 * y0)) -
 * the association np.dot (OpenBLAS ddot) uses for the antenna-pattern dot products of deterministic.py:364-372, so that the
 * per-source scalars of a whole catalogue are bit-identical to the reference's per-source loop without a Python-level loop.   */
int pta_dot3_host(const double *x_host, int64_t n, const double *y_host, double *out_host);
/* out_host[i] = pow(x_host[i], y) through libm's scalar pow (what the reference's per-source scalar expressions evaluate to;
 * NumPy's array power differs from it by an ulp on a few per cent of arguments).                                        */
int pta_pow_host(const double *x_host, double y, int64_t n, double *out_host);
/* HOST helper of replay mode: NumPy's LEGACY normal stream, stream i = np.random.RandomState(seeds_host[i]).randn(counts_host[i])
 * value for value (MT19937 by init_genrand, 53-bit doubles, polar method with the cached second deviate: what np.random.seed(seed);
 * np.random.randn(n) of white_noise.py:79-80,105-109,154-155,182 and red_noise.py:112-113,127 yields), written to out_host +
 * offsets_host[i]; the streams are drawn on n_threads host threads (0 = one per core, at most 16).  The state the LAST stream is left in comes back
 * in last_key_host [624], last_pos_has_host [2] = (pos, has_gauss), last_gauss_host [1] (any NULL: not wanted) - the caller installs it
 * with np.random.set_state so that the global stream continues as after the reference's sequential calls.                       */
int pta_legacy_randn(const uint32_t *seeds_host, const int64_t *counts_host, const int64_t *offsets_host, int n_streams,
                     double *out_host, uint32_t *last_key_host, int32_t *last_pos_has_host, double *last_gauss_host, int n_threads);

/* out[r,i] (+)= ecorr_epoch[epoch_of[i]] * z[r*ld_z + epoch_of[i]]
 * Replaces dt = (U*ecorrvec) @ randn(E) (white_noise.py:182): a gather, not an N x E matvec. */
int pta_ecorr(const int32_t *epoch_of, const double *ecorr_epoch, int N, int E, const double *z, int64_t ld_z, int R,
              double *out, int64_t ld_out, int accumulate, void *stream);

/* ---------------------------------------------------------------- ORF -------------- */
/* locs[a*2+0] = phi (RA, rad), locs[a*2+1] = theta (colatitude, rad) (red_noise.py:205-223). */

/* HOST helper (ABI 8) of the ORF set-up: arg_host[a*P+b] = sin(theta_a) sin(theta_b) cos(phi_a - phi_b) + cos(theta_a) cos(theta_b), the
 * argument of calczeta's arccos (spharmORFbasis.py:24), evaluated left to right through libm's scalar sin / cos like the reference's
 * per-pair loop (spharmORFbasis.py:400-408) - in C++ instead of 20 100 Python-level iterations at 200 pulsars (61 ms -> < 1 ms).
 * same_host[a*P+b] = 1 where both coordinates are identical (the exact-equality branch :23).  Symmetric [P, P] host arrays.
 * arccos / cos of the result stay with NumPy (its float64 arccos is not libm's on AVX512 hosts): spharmORFbasis.pair_zeta_cos. */
int pta_orf_pair_arguments(const double *locs_host, int P, double *arg_host, uint8_t *same_host);

/* lmax = 0, clm = [sqrt(4 pi)] fast path: orf[a*P+b] = 2*HD(zeta_ab), 2 on zeta == 0.
 * Equals red_noise.py:224-226 with the defaults.                                           */
int pta_orf_hd(const double *locs, int P, double *orf, void *stream);

/* basis[(k*P + a)*P + b], k = l*l + (m + l): spharmORFbasis.correlated_basis
 * (spharmORFbasis.py:385-434 and callees :14-382). lmax <= 8.
 * zeta_cos (device, may be NULL): zeta_cos[2*(a*P+b) + 0/1] = the pair's angular separation zeta and cos(zeta) exactly as the
 * reference computes them (calczeta :14-35, np.cos at :166; host libm).  The l >= 3 sums are ill-conditioned in cos(zeta)
 * (~1e5-1e6), so 1-2 ulp between device and libm acos/cos were a 7e-10 parity error; with the host's values it is the sums'
 * own rounding only.  NULL = both are derived on the device.                                   */
int pta_orf_basis(const double *locs, const double *zeta_cos, int P, int lmax, double *basis, void *stream);

/* orf = 2 * sum_k clm[k] * basis[k]  (red_noise.py:225-226). clm is a device array.        */
int pta_orf_combine(const double *basis, const double *clm, int nbasis, int P, double *orf, void *stream);

/* In-place lower Cholesky of B row-major n x n matrices (reads the lower triangle, zeroes the
 * strict upper one - np.linalg.cholesky's result, red_noise.py:235).  info[b] = 0, or j+1 if the
 * leading minor of order j+1 is not positive definite (LAPACK convention).  Blocked right-looking:
 * LDS panel factorisation + fp64 MFMA (v_mfma_f64_16x16x4_f64) trailing update.             */
int pta_potrf_batched(double *A, int n, int B, int32_t *info, void *stream);

/* The same factorisation for matrices with a leading dimension and a batch stride: matrix b is row-major at A + b * strideA
 * with row pitch lda >= n.  Schedule: right-looking over panels of NB = 1024 columns whose trailing update runs with K = NB
 * (the 128 x 128-tile fp64 MFMA GEMM reaches 56 TFLOP/s at K = 1024 against 36-40 at K = 256); each panel is factored
 * RECURSIVELY - left half, update of the right half with K = half the width, right half - down to 64 columns, where the
 * diagonal block is factored and inverted in registers and the rows below are solved by an MFMA product with that inverse.
 * A batch is split into two independent chains of matrices on internal streams (created on first use, joined back into
 * `stream` before returning), so that one chain's serial, memory-bound panel steps overlap the other's MFMA-bound updates.
 * flags: PTA_POTRF_ZERO_UPPER zeroes the strict upper triangle (np.linalg.cholesky's result); without it the upper triangle
 * is left holding scratch (the parked inverses of the diagonal blocks) - all TD mode needs, since pta_td_trmm_rng reads the
 * lower triangle only.  PTA_POTRF_NO_LOOKAHEAD keeps every launch on `stream` (one chain).  PTA_POTRF_NB(k) overrides the
 * panel width with k * 256 columns, PTA_POTRF_CHAINS(c) the number of chains (1..4) - both for A/B timing.             */
#define PTA_POTRF_ZERO_UPPER 1
#define PTA_POTRF_NO_LOOKAHEAD 2
#define PTA_POTRF_NB(k) (((k) & 0xFF) << 8) /* panel width override, in units of 256 columns */
#define PTA_POTRF_CHAINS(c) (((c) & 0xF) << 16) /* number of concurrent chains of matrices (default 2) */
#define PTA_POTRF_VALU 8          /* cross-check: VALU reference GEMM and substitution panel solve instead of the MFMA kernels */
#define PTA_POTRF_SUBSTITUTION 4  /* panel solves by forward substitution instead of the MFMA product with the inverted diagonal
                                     block: LAPACK-grade backward error also when the diagonal blocks are very ill-conditioned
                                     (cond(L11) * eps enters the product form) - used for the GWB grid covariance, cond ~ 3e14 */
#define PTA_POTRF_REG_STAGING 32  /* A/B: the 128 x 128-tile updates stage their operand slabs through registers (pta_dgemm algo 1)
                                     instead of by LDS DMA (global_load_lds_dwordx4, pta_dgemm algo 2 - the default) */
#define PTA_POTRF_DIAG_AHEAD 128   /* workspace scheme: the next panel's diagonal phase runs on an internal side stream as soon as its
                                     block of the trailing update is done, beside the rest of that update */
#define PTA_POTRF_DIAG64 16         /* A/B (workspace scheme): the panel's diagonal block by the 64-column recursion (k_potf2 / k_trsm_mfma /
                                     k_syrk64) + one inversion pass instead of the 128-column base case k_diag128 */
#define PTA_POTRF_LOCKSTEP 64       /* A/B (workspace scheme): all chains start together instead of one diagonal phase apart */
#define PTA_POTRF_EPI1 0x100000     /* A/B: the 128 x 128-tile products prefetch their C tile behind the last slab and store interior tiles
                                       unpredicated (pta_dgemm algo 3; also honoured by pta_potrf_ragged_plan) */
#define PTA_POTRF_LEFT 0x200000     /* workspace scheme (ABI 8), LEFT-LOOKING panel order: a finished panel is not applied to the trailing matrix;
                                     * before a panel is factored its block column is updated once with everything to its left (one tile
                                     * product of K = the panel's first column: a C tile is read and written once, not once per panel) */
#define PTA_POTRF_LEFT_SPLIT 0x400000 /* with PTA_POTRF_LEFT: the part of that update that only needs panels <= q - 2 runs ahead on an internal
                                       * side stream beside panel q - 1's diagonal phase and substitution */
#define PTA_POTRF_SOLVE_ROWS 0x800000 /* workspace scheme (ABI 8): the substitution on the rows below a panel as ONE launch - a workgroup keeps a
                                       * 128-row tile and walks the panel's 128-column blocks (reading back its own stores through L2: sc1 loads)
                                       * instead of one launch per block */
int pta_potrf_batched_ex(double *A, int n, int64_t lda, int64_t strideA, int B, int32_t *info, int flags, void *stream);

/* The same factorisation with a caller-owned workspace (device memory, `work_doubles` doubles, at least
 * pta_potrf_workspace_doubles(n, B, flags) - B * 1152^2 doubles = 10.6 MB per matrix at the default panel width; 0 = the workspace
 * scheme does not apply: n <= panel width, or the VALU / SUBSTITUTION paths).  With it a panel is factored on its nbo x nbo DIAGONAL
 * block only - a recursion whose base case is a whole 128-column group, factored AND inverted (W_jj) by one workgroup per matrix -,
 * the strips S_j = [-W_jj L11[j, <j] | W_jj] go to the workspace, and the rows below are solved by blocked substitution - ONE tile product X_j = [X_{<j} | B_j] S_j^T per 128 columns,
 * K = 128 (j + 1), in place - instead of by the recursion over the panel's full height: no 64-column solves or K = 64 updates over
 * all rows, a quarter of the launches, and what is left of the latency-bound kernels works on the diagonal block's few rows.
 * PTA_POTRF_DIAG_AHEAD additionally runs the next panel's diagonal phase on an internal side stream beside the trailing update.
 * cond(diagonal blocks) * eps enters X, as it does through the 64 x 64 inverses of pta_potrf_batched_ex; ill-conditioned matrices
 * take PTA_POTRF_SUBSTITUTION (workspace ignored).  work == NULL or too small: identical to pta_potrf_batched_ex.              */
int64_t pta_potrf_workspace_doubles(int n, int B, int flags);
int pta_potrf_batched_ws(double *A, int n, int64_t lda, int64_t strideA, int B, int32_t *info, int flags, double *work,
                         int64_t work_doubles, void *stream);

/* Creates the internal streams / events the chained schedules of pta_potrf_batched_ws / pta_potrf_ragged use (per calling thread and
 * device; nchain <= 0: the default two chains + their look-ahead streams) NOW instead of inside the first factorisation - a HIP stream
 * is a hardware queue, ~10 ms each to create (ABI 7; profiles/r05_prepare_td_first_call.txt).  Optional: the schedules create on demand. */
int pta_potrf_warmup(int nchain);

/* RAGGED batch (ABI 6): B matrices of DIFFERENT orders factored as ONE schedule - the shape of a real pulsar timing array (the
 * reference's noise_dicts/ng15_dict.json: 68 pulsars, 68 TOA counts; test_partim: 7758 / 23023 / 35037 TOAs), which the reference
 * handles by looping over pulsars (red_noise.py:286-298) and a batch-by-equal-order scheme would run as B batches of one.
 * Matrix b: order n[b], row-major lower triangle at A + off[b], leading dimension ld[b]; n, off, ld are HOST arrays, all entries
 * EVEN (16-byte operand rows: pad an odd order with an identity row / column at its end) and ld[b] >= n[b].
 * The matrices are end-aligned: embedded in a virtual matrix whose bottom-right corner they share, so that at every time step
 * (one panel of NB columns counted from the END; NB = 1024, or 2048 when the flop-weighted mean order is >= 16384, or PTA_POTRF_NB) all matrices that have been reached have the same panel boundaries,
 * trailing size and tile grids - every kernel of the step is one launch over them; a matrix enters at the step that contains its
 * first column, with the panel cut at its front (masked inside the kernels).  The diagonal phases (latency chains) are paid once
 * per time step instead of once per matrix and panel.  Chains / look-ahead / workspace scheme as pta_potrf_batched_ws.
 *   1. pta_potrf_ragged_plan(n, off, ld, B, flags, plan_host, &work_doubles) fills plan_host[pta_potrf_ragged_plan_words(B)]
 *      (int64 words; flags: PTA_POTRF_CHAINS / PTA_POTRF_NB / PTA_POTRF_NO_LOOKAHEAD) and returns the workspace size;
 *   2. the caller copies the plan to device memory (plan_dev) - it is reusable for any number of factorisations of this layout;
 *   3. pta_potrf_ragged(A, plan_host, plan_dev, info, work, work_doubles, stream): asynchronous; info[b] (device, caller's order)
 *      = 0 or the 1-based index of the first non-positive pivot of matrix b.  The upper triangles are left holding scratch.   */
int64_t pta_potrf_ragged_plan_words(int B);
int pta_potrf_ragged_plan(const int32_t *n, const int64_t *off, const int64_t *ld, int B, int flags, int64_t *plan_host,
                          int64_t *work_doubles);
int pta_potrf_ragged(double *A, const int64_t *plan_host, const int64_t *plan_dev, int32_t *info, double *work,
                     int64_t work_doubles, void *stream);


/* ---------------------------------------------------------------- GWB -------------- */
/* The reference's chain (red_noise.py:238-287): w -> M@w -> *sqrt(C), zero DC/Nyquist -> Hermitian
 * pack to n = 2Nf-2 -> real(ifft)/dt -> crop [i0, i0+npts) -> linear interpolation onto TOAs.
 * Only npts of the n time samples are ever used, and mixing with M commutes with the DFT, so the
 * device evaluates   G0 = W . T   (pruned inverse DFT as a dense fp64 GEMM on MFMA, T the twiddle
 * matrix with sqrt(C)/(n dt) folded in), then G = M . G0 on the npts grid, then interpolates.  */

/* T[(c*(Nf-2) + k-1)*ldt + jj], c = 0 (cos) / 1 (sin), k = 1..Nf-2, jj < npts:
 *   T0 =  (2/(n dt)) sqrtC[k] cos(2 pi (i0+jj) k / n),   T1 = -(2/(n dt)) sqrtC[k] sin(...)
 * (red_noise.py:265-279, :285 with i0 = 10).                                              */
int pta_gwb_twiddle(const double *sqrtC, int Nf, int npts, int i0, double inv_dt, double *T, int64_t ldt, void *stream);

/* G0[m*ldg + jj] = sum_k wre[m,k] T0[k-1,jj] + wim[m,k] T1[k-1,jj];  w interleaved (re,im),
 * row stride ldw doubles, w[m*ldw + 2k] = Re w[m,k].  algo: 0 = VALU reference kernel,
 * 1 = MFMA kernel.                                                                          */
int pta_gwb_idft(const double *w, int64_t ldw, int M, int Nf, const double *T, int64_t ldt, int npts, double *G0,
                 int64_t ldg, int algo, void *stream);

/* Throughput mode: same transform with w generated on chip - row m = r*P + a uses stream (GWB, a) of
 * realisation r0 + r, pair k -> (Re, Im) of w[a,k] (red_noise.py:238-240) - and with the output window's
 * mirror symmetry exploited:  x[c + j'] = E - O, x[c - j'] = E + O  around the window centre c, which
 * halves the flops.  pta_gwb_twiddle_sym lays the half-window twiddles out slab by slab exactly as the
 * kernel stages them through LDS (Tsym) plus the per-bin rotation e^{2 pi i c k / n} (rot); sizes in doubles
 * come from pta_gwb_twiddle_sym_size.  `variant` selects the column tiling of a workgroup - 0 = 19 tiles of 16 (the
 * whole half window of npts = 600, one wave per SIMD), 1 (recommended) = 10 tiles (two chunks, two waves per SIMD),
 * 2 = 7 tiles (three chunks) - and, since the Tsym layout depends on it, must be the same in all three calls.   */
int64_t pta_gwb_twiddle_sym_size(int Nf, int npts, int variant, int64_t *rot_doubles);
int pta_gwb_twiddle_sym(const double *sqrtC, int Nf, int npts, int i0, double inv_dt, double *Tsym, double *rot,
                        int variant, void *stream);
int pta_gwb_idft_rng(uint64_t seed, uint64_t r0, int R, int P, int Nf, const double *Tsym, const double *rot, int npts,
                     double *G0, int64_t ldg, int variant, int rng_fast, void *stream);

/* The same stage as a chirp-z (Bluestein) transform, the default of the batched engine whenever
 * (Nf-2) + npts - 2 < 4096 (pta_gwb_czt_fits): x_j = (2/(n dt)) Re(W^{j^2/2} sum_k (sqrtC_k w_k W^{k^2/2}) W^{-(j-k)^2/2}),
 * one circular convolution of length 4096 per (realisation, pulsar) row = two in-LDS radix-8 fp64 FFTs,
 * ~0.5 MFLOP per row instead of 3.6 MFLOP of dense DFT.  pta_gwb_czt_setup fills pre[2*4096] (pre-chirp with
 * sqrtC), FB[2*4096] (chirp spectrum / 4096, digit-reversed order, stored [q][b] per 8-point butterfly b), tw[2*4096] (FFT twiddles), post[2*npts].
 * pta_gwb_czt: w == NULL draws on chip (stream (GWB, a), pair k, like pta_gwb_idft_rng), else w[R*P x ldw]
 * interleaved (re, im) rows as in pta_gwb_idft (ldw = 0: one row of draws shared by all rows - timing probe).  */
int pta_gwb_czt_fits(int Nf, int npts, int i0);
/* pta_gwb_czt `variant`: 0 (default): six LDS exchanges, draws / product / output in registers, computed twiddles;
 *                        1: every stage through LDS with table twiddles (cross-check); 10+f: ladder step f              */
int pta_gwb_czt_setup(const double *sqrtC, int Nf, int npts, int i0, double inv_dt, double *pre, double *FB, double *tw,
                      double *post, void *stream);
int pta_gwb_czt(uint64_t seed, uint64_t r0, const double *w, int64_t ldw, int R, int P, int Nf, int npts, int i0,
                const double *pre, const double *FB, const double *tw, const double *post, double *G0, int64_t ldg,
                int variant, int rng_fast, void *stream);

/* G[r,a,:] = sum_b Mchol[a,b] G0[r,b,:]  (the M@w of red_noise.py:268, applied after the DFT). */
/* variant 0 (default): LDS-resident Mchol kernel when P <= 80, generic batched MFMA GEMM otherwise; 1: always the generic
 * MFMA GEMM; 2: the generic VALU reference GEMM (cross-check)                                                       */
int pta_gwb_mix(const double *Mchol, int P, const double *G0, int R, int npts, int64_t ldg, double *G, int variant, void *stream);

/* jlo[i] = last j with ut[j] <= toa_s[i], clamped to [0, npts-2] (numpy.interp's bracket, which
 * scipy.interpolate.interp1d(kind="linear") delegates to; red_noise.py:286-287).           */
int pta_gwb_bracket(const double *ut, int npts, const double *toa_s, int N, int32_t *jlo, void *stream);

/* w[i] = (toa_s[i] - ut[j]) / (ut[j+1] - ut[j]), j = jlo[i]: the realisation-independent half of the linear
 * interpolation red_noise.py:286-287, so the fused kernel evaluates fp[j] + (fp[j+1] - fp[j]) * w per realisation
 * (no division, no grid lookups in the hot loop; differs from interp1d's slope form by rounding only).   */
int pta_gwb_weights(const double *ut, int npts, const double *toa_s, const int32_t *jlo, int N, double *w, void *stream);

/* out[r*ld_out + i] (+)= slope*(toa_s[i] - ut[j]) + G[j],  j = jlo[i], G row = (r, psr_of_toa[i]).
 * scale multiplies the result (1/86400 gives the day-valued delay of red_noise.py:292).     */
int pta_gwb_interp(const double *G, int64_t ldg, int P, int npts, const double *ut, const double *toa_s,
                   const int32_t *psr_of_toa, const int32_t *jlo, int N, int R, double scale, double *out,
                   int64_t ld_out, int accumulate, void *stream);

/* ---------------------------------------------------------------- CGW -------------- */
/* Continuous-wave residual (deterministic.py:97-163). par_host[PTA_CGW_NPAR] holds the scalar
 * prefactors the Python wrapper computes exactly as deterministic.py:51-109 does:
 *  0 tref  1 w0  2 phase0(orbital)  3 w053  4 fac1  5 fac2  6 fac3  7 incfac1  8 incfac2
 *  9 cos2psi  10 sin2psi  11 fplus  12 fcross  13 pd*(1-cosMu)  14 mode (0 evolve, 1 phase_approx,
 *  2 monochromatic)  15 psrTerm (0/1)  16 omega_p (phase_approx, :126)  17 phase0 + fac2*(w053 -
 *  omega_p^(-5/3)) (phase_approx, :130)                                                      */
#define PTA_CGW_NPAR 18
int pta_cgw(const double *mjd, int N, const double *par_host, double *out, int accumulate, void *stream);

/* Catalogue of N_cw continuous-wave sources for one pulsar (add_catalog_of_cws and its numba kernels,
 * deterministic.py:188-561): out[i] (+)= sum_c waveform_c(mjd[i]), NaN terms dropped (:435,:556).
 * par[c*PTA_CW_NPAR + 0..] (device) = the per-source scalars the reference computes at the top of its loop body (:331-383),
 * evaluated on the HOST (pta_replicator_amd.deterministic.cw_source_params) so that they are bit-identical to the reference's:
 *  0 w0  1 phase0(orbital)  2 w0^(-5/3)  3 fac1  4 fac2  5 fac3  6 incfac1  7 incfac2  8 cos2psi  9 sin2psi  10 fplus  11 fcross
 *  12 pd*(1-cosMu) [s]  13 omega_p (phase_approx, :395)  14 phase0 + fac2*(w053 - omega_p^(-5/3)) (phase_approx, :399)  15 unused
 * mode 0 evolve / 1 phase_approx / 2 monochromatic.  partial_ws sized by pta_cw_catalog_workspace (doubles).             */
#define PTA_CW_NPAR 16
int pta_cw_catalog_workspace(int N, int ncw, int64_t *partial_doubles, int *nchunk);
int pta_cw_catalog(const double *mjd, int N, const double *par, int ncw, int psr_term, int mode, double tref,
                   double *partial_ws, double *out, int accumulate, void *stream);

/* ---------------------------------------------------------------- fused engine ----- */
/* One pass that writes R whole-array realisations: out[r, i] = RN + GWB + WN + ECORR + det,
 * every deviate generated on chip (throughput mode).  All arrays are device pointers; any
 * signal whose pointer is NULL is skipped.                                                  */
#define PTA_ENGINE_TILE 256     /* TOAs per workgroup tile; a tile never straddles two pulsars */
#define PTA_ENGINE_EPMAX 132    /* ECORR pairs (= 264 epochs) a tile can stage through LDS per realisation */
typedef struct {
  int32_t n_toa;              /* sum of N_a */
  int32_t n_psr;              /* P */
  int32_t rn_k;               /* 2*components (0 = no red noise) */
  int32_t gw_npts;            /* 0 = no GWB */
  int32_t tnequad;
  int32_t n_tiles;
  const int32_t *tile_psr;    /* [n_tiles] pulsar of the tile */
  const int32_t *tile_start;  /* [n_tiles] index (in the concatenated TOA axis) of the tile's first TOA */
  const int32_t *tile_count;  /* [n_tiles] TOAs in the tile, <= PTA_ENGINE_TILE */
  const int32_t *tile_ep0;    /* [n_tiles] first ECORR pair index (epoch >> 1) the tile touches */
  const int32_t *tile_epn;    /* [n_tiles] number of pairs it touches; 0 = more than PTA_ENGINE_EPMAX (epochs not
                                 contiguous along the TOA axis): deviates are then evaluated per TOA */
  const int32_t *idx_in_psr;  /* [n_toa] index of the TOA inside its pulsar (WN pair index) */
  const double *Ft;           /* [rn_k x ldf] design matrix rows for the concatenated TOAs */
  int64_t ldf;
  const double *rn_coef;      /* [R x n_psr x rn_k] sqrt(prior)*z, from pta_engine_rn_coef */
  const double *gw_G;         /* [R x n_psr x gw_npts] mixed GWB grid series */
  const int32_t *gw_jlo;      /* [n_toa] bracket of the TOA on the coarse grid, from pta_gwb_bracket */
  const double *gw_w;         /* [n_toa] (t - ut[j]) / (ut[j+1] - ut[j]), from pta_gwb_weights */
  const double *wn_a;         /* [n_toa] efac*sigma */
  const double *wn_b;         /* [n_toa] efac*equad (t2equad) or equad (tnequad) */
  const int32_t *epoch_of;    /* [n_toa] epoch index inside the pulsar */
  const double *ecorr_toa;    /* [n_toa] ecorr of the TOA's epoch (0 = none) */
  const double *det;          /* [n_toa] realisation-independent deterministic delay (e.g. CGW) */
  const double *wn_c;         /* opt-in, INSTEAD of wn_a / wn_b (ABI 3): [n_toa] sqrt(wn_a^2 + wn_b^2) - ONE deviate per TOA with the
                                 combined EFAC/EQUAD amplitude (same distribution as white_noise.py:105-109, half the Box-Muller pairs;
                                 not the reference's draw order: TOA idx takes branch (idx >> 4) & 1 of pair idx & ~16 of stream (WN, a)) */
  int32_t rng_fast;           /* Gaussian transform of the on-chip draws (see "RNG" above); 0 = fp64 (default) */
  int32_t synth_variant;      /* fused-kernel variant: 0 (default) = red-noise F @ y on the matrix cores (16 realisations x 256 TOAs
                                 per workgroup), workgroups dealt to the XCDs in contiguous (tile, realisation-group) ranges; 1 = same
                                 kernel in plain linear workgroup order (A/B); 4 / 6 / 8 = all-VALU kernel compiled for that many
                                 waves per SIMD (kept for cross-checks); 100 + k (k <= 64) = the default kernel with k KB of unused
                                 dynamic LDS per workgroup (occupancy probe of round 3: profiles/r03_bench_final.json, DESIGN.md §4.1 "measured and not kept") */
} pta_engine_plan;

/* coef[(r*P + a)*K + c] = amp[a*K + c] * z(seed, r0+r, (RN,a), c)   (red_noise.py:126-127)   */
int pta_engine_rn_coef(uint64_t seed, uint64_t r0, int R, int P, int K, const double *amp, double *coef, int rng_fast, void *stream);

int pta_engine_synth(const pta_engine_plan *plan_host, uint64_t seed, uint64_t r0, int R, double *out, int64_t ld_out,
                     void *stream);

/* One call = R whole-array realisations: pta_engine_rn_coef -> pta_gwb_czt (or pta_gwb_idft_rng) -> pta_gwb_mix ->
 * pta_engine_synth on `stream`, with plan->rn_coef / plan->gw_G taken from the workspace below.  The batched
 * counterpart of one pass of the reference's add_gwb + add_red_noise + add_measurement_noise + add_jitter (+ add_cgw)
 * over the array (red_noise.py:106-298, white_noise.py:47-198).                                            */
typedef struct {
  const double *rn_amp;       /* [n_psr x rn_k] sqrt(prior), red_noise.py:126 (NULL when rn_k == 0) */
  const double *Mchol;        /* [n_psr x n_psr] Cholesky factor of the ORF (NULL when gw_npts == 0) */
  int32_t gw_nf;              /* Nf of the GWB frequency grid */
  int32_t gw_i0;              /* crop offset (10, red_noise.py:285) */
  int32_t use_czt;            /* 1: chirp-z tables below; 0: Tsym / rot of the DFT-GEMM form */
  int32_t czt_variant;        /* `variant` of pta_gwb_czt (0 = default) */
  const double *czt_pre, *czt_FB, *czt_tw, *czt_post; /* from pta_gwb_czt_setup */
  const double *Tsym, *rot;   /* from pta_gwb_twiddle_sym */
  double *ws_coef;            /* [R x n_psr x rn_k] */
  double *ws_G0;              /* [R x n_psr x gw_npts] per-pulsar grid series */
  double *ws_G;               /* [R x n_psr x gw_npts] mixed grid series */
  int32_t idft_variant;       /* `variant` Tsym was built with (pta_gwb_twiddle_sym) */
  int32_t mix_variant;        /* `variant` of pta_gwb_mix (0 = default) */
} pta_engine_tables;

int pta_engine_generate(const pta_engine_plan *plan_host, const pta_engine_tables *tables_host, uint64_t seed, uint64_t r0, int R,
                        double *out, int64_t ld_out, void *stream);

/* ---------------------------------------------------------------- TD mode ---------- */
/* Dense time-domain path named by BASELINE.json's north_star (no counterpart in the reference,
 * which never forms an N_toa x N_toa object: SURVEY.md §0.2, App. A.1).
 * C[i,j] = sum_c phi[c] Ft[c,i] Ft[c,j] + (i==j) sigma2[i] + (epoch_of[i]==epoch_of[j]) ecorr2[i]
 * i.e. the covariance of the reference's RN (red_noise.py:98-101,126-128) + WN/ECORR
 * (white_noise.py:105-109,182) synthesis.  Only the lower triangle (col <= row) is written - that is
 * all pta_potrf_batched reads.  HBM traffic: 8 bytes per element of the triangle, written once; at K = 60 the
 * rank-K product costs 16.6 flop per byte, so the fp64 matrix rate, not HBM, bounds it.      */
int pta_td_cov_assemble(const double *Ft, int64_t ldf, int N, int K, const double *phi, const double *sigma2,
                        const int32_t *epoch_of, const double *ecorr2, double *C, int64_t ldc, void *stream);

/* The same assembly for all pulsars (blocks) of an array in ONE launch of 128 x 128 tiles: block b covers TOAs
 * [blk_off[b], blk_off[b] + blk_n[b]) of the concatenated axis (Ft columns, sigma2, epoch_of, ecorr2 are indexed by it),
 * phi[b*K + c] are its prior variances, and its covariance goes to Cbase + blk_pos[b] with row pitch blk_ld[b] (the layout
 * pta_td_plan describes; all index arrays are device pointers).  max_n = the largest blk_n.                              */
int pta_td_cov_assemble_all(const double *Ft, int64_t ldf, int K, const double *phi, const double *sigma2,
                            const int32_t *epoch_of, const double *ecorr2, double *Cbase, const int64_t *blk_pos,
                            const int32_t *blk_ld, const int32_t *blk_n, const int32_t *blk_off, int n_blocks, int max_n,
                            void *stream);

/* The same assembly by the COLUMN-WALKING kernel (ABI 7; 1 <= K <= 64, 64 ldf < 2^29): a wave keeps the phi-scaled operand of 64
 * columns in registers and walks down the rows 16 at a time; a store instruction writes four rows x 256 bytes (whole cache lines: a
 * lane's accumulators hold two neighbouring columns, 16 bytes per lane); work items = (block, 256-column group, segment of 512 rows),
 * exactly as many as the blocks' orders need - a ragged array launches no empty workgroups.
 * PRECONDITION (as for pta_td_plan): blk_pos[b] and blk_ld[b] must be EVEN - the 16-byte stores assume it; the entry point checks the
 * alignment of Cbase only (the arrays live on the device), engine_td builds both even (orders padded to even, pitches to 16).  pta_td_cov_walk_items writes item0[b] = first item of block b (b <= n_blocks: item0[n_blocks] = the total,
 * also returned; -1 on a bad argument) from the HOST copy of the orders; the caller keeps a device copy of item0 for the launch.
 * Every k index is clamped to K - 1 before it forms an address: nothing behind the [K, ldf] design matrix is ever read.
 * epoch_first (optional, device, indexed like epoch_of): the smallest TOA index INSIDE ITS BLOCK that shares the TOA's epoch - lets a step
 * whose rows have no epoch partner among the wave's columns skip the ECORR arithmetic (every step off the diagonal for time-ordered
 * TOAs); NULL keeps it everywhere.  variant (the SAME K and variant for both calls): 0 or 1 are accepted and select the SAME kernel
 * (reserved for further work-item geometries). */
int64_t pta_td_cov_walk_items(const int32_t *blk_n_host, int n_blocks, int K, int variant, int32_t *item0_host);
int pta_td_cov_assemble_walk(const double *Ft, int64_t ldf, int K, const double *phi, const double *sigma2,
                             const int32_t *epoch_of, const double *ecorr2, double *Cbase, const int64_t *blk_pos,
                             const int32_t *blk_ld, const int32_t *blk_n, const int32_t *blk_off, int n_blocks,
                             const int32_t *item0, int64_t n_items, const int32_t *epoch_first, int variant, void *stream);

/* ASSEMBLY + FACTORISATION of a uniform batch in one call (ABI 8; csrc/pta_td_fused.hip): the covariances are never written.  In the
 * left-looking panel order a tile of the lower triangle is touched once before its panel is factored - by its block column's update
 * C - L[:, <k0] L[:, <k0]^T - so that update COMPUTES C = F diag(phi) F^T + diag(sigma2) + ECORR (red_noise.py:98-101,126-128,
 * white_noise.py:105-109,182) in front of the factor product (four more K slabs of the same MFMA stream) and stores without reading:
 * no assembly launch, no 2 x 8 bytes per element of round trip (6.8 GB written + read at 68 x 5000^2).
 *   Fr [B n, 64]  row-major rows of the design matrix over the batch's concatenated TOAs (matrix b's rows = [b n, (b + 1) n)), columns
 *                 >= kf ZERO;   Gr [B n, 64] = -phi_k Fr[i, k];   kf = red-noise columns in use (0 .. 64; 0 skips the phase)
 *   sigma2 [B n], epoch_of [B n] / ecorr2 [B n] (both or neither NULL): as pta_td_cov_assemble_all
 *   A, n, lda, strideA, B, info, flags, work, work_doubles: as pta_potrf_batched_ws (A's contents are ignored; PTA_POTRF_LEFT is implied;
 *   the workspace scheme must apply: n > panel width and a workspace of pta_potrf_workspace_doubles(n, B, flags) doubles; n, lda, strideA
 *   even).  The factors equal pta_td_cov_assemble_walk + pta_potrf_batched_ws(PTA_POTRF_LEFT) to rounding (the K = 60 product is summed
 *   in another order), not bit for bit. */
int pta_td_assemble_potrf(const double *Fr, const double *Gr, int kf, const double *sigma2, const int32_t *epoch_of, const double *ecorr2,
                          double *A, int n, int64_t lda, int64_t strideA, int B, int32_t *info, int flags, double *work,
                          int64_t work_doubles, void *stream);

/* out[r*ld_out + i] (+)= sum_{j<=i} L[i*ldl + j] z[r*ld_z + j]   (L z, the draw of the dense path;
 * Z . L^T on the fp64 MFMA GEMM).  z holds N(0,1) deviates: NumPy's in replay mode, or
 * pta_rng_fill_normal(stream (TD, pulsar)) in throughput mode.                              */
int pta_td_trmm(const double *L, int64_t ldl, int N, const double *z, int64_t ld_z, int R, double *out, int64_t ld_out,
                int accumulate, int algo, void *stream);   /* algo: 1 = MFMA GEMM, 0 = VALU reference GEMM */

/* Throughput form of the same draw: Z is never materialised - every lane GENERATES its MFMA A operand in registers
 * (stream kind PTA_STREAM_TD / PTA_STREAM_TDGW, deviate j of row m = pair j >> 1, branch j & 1), so one launch covers all
 * pulsars:   out[m, blk_off[b] + i] = sum_{j <= i} L_b[i, j] z(m, b, j)  (+ GWB interpolation + deterministic delay).
 *   rows_per_real == 1 : row m = realisation r0 + m, factor block b = pulsar b, stream (stream_kind, b)   (per-pulsar N_a x N_a factors)
 *   rows_per_real == P : ONE factor block shared by all rows; row m = (realisation r0 + m / P, pulsar m % P), stream
 *                        (stream_kind, m % P)   (the npts x npts factor of the GWB grid covariance)
 * Factor b is row-major at Lbase + blk_pos[b] with leading dimension blk_ld[b]; blk_pos and blk_ld must be EVEN (16-byte
 * double2 loads) and only elements on or below the diagonal are read (pta_potrf_batched_ex may leave the upper triangle
 * unzeroed).  Work items = strips of up to PTA_TD_STRIP consecutive rows of one factor (item_n0, item_rows), sorted by the caller by
 * decreasing min(n, n0 + rows) (their K extent) so that the long strips start first.
 * Optional epilogue (rows_per_real == 1 only): gw_G[(m * n_blocks + b) * gw_npts + j] is the mixed GWB grid series of
 * (realisation m, pulsar b), interpolated with gw_jlo / gw_w exactly as pta_engine_synth does (red_noise.py:286-287);
 * det is added to every row.  All three are indexed by OUTPUT column.                                              */
#define PTA_TD_STRIP 256
typedef struct {
  const double *Lbase;
  const int64_t *blk_pos;     /* [n_blocks] element offset of factor b inside Lbase (even) */
  const int32_t *blk_ld;      /* [n_blocks] leading dimension (even, >= blk_n) */
  const int32_t *blk_n;       /* [n_blocks] order of factor b */
  const int32_t *blk_off;     /* [n_blocks] first output column of block b */
  const int32_t *item_blk;    /* [n_items] factor block of the strip */
  const int32_t *item_n0;     /* [n_items] first row of the strip (a multiple of 16; of PTA_TD_STRIP when item_rows is NULL) */
  int32_t n_blocks;
  int32_t n_items;
  int32_t rows_per_real;
  uint32_t stream_kind;       /* 5 = PTA_STREAM_TD, 6 = PTA_STREAM_TDGW */
  int32_t rng_fast;           /* 0 = fp64 Box-Muller (default), 1 = fp32 transcendentals; per call, not process-wide */
  int32_t gw_npts;
  const double *gw_G;         /* NULL = no GWB epilogue */
  const int32_t *gw_jlo;
  const double *gw_w;
  const double *det;          /* NULL = none */
  const double *z;            /* ABI 3 (ABI 7: also with rows_per_real > 1, row m = output row m): NULL = every deviate is generated in registers (default); else the deviates
                                 are READ: z[m * ld_z + blk_zoff[b] + j] = deviate j of row m for factor block b - what pta_rng_fill_normal
                                 (interleave = 1) writes for stream (stream_kind, b).  Same numbers either way, bit-identical output.
                                 A row must be READABLE up to blk_zoff[b] + blk_n[b] rounded up to a multiple of 4 (ld_z at least that);
                                 what lies behind a block's blk_n[b] deviates is never used, finite or not (ABI 6: the "16 finite doubles
                                 behind the last block" of ABI 5 is gone - the kernel clamps whole groups and zeroes k >= blk_n[b]) */
  int64_t ld_z;
  const int32_t *blk_zoff;    /* [n_blocks] first column of block b's deviates inside a row of z (even: pta_rng_fill_normal writes pairs) */
  const int32_t *item_rows;   /* ABI 8: [n_items] rows of the strip (1 .. PTA_TD_STRIP), or NULL = PTA_TD_STRIP for every strip (cut at the factor's
                                 last row).  Lets the caller put a factor's PARTIAL strip FIRST (rows [0, f), K extent f <= 256) instead of last
                                 (K extent = the factor's order): a strip always costs 16 column tiles per K slab, so the dead tiles of a
                                 partial strip are multiplied by its K - 7 of 16 tiles x K = 5000 per 5000-TOA pulsar = 3.9 % of the launch
                                 when the partial strip is the last one, nothing to speak of when it is the first (round 6).  Strips of one
                                 factor must tile its rows without overlap. */
} pta_td_plan;

int pta_td_trmm_rng(const pta_td_plan *plan_host, uint64_t seed, uint64_t r0, int M, double *out, int64_t ld_out, void *stream);

/* ---------------------------------------------------------------- host-sink replacement ---- */
/* Linearised timing-model fit of a whole ensemble, in place:  rows[r, off_a + i]  <-  r - M_a (M_a^T W_a M_a)^-1 M_a^T W_a r  for every
 * realisation r < R and pulsar a < P - what refitting the timing model after an injection removes from the residuals (the reference goes
 * through PINT once per pulsar: simulate.py:40-69).  Mt[k*ld + i] = column k of the design matrices over the concatenated TOAs (k < m <= 12),
 * Qt[k*ld + i] = row k of (M^T W M)^-1 M^T W (host, realisation independent), psr_off[P + 1] the pulsars' TOA offsets (device int32).
 * One workgroup = one pulsar x 8 realisations; the m x 8 column reductions run as wavefront shuffles + one LDS exchange.          */
int pta_tm_project(const double *Qt, const double *Mt, int64_t ld, int m, const int32_t *psr_off, int P, double *rows, int64_t ld_rows, int R,
                   void *stream);

/* ---------------------------------------------------------------- multi-GPU -------- */
/* The path's one collective (BASELINE.json north_star: "RCCL over xGMI only to all-gather the final residual arrays
 * back to rank 0"): realisations are sharded by contiguous row ranges - rank r owns rows [a_r, b_r) of the [total_rows x n_cols]
 * ensemble, sizes differing by at most one (a_r = r * (total / world) + min(r, total % world)) - and rank `dst` receives every
 * shard directly into its rows of `out` (point-to-point ncclRecv into row slices inside one group: no staging buffers, no
 * concatenation).  `comm` is the caller's ncclComm_t (RCCL); `local` this rank's shard [b_r - a_r, n_cols], `out` (on `dst` only)
 * the [total_rows, n_cols] result.  Asynchronous on `stream`.  world == 1: a device-to-device copy, no communicator needed.  RCCL is
 * bound at run time from the copy already loaded into the process (PyTorch's), so the library has no link-time dependency on it.
 * The reference has no counterpart (single process).
 * Restriction (world > 1): rows must be contiguous on both sides - ld_local == n_cols and, on `dst`, ld_out == n_cols (a send is
 * matched by ONE receive of the same count, so a wider destination would need row-wise sends too); to fill a window of a wider ensemble
 * tensor, gather into a [total_rows, n_cols] tensor and copy.  world == 1 accepts any leading dimensions.  Thread-safe (the RCCL
 * binding is resolved once).                                                                                                    */
int pta_gather_rank0(void *comm, int rank, int world, int dst, const double *local, int64_t total_rows, int64_t n_cols,
                     int64_t ld_local, double *out, int64_t ld_out, void *stream);

/* ---------------------------------------------------------------- fp64 GEMM -------- */
/* C[b] = alpha * A[b] * op(B[b]) + beta * C[b], row-major, batch `batch` with element strides.
 * A element (m,k) = A[m*lda + k*ska] (ska = 2 reads the real or imaginary plane of interleaved
 * complex rows); transB = 0: B is [K x N]; 1: B is [N x K].  lower_only = 1 touches only col <= row
 * (SYRK).  algo 0 = VALU reference kernel, 1 = v_mfma_f64_16x16x4_f64 kernel, 2 (3: with the C-tile prefetch epilogue, A/B) = the MFMA kernel whose 128 x 128-tile form stages
 * its operand slabs by LDS DMA (transB = 1, ska = 1, even K / lda / ldb / strides, 16-byte aligned A and B; else as algo 1). */
int pta_dgemm(int transB, int M, int N, int K, double alpha, const double *A, int64_t lda, int64_t ska,
              const double *B, int64_t ldb, double beta, double *C, int64_t ldc, int lower_only, int batch,
              int64_t strideA, int64_t strideB, int64_t strideC, int algo, void *stream);

/* ---------------------------------------------------------------- microbenchmarks -- */
/* Measure the roofline denominators on the device the library runs on (bench.py):
 * kind 0: fp64 MFMA (v_mfma_f64_16x16x4_f64) TFLOP/s, 1: fp64 FMA TFLOP/s,
 * 2: HBM write GB/s over `bytes`, 3: HBM copy GB/s, 4: Philox+Box-Muller G normals/s, 5: fp64 MFMA in the GEMM kernels'
 * register-tile pattern (4 A x 4 B fragments -> 16 accumulators), 6: that MFMA loop
 * and the FMA loop on alternating waves of the same SIMDs (sum of both rates: do they share the fp64 ALUs?).  kinds 0 / 1 / 4 / 5: `bytes` in 1..32 = 256-thread blocks per CU.       */
int pta_microbench(int kind, int64_t bytes, int iters, int option, double *result_host);   /* option of kind 4: 0 = default fp64 transform, 1 = rng_fast, 2 = the polynomial fp64 transform (A/B) */

/* Engine-clock probe (ABI 6): one wave on `stream` samples (s_memrealtime [100 MHz ticks], s_memtime [shader cycles]) into
 * samples[2 i], samples[2 i + 1] every `period_us` for `us` microseconds (at most max_samples pairs; unused slots zeroed).  Launch it
 * on a side stream, run the kernels of interest on another, and the slope cycles / (10 ns) between samples is the engine clock in GHz
 * while they ran (bench.py: roofline.engine_clock_GHz; scripts/gpu_r4_clocks.py).  Asynchronous.                                   */
int pta_clock_probe(uint64_t *samples, int max_samples, int us, int period_us, void *stream);

/* self-test of the fp64 MFMA lane layout used by the GEMM kernels: returns 0 when a 16x16x4
 * product with asymmetric operands matches the scalar result on the device.                */
int pta_selftest_mfma_f64(double *max_err_host);

#ifdef __cplusplus
}
#endif
#endif /* PTA_REPLICATOR_AMD_H */
