"""TD ("time-domain") mode of the ReplicaEngine: the dense path BASELINE.json's north_star names.

    per-pulsar covariance assembly  ->  batched blocked Cholesky (fp64 MFMA trailing update)  ->  L . z against Gaussian
    deviates drawn on chip  ->  + GWB  ->  residuals

The reference never forms an N_toa x N_toa object; what is factored here is the covariance its synthesis
implies:

    C_a = F diag(phi) F^T                                   red_noise.py:98-101,126-128 (rank 2 * components)
        + diag((efac sigma)^2 + (efac equad | equad)^2)     white_noise.py:105-109
        + sum_e ecorr_e^2 1_e 1_e^T                         white_noise.py:182
    GWB on the npts-sample grid: Sigma_g = T^T T, T the twiddle matrix of the pruned inverse DFT (sqrt(C(f)) folded in,
        red_noise.py:265-285) - the Toeplitz matrix 4/(dt^2 n^2) sum_k C_k cos(2 pi k (p - q) / n) of App. A.1 - times the
        ORF between pulsars (red_noise.py:224-235), then the same linear interpolation onto the TOAs (:286-287).

``prepare_td()`` assembles and factors once (one L_a per pulsar, one L_g for the grid); ``generate_td(R)`` then costs one
triangular product per pulsar per realisation, with every deviate generated in registers as the MFMA A operand
(``pta_td_trmm_rng``: stream (TD, a) for the N_a deviates of pulsar a, (TDGW, a) for its npts grid deviates).  A TD realisation
has the same distribution as a throughput-mode one but is not draw-for-draw comparable with the reference (different
deviates); its CPU oracle is ``oracle/pta_oracle.py: td_*`` on the dumped deviates (``dump_draws_td``).

Conditioning.  Sigma_g spans the spectrum's full dynamic range over the grid (condition number ~3e14 for gamma = 13/3), so
two correct Cholesky factorisations agree only to ~cond * eps in the factor; what is pinned for it is the backward error
|| L L^T - Sigma || / || Sigma || (tests) - the realisations are exact draws of N(0, L L^T).  If the device factorisation
meets a non-positive pivot, a relative diagonal jitter (1e-14 of the mean diagonal, x10 per retry) is added and recorded in
``gw_td_jitter``.
"""
import ctypes

import numpy as np
import torch

from . import _lib, device as dv

STREAM_TD, STREAM_TDGW = 5, 6


def _strips(counts, strip=_lib.TD_STRIP):
    """work items (block, first row, rows) of the triangular product, longest K extent first.  A factor whose order is not a multiple of
    the strip height gets its PARTIAL strip first - rows [0, f), f = (n mod strip) rounded up to 16 - and whole strips behind it: a strip
    costs 16 column tiles per K slab whatever its height, so a partial strip should be the one with the SHORTEST K extent (f columns),
    not the longest (n columns: 7 of 16 tiles x K = 5000 of a 5000-TOA pulsar, 3.9 % of the launch's matrix-pipe work)."""
    items = []
    for b, n in enumerate(counts):
        n = int(n)
        f = n % strip
        first = strip if f == 0 else min(strip, (f + 15) // 16 * 16)
        n0 = 0
        while n0 < n:
            rows = min(first if n0 == 0 else strip, n - n0)
            items.append((min(n, n0 + rows), b, n0, rows))
            n0 += rows
    items.sort(key=lambda x: -x[0])
    return (np.array([x[1] for x in items], dtype=np.int32), np.array([x[2] for x in items], dtype=np.int32),
            np.array([x[3] for x in items], dtype=np.int32))


class TimeDomainMixin:
    # ---------------------------------------------------------------- prepare -------------------
    def prepare_td(self, lookahead=True):
        """assemble and factor the dense covariances (once per array and noise model)."""
        if not self._prepared:
            self.prepare()
        if self._wn is None:
            raise ValueError("TD mode needs measurement noise (set_white_noise): without it the dense covariance is singular")
        P = self.P
        pl = self.plan
        counts = [int(c) for c in self.counts]
        # storage order: an odd TOA count gets one identity row / column at its END (diag(C, 1) factors to diag(L, 1); the product
        # kernel never reads it) so that every panel boundary of the end-aligned ragged factorisation falls on a 16-byte boundary
        nst = [n + (n & 1) for n in counts]
        ld = [(n + 15) // 16 * 16 for n in nst]           # rows start on 128-byte lines (and even: the product kernel loads double2)
        pos = np.concatenate([[0], np.cumsum([n * l for n, l in zip(nst, ld)])]).astype(np.int64)
        self.td_ld, self.td_pos, self.td_nst = ld, pos, nst
        cur = getattr(self, "d_Ltd", None)
        if cur is None or cur.numel() != int(pos[-1]):     # a buffer of the right size is factored in again (first-touch paid once)
            self.d_Ltd = None                              # release a previous factor buffer first: two do not fit at SKA scale
            self.d_Ltd = dv.empty((int(pos[-1]),))
        sigma2 = self.d_wn_a ** 2 + self.d_wn_b ** 2      # (efac sigma)^2 + (efac equad | equad)^2
        self._td_sigma2 = sigma2
        # block layout arrays (also what the product kernel reads)
        self._td_layout = [dv.i64(pos[:-1]), dv.i32(ld), dv.i32(counts), dv.i32(self.off[:-1])]
        self._td_pad_idx = None
        # everything below is stream-ordered on the caller's stream: operands computed by torch, the assembly launch, the factorisation's
        # internal chains (joined to the caller's stream through events).  Round 4 kept two host synchronisations here as insurance against
        # an unexplained NaN observation; its cause was an out-of-range operand read of the withdrawn assembly kernel (DESIGN.md §4.2,
        # csrc/pta_td_kernels.hip), not an ordering hazard, and tests/test_gpu_td.py::test_td_prepare_without_host_sync_* hold the
        # asynchronous sequence bit-equal to a fully serialised one.
        if self._td_can_fuse(lookahead):
            self.td_assemble_factorise()                  # uniform array: the assembly fused into the left-looking factorisation's updates
        else:
            self.td_assemble()
            self.td_factorise(lookahead=lookahead)
        blk, n0, rows = _strips(counts)
        self._td_keep = self._td_layout + [dv.i32(blk), dv.i32(n0), dv.i32(rows)]
        tp = _lib.TdPlan()
        tp.Lbase = self.d_Ltd.data_ptr()
        tp.blk_pos, tp.blk_ld, tp.blk_n, tp.blk_off, tp.item_blk, tp.item_n0, tp.item_rows = [x.data_ptr() for x in self._td_keep]
        tp.n_blocks, tp.n_items, tp.rows_per_real, tp.stream_kind, tp.rng_fast = P, len(blk), 1, STREAM_TD, 0
        tp.det = pl.det
        self.td_plan = tp

        # ---- GWB: factor of the grid covariance Sigma_g = T^T T (realisation independent, shared by all pulsars)
        if pl.gw_npts:
            self._prepare_gw_grid_factor()
            tp.gw_npts, tp.gw_jlo, tp.gw_w = pl.gw_npts, self.d_jlo.data_ptr(), self.d_gw_w.data_ptr()
        else:
            self.tdgw_plan = None
            self.gw_td_jitter = 0.0
        self._td_bufs = None
        self._td_prepared = True
        return self

    def _td_can_fuse(self, lookahead=True):
        """the fused assembly + factorisation (pta_td_assemble_potrf; OPT-IN: attribute td_fused = True) applies to what the headline array
        is: every pulsar the same EVEN TOA count above one panel, <= 64 red-noise columns, the default left-looking order with its workspace,
        no A/B flags.  Not the default because it is a wash: at 68 x 5000^2 its update launches take 33.25 ms against 31.22 + 1.98 for the
        plain updates + k_td_cov_walk (one chain, rocprofv3: scripts/gpu_r6_fused_trace.py), prepare_td() 54.2 against 54.7 ms - the K = 60
        product it moves into the updates is 109 GFLOP that the walking kernel already runs beside its HBM writes, and the C read it
        saves was hidden behind the products."""
        nst = self.td_nst
        return (bool(getattr(self, "td_fused", False)) and lookahead and len(set(nst)) == 1 and all(int(c) % 2 == 0 for c in self.counts)
                and nst[0] > 1152 and nst[0] < 16384 and self.plan.rn_k <= 64 and getattr(self, "td_potrf_mode", "auto") in ("auto", "uniform")
                and getattr(self, "td_potrf_order", "left") == "left" and bool(getattr(self, "td_potrf_workspace", True))
                and int(getattr(self, "td_potrf_flags", 0)) == 0 and getattr(self, "td_cov_kernel", "auto") == "auto")

    def td_assemble_factorise(self):
        """covariance assembly AND batched factorisation of a uniform array as ONE call: the covariances are never written - every block
        column of the left-looking factorisation is computed by its update, F diag(phi) F^T + diag(sigma^2) + ECORR - L L^T, from row-major
        copies of the design matrix (Fr, and Gr = -phi Fr: 2 x 64 columns x 8 bytes per TOA, built once per noise model).  Saves the
        assembly launch and the 2 x 8 bytes per element of round trip; the factors agree with the two-step path to rounding."""
        P, N, K, s = self.P, self.n_toa, self.plan.rn_k, dv.stream_ptr()
        key = (self.d_Ft.data_ptr() if K else 0, self.d_amp.data_ptr() if K else 0)
        op = getattr(self, "_td_fuse_ops", None)
        if op is None or op[0] != key:
            if K:
                Fr = dv.zeros((N, 64))
                Fr[:, :K] = self.d_Ft.t()
                phi = (self.d_amp ** 2)[self.d_psr_of.long()]                 # [N, K]: each TOA's pulsar's prior variances
                Gr = dv.zeros((N, 64))
                Gr[:, :K] = -(phi * Fr[:, :K])
            else:
                Fr = Gr = dv.zeros((64,))                                     # kf = 0: never read
            op = self._td_fuse_ops = (key, Fr, Gr)
        ecorr2 = (self.d_ecorr_toa ** 2).contiguous() if self.plan.ecorr_toa else None
        n, ld = self.td_nst[0], self.td_ld[0]
        info = dv.zeros((P,), dtype=torch.int32)
        flags = _lib.POTRF_LEFT
        need = int(_lib.lib.pta_potrf_workspace_doubles(n, P, flags))
        work = dv.empty((need,))
        _lib.call("pta_td_assemble_potrf", dv.ptr(op[1]), dv.ptr(op[2]), K, dv.ptr(self._td_sigma2),
                  dv.ptr(self.d_epoch_of) if ecorr2 is not None else None, dv.ptr(ecorr2) if ecorr2 is not None else None,
                  dv.ptr(self.d_Ltd), n, ld, n * ld, P, dv.ptr(info), flags, dv.ptr(work), need, s)
        del work
        self.td_cov_kernel_used, self.td_potrf_mode_used = "fused", "uniform"
        bad = info.cpu().numpy()
        if np.any(bad != 0):
            a = int(np.nonzero(bad)[0][0])
            raise np.linalg.LinAlgError(f"TD covariance of {self.names[a]} is not positive definite (leading minor {int(bad[a])})")

    def td_assemble(self, kernel=None):
        """ONE assembly launch over all pulsars: lower triangles of C_a into the factor buffer (+ the identity tail of odd orders).
        kernel (default: attribute td_cov_kernel, "auto"): "walk" = the column-walking kernel (pta_td_cov_assemble_walk: 1 <= K <= 64,
        work items sized by every pulsar's own order), "tile" = the 64 x 128-tile kernel (pta_td_cov_assemble_all: any K, grid sized by
        the largest pulsar), "auto" = walk where it applies."""
        pl, P, N, s = self.plan, self.P, self.n_toa, dv.stream_ptr()
        counts = [int(c) for c in self.counts]
        K = pl.rn_k
        phi = (self.d_amp ** 2).contiguous() if K else None
        ecorr2 = (self.d_ecorr_toa ** 2).contiguous() if pl.ecorr_toa else None
        kernel = kernel or getattr(self, "td_cov_kernel", "auto")
        walk_ok = 1 <= K <= 64 and 64 * N < (1 << 29)
        if kernel == "auto":
            kernel = "walk" if walk_ok else "tile"
        if kernel == "walk":
            if not walk_ok:
                raise ValueError(f"td_cov_kernel='walk' needs 1 <= K <= 64 red-noise columns and 64 n_toa < 2^29 (K={K}, n_toa={N})")
            variant = int(getattr(self, "td_cov_walk_variant", 0))     # pta_td_cov_assemble_walk: 0 (default) or 1, one kernel either way (reserved)
            items = getattr(self, "_td_walk_items", None)
            if items is None or items[0] != (tuple(counts), K, variant):
                h_n = np.ascontiguousarray(counts, dtype=np.int32)
                item0 = np.zeros(P + 1, dtype=np.int32)
                total = int(_lib.lib.pta_td_cov_walk_items(dv.hptr(h_n), P, K, variant, dv.hptr(item0)))
                if total <= 0:
                    raise _lib.PtaError("pta_td_cov_walk_items failed")
                items = self._td_walk_items = ((tuple(counts), K, variant), dv.i32(item0), total)
            efirst = None
            if ecorr2 is not None:   # per TOA: the first TOA (index inside the pulsar) of its ECORR epoch - where its epoch partners can begin
                efirst = getattr(self, "_td_epoch_first", None)
                if efirst is None or efirst[0] is not self.d_epoch_of:
                    lo = np.empty(N, dtype=np.int32)
                    for a in range(P):
                        ep = np.asarray(self.epoch_of[a], dtype=np.int64)
                        first = np.full(int(ep.max()) + 1 if len(ep) else 1, len(ep), dtype=np.int64)
                        np.minimum.at(first, ep, np.arange(len(ep)))
                        lo[self.off[a]:self.off[a + 1]] = first[ep]
                    efirst = self._td_epoch_first = (self.d_epoch_of, dv.i32(lo))
            _lib.call("pta_td_cov_assemble_walk", dv.ptr(self.d_Ft), N, K, dv.ptr(phi), dv.ptr(self._td_sigma2),
                      dv.ptr(self.d_epoch_of) if ecorr2 is not None else None, dv.ptr(ecorr2) if ecorr2 is not None else None,
                      dv.ptr(self.d_Ltd), *[dv.ptr(x) for x in self._td_layout], P, dv.ptr(items[1]), items[2],
                      dv.ptr(efirst[1]) if efirst is not None else None, variant, s)
        elif kernel == "tile":
            _lib.call("pta_td_cov_assemble_all", dv.ptr(self.d_Ft) if K else None, N, K, dv.ptr(phi) if K else None, dv.ptr(self._td_sigma2),
                      dv.ptr(self.d_epoch_of) if ecorr2 is not None else None, dv.ptr(ecorr2) if ecorr2 is not None else None,
                      dv.ptr(self.d_Ltd), *[dv.ptr(x) for x in self._td_layout], P, max(counts), s)
        else:
            raise ValueError(f"td_cov_kernel={kernel!r}: 'auto', 'walk' or 'tile'")
        self.td_cov_kernel_used = kernel
        pad = getattr(self, "_td_pad_idx", None)
        if pad is None:   # the identity row of every odd order: positions of its zeros and of its one, built once per layout
            rows, ones = [], []
            for a, n in enumerate(counts):
                if n & 1:
                    start = int(self.td_pos[a]) + n * self.td_ld[a]
                    rows.append(np.arange(start, start + n, dtype=np.int64))
                    ones.append(start + n)
            pad = self._td_pad_idx = (dv.i64(np.concatenate(rows)), dv.i64(ones)) if ones else ()
        if pad:
            self.d_Ltd.index_fill_(0, pad[0], 0.0)
            self.d_Ltd.index_fill_(0, pad[1], 1.0)

    def td_factorise(self, lookahead=True, mode=None):
        """the batched factorisation of the assembled covariances, in place.  mode (default: attribute td_potrf_mode, "auto"):
        "uniform" = runs of consecutive pulsars with the same TOA count share one pta_potrf_batched_ws launch sequence (a real array
        degenerates to batches of one); "ragged" = ALL pulsars as one end-aligned schedule (pta_potrf_ragged); "auto" = uniform when
        every pulsar has the same count, ragged otherwise."""
        P, s = self.P, dv.stream_ptr()
        nst, ld, pos = self.td_nst, self.td_ld, self.td_pos
        mode = mode or getattr(self, "td_potrf_mode", "auto")
        if mode == "auto":
            mode = "uniform" if len(set(nst)) == 1 else "ragged"
        info = dv.zeros((P,), dtype=torch.int32)
        extra = int(getattr(self, "td_potrf_flags", 0))
        if mode == "ragged":
            # the ragged schedule always runs the workspace scheme (P x NBO^2 doubles for the factorisation only: 8.4 MB per pulsar at
            # 1024-column panels, 33.5 MB at the 2048 columns the plan takes for arrays of large matrices) with its own look-ahead:
            # options that only the uniform path honours are refused instead of being ignored
            if not bool(getattr(self, "td_potrf_workspace", True)):
                raise ValueError("td_potrf_workspace = False is a uniform-batch option: the ragged schedule (pulsars of different TOA counts) needs its "
                                 "workspace; use td_potrf_mode = 'uniform' for the workspace-free per-matrix schedule")
            bad = extra & (_lib.POTRF_DIAG_AHEAD | _lib.POTRF_LOCKSTEP)
            if bad:
                raise ValueError(f"td_potrf_flags 0x{bad:x} (PTA_POTRF_DIAG_AHEAD / PTA_POTRF_LOCKSTEP) are uniform-batch options: the ragged schedule "
                                 "has its own look-ahead (PTA_POTRF_NO_LOOKAHEAD turns it off)")
            flags = (0 if lookahead else _lib.POTRF_NO_LOOKAHEAD) | extra
            key = (tuple(nst), flags)
            if getattr(self, "_td_rag_key", None) != key:
                words = int(_lib.lib.pta_potrf_ragged_plan_words(P))
                plan = np.zeros(words, dtype=np.int64)
                need = ctypes.c_int64(0)
                h_n, h_off, h_ld = np.ascontiguousarray(nst, dtype=np.int32), np.ascontiguousarray(pos[:-1], dtype=np.int64), np.ascontiguousarray(ld, dtype=np.int64)
                _lib.call("pta_potrf_ragged_plan", dv.hptr(h_n), dv.hptr(h_off), dv.hptr(h_ld), P, flags, dv.hptr(plan), ctypes.byref(need))
                self._td_rag_plan, self._td_rag_plan_dev, self._td_rag_need, self._td_rag_key = plan, dv.i64(plan), int(need.value), key
            work = dv.empty((self._td_rag_need,))
            _lib.call("pta_potrf_ragged", dv.ptr(self.d_Ltd), dv.hptr(self._td_rag_plan), dv.ptr(self._td_rag_plan_dev), dv.ptr(info), dv.ptr(work),
                      self._td_rag_need, s)
            del work                                    # stream-ordered: the caching allocator reuses it only behind these kernels
        else:
            # workspace scheme of the factorisation (include/pta_replicator_amd.h: pta_potrf_batched_ws): strips [-W_jj L11[j, <j] | W_jj] of
            # the panels' diagonal blocks, 10.6 MB per matrix at the default panel width, released right after the factorisation - and
            # the next panel's diagonal phase run ahead on a side stream (PTA_POTRF_DIAG_AHEAD).  68 x 5000^2: 53.2 ms against 56.5 ms
            # without; td_potrf_workspace = False keeps the workspace-free two-chain schedule.
            # panel order (round 6): "left" (default) = left-looking - a finished panel is not applied to the trailing matrix; each block
            # column is updated once, right before it is factored, by ONE product over all columns to its left (pta_potrf_batched_ws with
            # PTA_POTRF_LEFT: 51.2-51.3 ms = 0.70 at 68 x 5000^2 against 52.3-52.6 = 0.69 for "right", the right-looking schedule with the
            # next panel's diagonal phase run ahead, PTA_POTRF_DIAG_AHEAD; 16 x 10 000^2: 93.0 against 94.1; profiles/r06_potrf_left_looking.txt)
            use_ws = bool(getattr(self, "td_potrf_workspace", True))
            order = getattr(self, "td_potrf_order", "left")
            if order not in ("left", "right"):
                raise ValueError(f"td_potrf_order={order!r}: 'left' or 'right'")
            ahead = _lib.POTRF_LEFT if (order == "left" and not extra & (_lib.POTRF_DIAG_AHEAD | _lib.POTRF_LEFT)) else _lib.POTRF_DIAG_AHEAD
            flags0 = (ahead if use_ws else 0) if lookahead else _lib.POTRF_NO_LOOKAHEAD
            flags0 |= extra
            a = 0
            while a < P:
                b = a
                while b + 1 < P and nst[b + 1] == nst[a]:
                    b += 1
                # panels of 2048 columns for large matrices (as the ragged schedule's default): half as many passes over the trailing matrix
                flags = flags0 | (_lib.POTRF_NB(8) if (use_ws and nst[a] >= 16384 and not (flags0 >> 8) & 0xFF) else 0)
                need = int(_lib.lib.pta_potrf_workspace_doubles(nst[a], b - a + 1, flags)) if use_ws else 0
                work = dv.empty((need,)) if need else None
                _lib.call("pta_potrf_batched_ws", ctypes.c_void_p(self.d_Ltd.data_ptr() + 8 * int(pos[a])), nst[a], ld[a],
                          nst[a] * ld[a], b - a + 1, ctypes.c_void_p(info.data_ptr() + 4 * a), flags, dv.ptr(work), need, s)
                del work
                a = b + 1
        self.td_potrf_mode_used = mode
        bad = info.cpu().numpy()
        if np.any(bad != 0):
            a = int(np.nonzero(bad)[0][0])
            raise np.linalg.LinAlgError(f"TD covariance of {self.names[a]} is not positive definite (leading minor {int(bad[a])})")

    def _prepare_gw_grid_factor(self):
        """Cholesky factor L_g of the covariance of the npts GWB grid samples of one pulsar, Sigma_g = T^T T (T the twiddle
        matrix of the pruned inverse DFT with sqrt(C(f)) folded in, red_noise.py:265-285; equals the Toeplitz matrix of
        SURVEY.md App. A.1) and the pta_td_plan that draws through it (rows = (realisation, pulsar), stream (TDGW, pulsar))."""
        if getattr(self, "_gw_grid_ready", False):
            return
        if not self._prepared:
            self.prepare()
        pl, s, P = self.plan, dv.stream_ptr(), self.P
        npts, Nf = pl.gw_npts, self.grid["Nf"]
        ldg = (npts + 1) // 2 * 2
        Sg = dv.zeros((npts, ldg))
        _lib.call("pta_dgemm", 0, npts, npts, 2 * (Nf - 2), ctypes.c_double(1.0), dv.ptr(self.d_T), 1, self.ldt, dv.ptr(self.d_T),
                  self.ldt, ctypes.c_double(0.0), dv.ptr(Sg), ldg, 1, 1, 0, 0, 0, 1, s)
        self.Sg_td = Sg
        ginfo = dv.zeros((1,), dtype=torch.int32)
        mean_diag = float(torch.diagonal(Sg[:, :npts]).mean().item())
        eps = 0.0
        while True:
            Lg = Sg.clone()
            if eps:
                torch.diagonal(Lg[:, :npts]).add_(eps * mean_diag)
            # forward-substitution panel solves: LAPACK-grade backward error on this cond ~ 3e14 matrix
            _lib.call("pta_potrf_batched_ex", dv.ptr(Lg), npts, ldg, npts * ldg, 1, dv.ptr(ginfo), _lib.POTRF_SUBSTITUTION, s)
            if int(ginfo.item()) == 0:
                break
            eps = 1e-14 if eps == 0.0 else eps * 10.0
            if eps > 1e-8:
                raise np.linalg.LinAlgError("GWB grid covariance is not positive definite even with 1e-8 relative jitter")
        self.gw_td_jitter = eps
        self.d_Lg, self.td_ldg = Lg, ldg
        gblk, gn0, grows = _strips([npts])
        self._tdgw_keep = [dv.i64([0]), dv.i32([ldg]), dv.i32([npts]), dv.i32([0]), dv.i32(gblk), dv.i32(gn0), dv.i32(grows)]
        # "memory" draws of the grid stage: one row of npts (rounded up to even) deviates per (realisation, pulsar), written by
        # pta_rng_fill_normal_blocks as P blocks of a realisation's row - stream (TDGW, pulsar), the numbers the register form draws
        self._tdgw_zld = (npts + 1) // 2 * 2
        self._tdgw_fill = [dv.i32([npts] * P), dv.i32(np.arange(P) * self._tdgw_zld), dv.i32([0])]
        gp = _lib.TdPlan()
        gp.Lbase = Lg.data_ptr()
        gp.blk_pos, gp.blk_ld, gp.blk_n, gp.blk_off, gp.item_blk, gp.item_n0, gp.item_rows = [x.data_ptr() for x in self._tdgw_keep]
        gp.n_blocks, gp.n_items, gp.rows_per_real, gp.stream_kind, gp.rng_fast = 1, len(gblk), P, STREAM_TDGW, 0
        self.tdgw_plan = gp
        self._gw_grid_ready = True

    # ---------------------------------------------------------------- generate ------------------
    def generate_td(self, R, r0=0, out=None, chunk=4096):
        """out[R, n_toa] (device, seconds): realisations r0 .. r0+R-1 of the dense path, deviates drawn on chip.

        The batch runs in chunks.  ``td_overlap = True`` (opt-in, "memory" draws only) prepares chunk c + 1 - its deviates
        (pta_rng_fill_normal_blocks) and its GWB grid series (the 600 x 600 factor's product + the ORF mix) - on a side stream while the
        triangular product of chunk c occupies the matrix cores: the same kernels on the same counters, bit-identical output.  Measured
        and NOT the default (profiles/r04_bench_mid_round.json): 32.8 ms per 1024 realisations of the 68 x 5000 array in four pipelined
        chunks of 256 against 31.2 ms as one chunk - every chunk streams the 13.6 GB of factors again, and the preparation kernels are
        VALU work on the ALUs the matrix pipe shares, not idle time to fill."""
        if not getattr(self, "_td_prepared", False) or not self._prepared:
            self.prepare_td()
        if out is None:
            out = dv.empty((R, self.n_toa))
        P, npts = self.P, self.plan.gw_npts
        tp = self.td_plan
        tp.rng_fast = int(self.rng_fast)
        if self.tdgw_plan is not None:
            self.tdgw_plan.rng_fast = int(self.rng_fast)
        # "memory" (default): the deviates of a batch are written once (pta_rng_fill_normal, 8 bytes each: 2.8 GB per 1024 realisations of
        # the 68 x 5000 array) and READ by the product; "registers": generated inside the product's loop (no buffer) - the same numbers,
        # bit-identical realisations; 31.2 against 35.0 ms per 1024 (the fp64 Box-Muller shares the double-precision ALUs with the MFMAs)
        zmem = getattr(self, "td_draws", "memory") == "memory"
        overlap = zmem and bool(getattr(self, "td_overlap", False)) and R > int(getattr(self, "td_chunk", 256))
        chunk = int(min(chunk, R))
        if zmem:
            chunk = int(min(chunk, 1024))
        if overlap:
            chunk = int(min(chunk, getattr(self, "td_chunk", 256)))
        nbuf = 2 if overlap else 1
        bufs = getattr(self, "_td_bufs", None)
        zcols = int(np.sum((self.counts + 1) // 2 * 2)) + 16
        if bufs is None or len(bufs) < nbuf or bufs[0]["chunk"] < chunk or (zmem and bufs[0]["z"] is None):
            bufs = []
            for _ in range(nbuf):
                b = {"chunk": chunk, "z": None, "zg": None, "G0": None, "G": None}
                if npts:
                    b["G0"], b["G"] = dv.empty((chunk, P, npts)), dv.empty((chunk, P, npts))
                if zmem:
                    b["z"] = dv.zeros((chunk, zcols))
                    if npts:   # the grid stage's deviates: [chunk * P, npts (even)] (+ a pad row for the product's 32-byte reads)
                        b["zg"] = dv.zeros((chunk * P + 1, self._tdgw_zld))
                bufs.append(b)
            self._td_bufs = bufs
            zoff = np.concatenate([[0], np.cumsum((self.counts + 1) // 2 * 2)]).astype(np.int64)   # every block starts on an even column
            self._td_zoff = dv.i32(zoff[:-1])
        if not zmem:
            tp.z, tp.ld_z, tp.blk_zoff = None, 0, None
        main = torch.cuda.current_stream()
        s = ctypes.c_void_p(main.cuda_stream)
        if overlap:
            side = getattr(self, "_td_side", None)
            if side is None:
                side = self._td_side = torch.cuda.Stream()
                self._td_ev = [[torch.cuda.Event() for _ in range(2)] for _ in range(2)]
            ev_prep, ev_done = self._td_ev
            side.wait_stream(main)
        los = list(range(0, R, chunk))

        # the deviate fill (VALU + stores) and the GWB grid stage (a small MFMA product + the ORF mix) of a chunk are independent: the fill
        # goes to a second stream and is joined in front of the product (td_fill_beside_gwb, default on; same kernels, same counters,
        # bit-identical).  Round 4 measured nothing to gain (31.26 against 31.27 ms: the grid stage then drew in registers - VALU work like
        # the fill); with the grid deviates read from memory it is MFMA work beside a VALU + store kernel: 29.95 against 30.05-30.2 ms
        beside = zmem and bool(npts) and not overlap and bool(getattr(self, "td_fill_beside_gwb", True))
        if beside:
            fstream = getattr(self, "_td_fill_stream", None)
            if fstream is None:
                fstream = self._td_fill_stream = torch.cuda.Stream()
                self._td_fill_ev = torch.cuda.Event()

        def fill_chunk(b, lo, n, sp):   # every pulsar's deviates of this chunk in one launch (stream (STREAM_TD, pulsar), as the register form draws them)
            _lib.call("pta_rng_fill_normal_blocks", self.seed, r0 + lo, n, STREAM_TD, P, dv.ptr(self._td_layout[2]), dv.ptr(self._td_zoff),
                      int(max(self.counts)), dv.ptr(b["z"]), b["z"].stride(0), int(self.rng_fast), sp)

        def prepare_chunk(c, sp):
            """deviates + GWB grid series of chunk c into buffer c & 1, on stream pointer sp"""
            b, lo = bufs[c % nbuf], los[c]
            n = min(chunk, R - lo)
            if beside:
                fstream.wait_stream(main)                    # behind the product that last read this Z buffer
                with torch.cuda.stream(fstream):
                    fill_chunk(b, lo, n, ctypes.c_void_p(fstream.cuda_stream))
                    self._td_fill_ev.record(fstream)
            if npts:
                gp = self.tdgw_plan
                if zmem:   # grid deviates written once and READ by the product (as the per-pulsar part): the fp64 Box-Muller of the register
                    # form shares the double-precision ALUs with the MFMAs (1.3 -> 0.8 ms per 1024 x 68 rows; bit-identical)
                    _lib.call("pta_rng_fill_normal_blocks", self.seed, r0 + lo, n, STREAM_TDGW, P, dv.ptr(self._tdgw_fill[0]), dv.ptr(self._tdgw_fill[1]),
                              npts, dv.ptr(b["zg"]), P * self._tdgw_zld, int(self.rng_fast), sp)
                    gp.z, gp.ld_z, gp.blk_zoff = b["zg"].data_ptr(), self._tdgw_zld, self._tdgw_fill[2].data_ptr()
                else:
                    gp.z, gp.ld_z, gp.blk_zoff = None, 0, None
                _lib.call("pta_td_trmm_rng", ctypes.byref(gp), self.seed, r0 + lo, n * P, dv.ptr(b["G0"]), npts, sp)
                _lib.call("pta_gwb_mix", dv.ptr(self.d_M), P, dv.ptr(b["G0"]), n, npts, npts, dv.ptr(b["G"]), int(self.mix_variant), sp)
            if beside:
                main.wait_event(self._td_fill_ev)
            elif zmem:
                fill_chunk(b, lo, n, sp)

        for c, lo in enumerate(los):
            b = bufs[c % nbuf]
            n = min(chunk, R - lo)
            if overlap:
                k = c & 1
                with torch.cuda.stream(side):
                    if c >= 2:
                        side.wait_event(ev_done[k])          # the product of chunk c - 2 has released this buffer
                    prepare_chunk(c, ctypes.c_void_p(side.cuda_stream))
                    ev_prep[k].record(side)
                main.wait_event(ev_prep[k])
            else:
                prepare_chunk(c, s)
            if npts:
                tp.gw_G = b["G"].data_ptr()
            if zmem:
                tp.z, tp.ld_z, tp.blk_zoff = b["z"].data_ptr(), b["z"].stride(0), self._td_zoff.data_ptr()
            _lib.call("pta_td_trmm_rng", ctypes.byref(tp), self.seed, r0 + lo, n, ctypes.c_void_p(out.data_ptr() + 8 * lo * out.stride(0)),
                      out.stride(0), s)
            if overlap:
                ev_done[c & 1].record(main)
        return out

    # ---------------------------------------------------------------- draws / replay ------------
    def dump_draws_td(self, r):
        """the deviates realisation r uses in generate_td(): {'td': [z_a[N_a]], 'gwb': z[P, npts] (if a GWB is set)}."""
        if not getattr(self, "_td_prepared", False):
            self.prepare_td()
        from .engine import stream_id
        s = dv.stream_ptr()
        d = {"td": []}
        for a in range(self.P):
            n = int(self.counts[a])
            npair = (n + 1) // 2
            buf = dv.empty((2 * npair,))
            _lib.call("pta_rng_fill_normal", self.seed, r, 1, stream_id(STREAM_TD, a), npair, 1, dv.ptr(buf), None, 2 * npair, int(self.rng_fast), s)
            d["td"].append(buf.cpu().numpy()[:n])
        if self.plan.gw_npts:
            npts = self.plan.gw_npts
            npair = (npts + 1) // 2
            buf = dv.empty((self.P, 2 * npair))
            for a in range(self.P):
                _lib.call("pta_rng_fill_normal", self.seed, r, 1, stream_id(STREAM_TDGW, a), npair, 1,
                          ctypes.c_void_p(buf.data_ptr() + 16 * npair * a), None, 2 * npair, int(self.rng_fast), s)
            d["gwb"] = buf.cpu().numpy()[:, :npts]
        return d

    def td_factor(self, a):
        """lower Cholesky factor of pulsar a's covariance as a [N_a, N_a] device tensor (copy, upper triangle zeroed)."""
        n, ld = int(self.counts[a]), self.td_ld[a]
        v = self.d_Ltd[int(self.td_pos[a]):int(self.td_pos[a]) + n * ld].view(n, ld)[:, :n]
        return torch.tril(v).contiguous()

    def gw_grid_factor(self):
        """lower Cholesky factor of the GWB grid covariance as a [npts, npts] device tensor (copy, upper triangle zeroed)."""
        npts = self.plan.gw_npts
        return torch.tril(self.d_Lg[:, :npts]).contiguous()

    def replay_td(self, draws_list):
        """caller-supplied deviates (shaped like dump_draws_td()) through the explicit-operand kernels - pta_td_trmm (plain
        MFMA GEMM on a stored Z), pta_gwb_mix, pta_gwb_interp: the cross-check of the in-register-draw kernel."""
        if not getattr(self, "_td_prepared", False):
            self.prepare_td()
        s = dv.stream_ptr()
        R, P, N = len(draws_list), self.P, self.n_toa
        out = dv.zeros((R, N))
        for a in range(P):
            n = int(self.counts[a])
            L = self.td_factor(a)
            z = dv.f64(np.stack([d["td"][a] for d in draws_list]))
            _lib.call("pta_td_trmm", dv.ptr(L), n, n, dv.ptr(z), n, R, ctypes.c_void_p(out.data_ptr() + 8 * int(self.off[a])), N, 0, 1, s)
            torch.cuda.current_stream().synchronize()
        if self.plan.gw_npts:
            npts = self.plan.gw_npts
            Lg = self.gw_grid_factor()
            zg = dv.f64(np.stack([d["gwb"] for d in draws_list]).reshape(R * P, npts))
            G0, G = dv.empty((R * P, npts)), dv.empty((R * P, npts))
            _lib.call("pta_td_trmm", dv.ptr(Lg), npts, npts, dv.ptr(zg), npts, R * P, dv.ptr(G0), npts, 0, 1, s)
            _lib.call("pta_gwb_mix", dv.ptr(self.d_M), P, dv.ptr(G0), R, npts, npts, dv.ptr(G), 0, s)
            _lib.call("pta_gwb_interp", dv.ptr(G), npts, P, npts, dv.ptr(self.d_ut), dv.ptr(self.d_toa_s), dv.ptr(self.d_psr_of),
                      dv.ptr(self.d_jlo), N, R, ctypes.c_double(1.0), dv.ptr(out), N, 1, s)
        if self.plan.det:
            out += self.d_det
        torch.cuda.current_stream().synchronize()
        return out
