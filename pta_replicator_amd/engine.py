"""ReplicaEngine: R independent realisations of a whole pulsar-timing array per call, on one MI355X.

The reference produces one realisation per sequence of ``add_*`` calls (a Python loop over pulsars, one pass
through PINT per call).  For ensembles the realisation-independent work - Fourier design matrices
(red_noise.py:36-103), epoch maps (white_noise.py:7-44), ORF + Cholesky (red_noise.py:200-235), spectrum
(:243-265), the DFT twiddle matrix, interpolation brackets (:286-287), CGW waveforms (deterministic.py:13-185) -
is done once in ``prepare()``; ``generate()`` then runs three kernels per batch:

  pta_engine_rn_coef   60 red-noise coefficients per (realisation, pulsar)
  pta_gwb_idft_rng     pruned inverse DFT of on-chip draws, fp64 MFMA      + pta_gwb_mix (ORF mix, MFMA)
  pta_engine_synth     fused RN + GWB-interp + EFAC/EQUAD + ECORR + deterministic -> out[R, sum N_a]

Draws are counter-based (Philox-4x32-10 keyed by seed; counter = pair, stream, realisation), so realisation r
is the same numbers whatever the batch size, launch geometry or number of GPUs.  ``replay()`` pushes
caller-supplied draws (e.g. NumPy's legacy stream in the reference's order, or ``dump_draws()``) through the
per-signal kernels - that is the parity path.

Realisations are defined on the IDEAL TOAs.  The reference applies each signal's delay to the TOAs before the
next ``add_*`` reads them (adjust_TOAs in every function); the induced difference is O(delay * f) ~ 1e-12
relative (SURVEY.md §7 "hard parts" ii) and is the stated tolerance floor of batched-vs-sequential parity.
"""
import ctypes

import numpy as np
import torch

from . import _lib, device as dv
from . import deterministic as det
from . import red_noise as rn
from . import white_noise as wn
from ._position import ra_dec
from .constants import DAY_IN_SEC, YEAR_IN_SEC
from .engine_td import TimeDomainMixin

STREAM_GWB, STREAM_RN, STREAM_WN, STREAM_ECORR, STREAM_TD, STREAM_TDGW = 1, 2, 3, 4, 5, 6


def stream_id(kind, pulsar):
    return ((kind << 24) | (pulsar & 0xFFFFFF)) & 0xFFFFFFFF


class ReplicaEngine(TimeDomainMixin):
    def __init__(self, psrs, seed=0):
        # pulsars with the reference's SimulatedPulsar surface are used as they are; enterprise-style ones (toas [s] / toaerrs [s] /
        # flags / pos as plain arrays: what simulate.py:91-95 hands on) are wrapped once through simulate.from_enterprise
        from .simulate import as_simulated
        self.psrs = [as_simulated(p) for p in psrs]
        self.P = len(self.psrs)
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.names = [p.name for p in self.psrs]
        # ideal TOAs: float64 MJD (GWB/ECORR/CGW read get_mjds, red_noise.py:287, white_noise.py:158,
        # deterministic.py:98) and float64(tdbld)*86400 (RN, red_noise.py:123)
        self.mjd = [np.asarray(p.toas.get_mjds().value, dtype=np.float64) for p in self.psrs]
        self.tdb_s = [np.array(p.toas.table["tdbld"], dtype="float64") * DAY_IN_SEC for p in self.psrs]
        self.sigma_s = [np.asarray(p.toas.get_errors().to("s").value, dtype=np.float64) for p in self.psrs]
        self.counts = np.array([len(m) for m in self.mjd])
        self.off = np.concatenate([[0], np.cumsum(self.counts)]).astype(np.int64)
        self.n_toa = int(self.off[-1])
        self._rn = None
        self._wn = None
        self._ec = None
        self._gw = None
        self._det = None
        self._delays = None
        self._prepared = False
        # frequency -> time transform of the GWB in throughput mode: "auto" = chirp-z FFT when it fits, else the MFMA DFT-GEMM
        self.gwb_transform = "auto"
        # per-engine options, copied into the plan / tables structs of every call (the library has no process-wide switches)
        self.rng_fast = 0        # 1 = fp32-transcendental Gaussian transform (opt-in; all ranks of a job must agree)
        self.synth_variant = 0   # pta_engine_plan.synth_variant
        self.czt_variant = 0     # pta_engine_tables.czt_variant
        self.mix_variant = 0     # pta_engine_tables.mix_variant
        self.idft_variant = 1    # column tiling of the DFT-GEMM form; fixed at prepare() (the twiddle layout depends on it)
        # how the GWB of generate() is drawn: "fourier" (default) = the reference's 2 Nf complex-bin deviates per pulsar through
        # its frequency-domain synthesis (replayable through the reference algebra, dump_draws / replay); "grid" = npts deviates
        # per pulsar through the Cholesky factor of the covariance that synthesis implies on the npts-sample grid (SURVEY.md
        # App. A.1) - same distribution, 10x fewer deviates, the transform becomes an MFMA triangular product
        self.gwb_mode = "fourier"
        # how the EFAC/EQUAD term of generate() is drawn: "reference" (default) = two deviates per TOA, (efac sigma) z1 + (efac equad) z2,
        # the reference's own draws (white_noise.py:105-109: replayable, dump_draws / replay); "single" = ONE deviate per TOA with the
        # combined amplitude sqrt((efac sigma)^2 + (efac equad | equad)^2) - same distribution, half the Box-Muller pairs of the fused
        # kernel, not the reference's draw order (opt-in secondary number of bench.py, never `value`)
        self.wn_mode = "reference"
        self.workspace_bytes = 8 << 30   # upper bound of the per-batch workspace of generate() (cap by bytes, not only by count)

    # ---------------------------------------------------------------- configuration -------------
    def set_red_noise(self, log10_amplitude, spectral_index, components=30, libstempo_convention=False):
        """per-pulsar power-law red noise; entries may be None to skip a pulsar (red_noise.py:106-135)."""
        self._rn = dict(A=list(np.broadcast_to(np.asarray(log10_amplitude, dtype=object), (self.P,))),
                        g=list(np.broadcast_to(np.asarray(spectral_index, dtype=object), (self.P,))),
                        components=int(components), libstempo=bool(libstempo_convention))
        self._prepared = False

    def set_white_noise(self, efac=1.0, log10_equad=None, flagid="f", flags=None, tnequad=False):
        """EFAC/EQUAD (white_noise.py:47-125).  Scalars, per-pulsar lists of scalars, or (with `flags`, a per-pulsar
        list of flag lists) per-pulsar lists of per-flag arrays."""
        self._wn = dict(efac=efac, log10_equad=log10_equad, flagid=flagid, flags=flags, tnequad=bool(tnequad))
        self._prepared = False

    def set_jitter(self, log10_ecorr, flagid="f", flags=None, coarsegrain=0.1):
        """ECORR (white_noise.py:128-198)."""
        self._ec = dict(log10_ecorr=log10_ecorr, flagid=flagid, flags=flags, coarsegrain=float(coarsegrain))
        self._prepared = False

    def set_gwb(self, log10_amplitude, spectral_index, no_correlations=False, turnover=False,
                clm=(np.sqrt(4.0 * np.pi),), lmax=0, f0=1e-9, beta=1, power=1, userSpec=None, npts=600, howml=10):
        """common GWB process (red_noise.py:138-298)."""
        self._gw = dict(A=log10_amplitude, g=spectral_index, no_correlations=no_correlations, turnover=turnover, clm=clm,
                        lmax=lmax, f0=f0, beta=beta, power=power, userSpec=userSpec, npts=npts, howml=howml)
        self._prepared = False

    def add_cgw(self, **kw):
        """a continuous-wave source shared by all pulsars (deterministic.py:13-185); same keyword arguments."""
        self._det = (self._det or []) + [kw]
        self._prepared = False

    def add_delays(self, delays):
        """any precomputed deterministic delay (seconds; one array per pulsar or one concatenated array), e.g. the
        ``added_signals_time`` entry left by add_burst / add_gw_memory / add_catalog_of_cws: added to every realisation."""
        d = np.concatenate([np.asarray(x, dtype=np.float64).ravel() for x in delays]) if isinstance(delays, (list, tuple)) \
            else np.asarray(delays, dtype=np.float64).ravel()
        if d.shape != (self.n_toa,):
            raise ValueError("delays must cover all {} TOAs, got {}".format(self.n_toa, d.shape))
        self._delays = d if getattr(self, "_delays", None) is None else self._delays + d
        self._prepared = False

    # ---------------------------------------------------------------- helpers -------------------
    def _flag_array(self, a, flagid):
        return np.array([f[flagid] for f in self.psrs[a].toas.table["flags"].data])

    # ---------------------------------------------------------------- prepare -------------------
    def prepare(self):
        dev = dv.require_gpu()
        self._ws = None  # tables of a previous prepare() point at replaced buffers
        self._td_prepared = False  # and so do the dense factors of TD mode
        self._gw_grid_ready = False
        s = dv.stream_ptr()
        P, N = self.P, self.n_toa
        self.d_psr_of = dv.i32(np.repeat(np.arange(P), self.counts))
        self.d_idx_in = dv.i32(np.concatenate([np.arange(c) for c in self.counts]))
        self.d_toa_s = dv.f64(np.concatenate([m.astype(float) * 86400 for m in self.mjd]))
        self.plan = _lib.EnginePlan()
        pl = self.plan
        pl.n_toa, pl.n_psr = N, P
        pl.idx_in_psr = self.d_idx_in.data_ptr()

        # ---- red noise: Ft [K, N] over the concatenated TOAs and amp = sqrt(prior) [P, K]
        pl.rn_k = 0
        if self._rn is not None:
            nm = self._rn["components"]
            K = 2 * nm
            self.K = K
            self.d_Ft = dv.zeros((K, N))
            amp = np.zeros((P, K))
            self.rn_freqs = []
            for a in range(P):
                lA, g = self._rn["A"][a], self._rn["g"][a]
                toas = self.tdb_s[a]
                Tspan = toas.max() - toas.min()
                f = 1.0 * np.arange(1, nm + 1) / Tspan
                self.rn_freqs.append(f)
                if lA is None or g is None:
                    continue
                freqs = np.repeat(f, 2)
                fyr = 1 / YEAR_IN_SEC
                prior = (10 ** lA) ** 2 * (freqs / fyr) ** (-g) / (12 * np.pi ** 2 * Tspan) * YEAR_IN_SEC ** 3
                amp[a] = np.sqrt(prior)
                t_d, f_d = dv.f64(toas), dv.f64(f)
                lib_conv = self._rn["libstempo"]
                t_ref = float(toas[0]) if lib_conv else 0.0
                _lib.call("pta_rn_basis", dv.ptr(t_d), len(toas), ctypes.c_double(t_ref), dv.ptr(f_d), None, nm,
                          1 if lib_conv else 0, ctypes.c_void_p(self.d_Ft.data_ptr() + 8 * int(self.off[a])), N, s)
            torch.cuda.current_stream().synchronize()  # t_d/f_d temporaries die with the loop
            self.rn_amp = amp
            self.d_amp = dv.f64(amp)
            pl.rn_k, pl.Ft, pl.ldf = K, self.d_Ft.data_ptr(), N

        # ---- EFAC / EQUAD vectors
        pl.wn_a = pl.wn_b = pl.wn_c = None
        if self._wn is not None:
            c = self._wn
            wa, wb = np.zeros(N), np.zeros(N)
            self.efacvec, self.equadvec = [], []
            for a in range(P):
                n = self.counts[a]
                efac = self._pick(c["efac"], a)
                l10 = self._pick(c["log10_equad"], a)
                flags = self._pick_flags(c["flags"], a)
                equad = 10 ** np.asarray(l10, dtype=float) if l10 is not None else 0.0
                if flags is None:
                    if not np.isscalar(efac) or np.ndim(equad) != 0:
                        raise ValueError("ERROR: If flags is None, efac and equad must be a scalar")
                    ev, qv = np.ones(n) * efac, np.ones(n) * float(equad)
                else:
                    ev, qv = np.zeros(n), np.zeros(n)
                    if not (len(efac) == len(flags) and len(equad) == len(flags)):
                        raise ValueError("ERROR: flags must be same length as efac and log10_equad")
                    tf = self._flag_array(a, c["flagid"])
                    for ct, flag in enumerate(flags):
                        ev[flag == tf] = efac[ct]
                        qv[flag == tf] = equad[ct]
                self.efacvec.append(ev)
                self.equadvec.append(qv)
                sl = slice(self.off[a], self.off[a + 1])
                wa[sl] = ev * self.sigma_s[a]
                wb[sl] = qv if c["tnequad"] else ev * qv
            self.d_wn_a, self.d_wn_b, self.d_wn_c = dv.f64(wa), dv.f64(wb), dv.f64(np.hypot(wa, wb))
            pl.wn_a, pl.wn_b, pl.tnequad = self.d_wn_a.data_ptr(), self.d_wn_b.data_ptr(), int(c["tnequad"])

        # ---- ECORR epoch maps
        pl.epoch_of = pl.ecorr_toa = None
        if self._ec is not None:
            c = self._ec
            ep_all, ec_all = np.zeros(N, dtype=np.int32), np.zeros(N)
            self.epoch_of, self.ecorrvec = [], []
            for a in range(P):
                l10 = self._pick(c["log10_ecorr"], a)
                flags = self._pick_flags(c["flags"], a)
                epoch_of, first = wn.epoch_map(self.mjd[a], c["coarsegrain"])
                ne = len(first)
                if l10 is None:
                    vec = np.zeros(ne)
                elif flags is None:
                    if not np.isscalar(l10):
                        raise ValueError("ERROR: If flags is None, jitter must be a scalar")
                    vec = np.ones(ne) * 10 ** l10
                else:
                    ecorr = 10 ** np.asarray(l10, dtype=float)
                    if len(ecorr) != len(flags):
                        raise ValueError("ERROR: flags must be same length as jitter")
                    aveflags = self._flag_array(a, c["flagid"])[first]
                    vec = np.zeros(ne)
                    for ct, flag in enumerate(flags):
                        vec[flag == aveflags] = ecorr[ct]
                self.epoch_of.append(epoch_of)
                self.ecorrvec.append(vec)
                sl = slice(self.off[a], self.off[a + 1])
                ep_all[sl] = epoch_of
                ec_all[sl] = vec[epoch_of]
            self.d_epoch_of, self.d_ecorr_toa = dv.i32(ep_all), dv.f64(ec_all)
            pl.epoch_of, pl.ecorr_toa = self.d_epoch_of.data_ptr(), self.d_ecorr_toa.data_ptr()

        # ---- workgroup tiles: <= ENGINE_TILE consecutive TOAs of ONE pulsar, plus the ECORR pair range they touch
        T = _lib.ENGINE_TILE
        t_psr, t_start, t_count, t_ep0, t_epn = [], [], [], [], []
        for a in range(P):
            for s0 in range(0, int(self.counts[a]), T):
                cnt = min(T, int(self.counts[a]) - s0)
                t_psr.append(a); t_start.append(int(self.off[a]) + s0); t_count.append(cnt)
                if self._ec is not None:
                    e = self.epoch_of[a][s0:s0 + cnt]
                    p0, p1 = int(e.min()) >> 1, int(e.max()) >> 1
                    t_ep0.append(p0); t_epn.append(p1 - p0 + 1 if p1 - p0 + 1 <= _lib.ENGINE_EPMAX else 0)
                else:
                    t_ep0.append(0); t_epn.append(0)
        self.d_tiles = [dv.i32(x) for x in (t_psr, t_start, t_count, t_ep0, t_epn)]
        pl.n_tiles = len(t_psr)
        pl.tile_psr, pl.tile_start, pl.tile_count, pl.tile_ep0, pl.tile_epn = [x.data_ptr() for x in self.d_tiles]

        # ---- GWB: grid, ORF, Cholesky, spectrum, twiddles, brackets
        pl.gw_npts = 0
        if self._gw is not None:
            c = self._gw
            grid = rn.gwb_time_grid(self.psrs_ideal_view(), c["npts"], c["howml"])
            self.grid = grid
            npts, Nf = grid["npts"], grid["Nf"]
            ORF = rn.gwb_orf_device(self.psrs, c["no_correlations"], c["clm"], c["lmax"])
            self.ORF = ORF.clone()
            self.d_M = rn.cholesky_device(ORF)
            self.C = rn.gwb_spectrum(grid["f"], grid["dur"], c["howml"], c["A"], c["g"], c["turnover"], c["f0"], c["beta"],
                                     c["power"], c["userSpec"])
            self.ldt = rn.pad16(npts)
            self.d_T = dv.empty((2 * (Nf - 2), self.ldt))
            sqrtC = dv.f64(self.C ** 0.5)
            _lib.call("pta_gwb_twiddle", dv.ptr(sqrtC), Nf, npts, 10, ctypes.c_double(1.0 / grid["dt"]), dv.ptr(self.d_T), self.ldt, s)
            # throughput-mode layout of the same twiddles: half window, slab-major, plus the per-bin rotation
            nrot = ctypes.c_int64(0)
            nsym = _lib.lib.pta_gwb_twiddle_sym_size(Nf, npts, self.idft_variant, ctypes.byref(nrot))
            self._idft_variant_built = self.idft_variant
            self.d_Tsym, self.d_rot = dv.empty((nsym,)), dv.empty((nrot.value,))
            _lib.call("pta_gwb_twiddle_sym", dv.ptr(sqrtC), Nf, npts, 10, ctypes.c_double(1.0 / grid["dt"]), dv.ptr(self.d_Tsym),
                      dv.ptr(self.d_rot), self.idft_variant, s)
            # chirp-z tables (default frequency -> time transform whenever one 4096-point convolution covers the window)
            self.use_czt = bool(_lib.lib.pta_gwb_czt_fits(Nf, npts, 10)) and self.gwb_transform != "gemm"
            self.d_czt = None
            if self.use_czt:
                self.d_czt = [dv.empty((2 * 4096,)), dv.empty((2 * 4096,)), dv.empty((2 * 4096,)), dv.empty((2 * npts,))]
                _lib.call("pta_gwb_czt_setup", dv.ptr(sqrtC), Nf, npts, 10, ctypes.c_double(1.0 / grid["dt"]),
                          *[dv.ptr(x) for x in self.d_czt], s)
            self.d_ut = dv.f64(grid["ut"])
            self.d_jlo = dv.empty((N,), dtype=torch.int32)
            _lib.call("pta_gwb_bracket", dv.ptr(self.d_ut), npts, dv.ptr(self.d_toa_s), N, dv.ptr(self.d_jlo), s)
            torch.cuda.current_stream().synchronize()
            self.d_gw_w = dv.empty((N,))
            _lib.call("pta_gwb_weights", dv.ptr(self.d_ut), npts, dv.ptr(self.d_toa_s), dv.ptr(self.d_jlo), N, dv.ptr(self.d_gw_w), s)
            torch.cuda.current_stream().synchronize()
            pl.gw_npts, pl.gw_jlo, pl.gw_w = npts, self.d_jlo.data_ptr(), self.d_gw_w.data_ptr()

        # ---- deterministic signals (CGW), summed once
        pl.det = None
        extra = getattr(self, "_delays", None)
        if self._det or extra is not None:
            self.d_det = dv.zeros((N,)) if extra is None else dv.f64(extra)
            mjd_all = dv.f64(np.concatenate(self.mjd)) if self._det else None     # one upload; the kernel takes host scalars per call
            for kw in self._det or []:
                for a in range(P):
                    ra, dec = ra_dec(self.psrs[a])
                    par, _, _, _ = det.cgw_parameters(np.pi / 2 - dec, ra, **{k: v for k, v in kw.items() if k != "signal_name"})
                    par = np.ascontiguousarray(par)          # copied into the kernel argument at launch: no lifetime issue
                    o = int(self.off[a])
                    _lib.call("pta_cgw", ctypes.c_void_p(mjd_all.data_ptr() + 8 * o), len(self.mjd[a]), dv.hptr(par),
                              ctypes.c_void_p(self.d_det.data_ptr() + 8 * o), 1, s)
            if self._det:
                torch.cuda.current_stream().synchronize()      # once, not once per pulsar per source
            pl.det = self.d_det.data_ptr()
        self._prepared = True
        # set-up time is where one-off driver costs belong: with measurement noise configured the dense path can follow, and the
        # factorisation's internal streams (hardware queues, ~10 ms each on this stack: profiles/r05_prepare_td_first_call.txt) are
        # created now instead of inside the first prepare_td() (td_warmup = False leaves them to be created on demand)
        if self._wn is not None and getattr(self, "td_warmup", True):
            _lib.call("pta_potrf_warmup", 2)
        return self

    def psrs_ideal_view(self):
        """objects exposing first_MJD/last_MJD of the IDEAL TOAs for the GWB grid (red_noise.py:182-183)."""
        class _V:
            pass
        out = []
        for m in self.mjd:
            v = _V()
            v.toas = _V()
            v.toas.first_MJD = _V()
            v.toas.last_MJD = _V()
            v.toas.first_MJD.value = float(m.min())
            v.toas.last_MJD.value = float(m.max())
            out.append(v)
        return out

    def _pick(self, val, a):
        """per-pulsar entry of a scalar / list-of-P parameter."""
        if val is None or np.isscalar(val):
            return val
        if len(val) != self.P:
            raise ValueError(f"expected a scalar or one entry per pulsar ({self.P}), got {len(val)}")
        return val[a]

    def _pick_flags(self, flags, a):
        if flags is None:
            return None
        if len(flags) != self.P:
            raise ValueError("flags must be a per-pulsar list of flag lists")
        return flags[a]

    # ---------------------------------------------------------------- throughput mode -----------
    def workspace(self, R):
        """device buffers for a batch of R realisations (reused across generate() calls) and the pta_engine_tables that
        point at them."""
        ws = getattr(self, "_ws", None)
        if ws is None or ws["R"] < R:
            ws = {"R": R}
            tb = _lib.EngineTables()
            if self.plan.rn_k:
                ws["coef"] = dv.empty((R, self.P, self.K))
                tb.rn_amp, tb.ws_coef = self.d_amp.data_ptr(), ws["coef"].data_ptr()
            if self.plan.gw_npts:
                ws["G0"] = dv.empty((R, self.P, self.plan.gw_npts))
                ws["G"] = dv.empty((R, self.P, self.plan.gw_npts))
                tb.Mchol, tb.ws_G0, tb.ws_G = self.d_M.data_ptr(), ws["G0"].data_ptr(), ws["G"].data_ptr()
                tb.gw_nf, tb.gw_i0 = self.grid["Nf"], 10
                if getattr(self, "d_czt", None) is not None:
                    tb.czt_pre, tb.czt_FB, tb.czt_tw, tb.czt_post = (x.data_ptr() for x in self.d_czt)
                tb.Tsym, tb.rot = self.d_Tsym.data_ptr(), self.d_rot.data_ptr()
                tb.idft_variant = self._idft_variant_built
            ws["tables"] = tb
            self._ws = ws
        ws["tables"].czt_variant, ws["tables"].mix_variant = int(self.czt_variant), int(self.mix_variant)
        if self.plan.gw_npts:
            if self.use_czt and getattr(self, "d_czt", None) is None:
                raise ValueError("use_czt was switched on after prepare(): the chirp-z tables were not built")
            ws["tables"].use_czt = 1 if self.use_czt else 0
        self.plan.rng_fast, self.plan.synth_variant = int(self.rng_fast), int(self.synth_variant)
        return ws

    def max_batch(self):
        """realisations per launch sequence: the launch-grid limit of the mix / fused kernels (65536) and the workspace byte
        budget (coef + G0 + G per realisation: 0.68 MB at 68 pulsars, so 8 GiB hold 12 600 realisations)."""
        per_real = 8 * self.P * ((self.K if self.plan.rn_k else 0) + 2 * self.plan.gw_npts)
        return int(max(16, min(65536, self.workspace_bytes // max(per_real, 1))))

    def generate(self, R, r0=0, out=None):
        """out[R, n_toa] (device tensor, seconds): realisations r0 .. r0+R-1, every deviate drawn on chip: one call of
        pta_engine_generate (coefficients -> GWB transform -> mix -> fused synthesis, all queued on the current stream)."""
        if not self._prepared:
            self.prepare()
        if out is None:
            out = dv.empty((R, self.n_toa))
        step = self.max_batch()
        ws = self.workspace(min(R, step))
        grid_mode = self.gwb_mode == "grid" and self.plan.gw_npts
        if grid_mode:
            self._prepare_gw_grid_factor()
            self.tdgw_plan.rng_fast = int(self.rng_fast)
            self.tdgw_plan.z, self.tdgw_plan.ld_z, self.tdgw_plan.blk_zoff = None, 0, None   # (generate_td may have pointed it at a deviate buffer)
        s = dv.stream_ptr()
        for lo in range(0, R, step):
            n = min(step, R - lo)
            optr = ctypes.c_void_p(out.data_ptr() + 8 * lo * out.stride(0))
            if not grid_mode:
                _lib.call("pta_engine_generate", ctypes.byref(self._plan_for_mode()), ctypes.byref(ws["tables"]), self.seed, r0 + lo, n, optr, out.stride(0), s)
                continue
            pl = _lib.EnginePlan.from_buffer_copy(self._plan_for_mode())
            if pl.rn_k:
                _lib.call("pta_engine_rn_coef", self.seed, r0 + lo, n, self.P, self.K, dv.ptr(self.d_amp), dv.ptr(ws["coef"]), int(self.rng_fast), s)
                pl.rn_coef = ws["coef"].data_ptr()
            _lib.call("pta_td_trmm_rng", ctypes.byref(self.tdgw_plan), self.seed, r0 + lo, n * self.P, dv.ptr(ws["G0"]), pl.gw_npts, s)
            _lib.call("pta_gwb_mix", dv.ptr(self.d_M), self.P, dv.ptr(ws["G0"]), n, pl.gw_npts, pl.gw_npts, dv.ptr(ws["G"]), int(self.mix_variant), s)
            pl.gw_G = ws["G"].data_ptr()
            _lib.call("pta_engine_synth", ctypes.byref(pl), self.seed, r0 + lo, n, optr, out.stride(0), s)
        # per-kernel callers (bench.py, replay) read the workspace pointers from the plan
        if self.plan.rn_k:
            self.plan.rn_coef = ws["coef"].data_ptr()
        if self.plan.gw_npts:
            self.plan.gw_G = ws["G"].data_ptr()
        return out

    def _plan_for_mode(self):
        """the plan generate() launches with: the shared one, or - wn_mode "single" - a private copy whose white-noise operands are
        replaced by the combined amplitude vector (pta_engine_plan.wn_c)."""
        if self.wn_mode == "reference" or not self.plan.wn_a:
            return self.plan
        if self.wn_mode != "single":
            raise ValueError(f"wn_mode={self.wn_mode!r} must be 'reference' or 'single'")
        pl = _lib.EnginePlan.from_buffer_copy(self.plan)
        pl.wn_a = pl.wn_b = None
        pl.wn_c = self.d_wn_c.data_ptr()
        return pl

    def _require_reference_draws(self, what):
        """the replay entry points describe the reference's draws (two deviates per TOA, white_noise.py:105-109): in wn_mode "single"
        generate() draws something else, so they refuse instead of silently answering for another mode."""
        if self.wn_mode != "reference" and self._wn is not None:
            raise ValueError(f"{what}() describes the reference-order draws; wn_mode={self.wn_mode!r} realisations are not replayable "
                             "(use dump_draws_wn_single / generate_per_signal, or set wn_mode = 'reference')")

    def dump_draws_wn_single(self, r):
        """the ONE deviate per TOA realisation r uses in wn_mode "single": TOA idx of pulsar a takes branch (idx >> 4) & 1 of pair
        idx & ~16 of stream (WN, a) - one array per pulsar."""
        s = dv.stream_ptr()
        out = []
        for a in range(self.P):
            n = int(self.counts[a])
            z0, z1 = dv.empty((n,)), dv.empty((n,))
            _lib.call("pta_rng_fill_normal", self.seed, r, 1, stream_id(STREAM_WN, a), n, 0, dv.ptr(z0), dv.ptr(z1), n, int(self.rng_fast), s)
            idx = np.arange(n)
            z0, z1 = z0.cpu().numpy(), z1.cpu().numpy()
            out.append(np.where((idx >> 4) & 1, z1[idx & ~16], z0[idx & ~16]))
        return out

    def generate_per_signal(self, R, r0=0):
        """{'rn', 'gwb', 'wn', 'ecorr', 'det', 'total'}: the same realisations as generate(R, r0), one [R, n_toa] array per
        signal (the batched counterpart of the reference's per-signal ``added_signals_time`` entries).  Every deviate is a pure
        function of (seed, realisation, stream, index), so running the fused kernel once per signal with the other inputs
        switched off reproduces exactly the deviates of the combined pass."""
        if not self._prepared:
            self.prepare()
        if R > self.max_batch():
            raise ValueError(f"generate_per_signal: at most {self.max_batch()} realisations per call (one workspace batch)")
        total = self.generate(R, r0=r0)                      # also fills the workspace (coefficients, mixed GWB grid series)
        s = dv.stream_ptr()
        base = self._plan_for_mode()                         # wn_mode "single": the white-noise term is the one-deviate form, as in generate()
        keep = (self.plan.rn_k, self.plan.gw_npts, base.wn_a, base.wn_b, self.plan.ecorr_toa, self.plan.epoch_of, self.plan.det, base.wn_c)
        out = {"total": total}

        def one(name, **on):
            pl = _lib.EnginePlan.from_buffer_copy(self.plan)   # a private copy: the shared plan is never mutated
            pl.rn_k, pl.gw_npts, pl.wn_a, pl.wn_b, pl.ecorr_toa, pl.epoch_of, pl.det, pl.wn_c = (
                on.get("rn_k", 0), on.get("gw_npts", 0), on.get("wn_a"), on.get("wn_b"), on.get("ecorr_toa"),
                on.get("epoch_of"), on.get("det"), on.get("wn_c"))
            buf = dv.empty((R, self.n_toa))
            _lib.call("pta_engine_synth", ctypes.byref(pl), self.seed, r0, R, dv.ptr(buf), buf.stride(0), s)
            out[name] = buf
        if keep[0]:
            one("rn", rn_k=keep[0])
        if keep[1]:
            one("gwb", gw_npts=keep[1])
        if keep[2] or keep[7]:
            one("wn", wn_a=keep[2], wn_b=keep[3], wn_c=keep[7])
        if keep[4]:
            one("ecorr", ecorr_toa=keep[4], epoch_of=keep[5])
        if keep[6]:
            out["det"] = self.d_det.unsqueeze(0).expand(R, self.n_toa)
        return out

    def stream_to_host(self, total, chunk=480, r0=0):
        """Generator over (first_realisation, host_array[n, n_toa]) covering realisations r0 .. r0+total-1: generation on
        the current stream, device->host copies of the previous chunk on a second stream into two pinned buffers, so the
        PCIe link (2.72 MB per realisation at 68 x 5000) is the only thing waited for.  The yielded array is a view of a
        pinned buffer and is valid ONLY until the generator is resumed: asking for the next chunk immediately queues the copy
        of chunk k+2 into the very buffer chunk k was yielded from - copy the array if it must outlive the next next()."""
        if not self._prepared:
            self.prepare()
        chunk = int(min(chunk, total))
        dev = [dv.empty((chunk, self.n_toa)) for _ in range(2)]
        host = [torch.empty((chunk, self.n_toa), dtype=torch.float64).pin_memory() for _ in range(2)]
        copy_stream = torch.cuda.Stream()
        filled = [torch.cuda.Event(), torch.cuda.Event()]   # generation of buffer b finished
        copied = [torch.cuda.Event(), torch.cuda.Event()]   # its copy to the host finished
        main = torch.cuda.current_stream()
        pending = []                                        # (buffer, first realisation, count) copies in flight
        done = 0
        while done < total or pending:
            if done < total:
                b = (done // chunk) & 1
                n = min(chunk, total - done)
                if done >= 2 * chunk:
                    main.wait_event(copied[b])              # the device buffer is free again
                self.generate(n, r0=r0 + done, out=dev[b][:n])
                filled[b].record(main)
                with torch.cuda.stream(copy_stream):
                    copy_stream.wait_event(filled[b])
                    host[b][:n].copy_(dev[b][:n], non_blocking=True)
                    copied[b].record(copy_stream)
                pending.append((b, r0 + done, n))
                done += n
            if len(pending) == 2 or done >= total:
                b, first, n = pending.pop(0)
                copied[b].synchronize()
                yield first, host[b][:n].numpy()

    # ---------------------------------------------------------------- replay mode ---------------
    def dump_draws(self, r):
        """The normals realisation r uses in generate(), as NumPy arrays in the reference's shapes:
        {'gwb': w[P,Nf] complex, 'rn': [z[K]], 'wn': [(z1[N_a], z2[N_a])], 'ecorr': [z[E_a]]}."""
        self._require_reference_draws("dump_draws")
        if not self._prepared:
            self.prepare()
        s = dv.stream_ptr()
        d = {}
        if self.plan.gw_npts:
            Nf = self.grid["Nf"]
            buf = dv.empty((self.P, 2 * Nf))
            for a in range(self.P):
                _lib.call("pta_rng_fill_normal", self.seed, r, 1, stream_id(STREAM_GWB, a), Nf, 1,
                          ctypes.c_void_p(buf.data_ptr() + 16 * Nf * a), None, 2 * Nf, int(self.rng_fast), s)
            w = buf.cpu().numpy().reshape(self.P, Nf, 2)
            d["gwb"] = w[:, :, 0] + 1j * w[:, :, 1]
        if self.plan.rn_k:
            d["rn"] = []
            for a in range(self.P):
                buf = dv.empty((self.K,))
                _lib.call("pta_rng_fill_normal", self.seed, r, 1, stream_id(STREAM_RN, a), self.K // 2, 1, dv.ptr(buf), None, self.K, int(self.rng_fast), s)
                d["rn"].append(buf.cpu().numpy())
        if self.plan.wn_a:
            d["wn"] = []
            for a in range(self.P):
                n = int(self.counts[a])
                z1, z2 = dv.empty((n,)), dv.empty((n,))
                _lib.call("pta_rng_fill_normal", self.seed, r, 1, stream_id(STREAM_WN, a), n, 0, dv.ptr(z1), dv.ptr(z2), n, int(self.rng_fast), s)
                d["wn"].append((z1.cpu().numpy(), z2.cpu().numpy()))
        if self.plan.ecorr_toa:
            d["ecorr"] = []
            for a in range(self.P):
                ne = len(self.ecorrvec[a])
                npair = (ne + 1) // 2
                buf = dv.empty((2 * npair,))
                _lib.call("pta_rng_fill_normal", self.seed, r, 1, stream_id(STREAM_ECORR, a), npair, 1, dv.ptr(buf), None, 2 * npair, int(self.rng_fast), s)
                d["ecorr"].append(buf.cpu().numpy()[:ne])
        return d

    def replay(self, draws_list, per_signal=False):
        """Push caller-supplied draws through the per-signal kernels (replay mode / parity path).

        draws_list: one dict per realisation, shaped like dump_draws().  Returns out[R, n_toa] (device), or with
        per_signal=True a dict of such tensors keyed 'rn', 'gwb', 'wn', 'ecorr', 'det', 'total'."""
        self._require_reference_draws("replay")
        if not self._prepared:
            self.prepare()
        s = dv.stream_ptr()
        R, P, N = len(draws_list), self.P, self.n_toa
        sig = {}
        pl = self.plan
        if pl.rn_k:
            coef = np.zeros((R, P, self.K))
            for r, d in enumerate(draws_list):
                for a in range(P):
                    coef[r, a] = self.rn_amp[a] * d["rn"][a]
            coef_d = dv.f64(coef)
            out = dv.zeros((R, N))
            for a in range(P):
                n = int(self.counts[a])
                _lib.call("pta_rn_synth", ctypes.c_void_p(self.d_Ft.data_ptr() + 8 * int(self.off[a])), N, n, self.K,
                          ctypes.c_void_p(coef_d.data_ptr() + 8 * a * self.K), P * self.K, R,
                          ctypes.c_void_p(out.data_ptr() + 8 * int(self.off[a])), N, 0, s)
            sig["rn"] = out
        if pl.gw_npts:
            Nf, npts = self.grid["Nf"], pl.gw_npts
            w = np.zeros((R, P, Nf, 2))
            for r, d in enumerate(draws_list):
                w[r, :, :, 0], w[r, :, :, 1] = d["gwb"].real, d["gwb"].imag
            w_d = dv.f64(w)
            G0, G = dv.empty((R * P, npts)), dv.empty((R * P, npts))
            _lib.call("pta_gwb_idft", dv.ptr(w_d), 2 * Nf, R * P, Nf, dv.ptr(self.d_T), self.ldt, npts, dv.ptr(G0), npts, 1, s)
            _lib.call("pta_gwb_mix", dv.ptr(self.d_M), P, dv.ptr(G0), R, npts, npts, dv.ptr(G), int(self.mix_variant), s)
            out = dv.empty((R, N))
            _lib.call("pta_gwb_interp", dv.ptr(G), npts, P, npts, dv.ptr(self.d_ut), dv.ptr(self.d_toa_s), dv.ptr(self.d_psr_of),
                      dv.ptr(self.d_jlo), N, R, ctypes.c_double(1.0), dv.ptr(out), N, 0, s)
            sig["gwb"] = out
            self._last_G = G
        if pl.wn_a:
            z1 = np.zeros((R, N)); z2 = np.zeros((R, N))
            for r, d in enumerate(draws_list):
                z1[r] = np.concatenate([x[0] for x in d["wn"]]); z2[r] = np.concatenate([x[1] for x in d["wn"]])
            sg, ef, eq = dv.f64(np.concatenate(self.sigma_s)), dv.f64(np.concatenate(self.efacvec)), dv.f64(np.concatenate(self.equadvec))
            z1_d, z2_d = dv.f64(z1), dv.f64(z2)
            out = dv.empty((R, N))
            _lib.call("pta_wn", dv.ptr(sg), dv.ptr(ef), dv.ptr(eq), N, int(self._wn["tnequad"]), dv.ptr(z1_d), dv.ptr(z2_d), N, R,
                      dv.ptr(out), N, 0, s)
            sig["wn"] = out
        if pl.ecorr_toa:
            out = dv.zeros((R, N))
            keep = []                                           # operands stay alive until the single synchronise below
            for a in range(P):
                ne, n = len(self.ecorrvec[a]), int(self.counts[a])
                z = np.zeros((R, ne))
                for r, d in enumerate(draws_list):
                    z[r] = d["ecorr"][a]
                z_d, ep_d, ec_d = dv.f64(z), dv.i32(self.epoch_of[a]), dv.f64(self.ecorrvec[a])
                keep += [z_d, ep_d, ec_d]
                _lib.call("pta_ecorr", dv.ptr(ep_d), dv.ptr(ec_d), n, ne, dv.ptr(z_d), ne, R,
                          ctypes.c_void_p(out.data_ptr() + 8 * int(self.off[a])), N, 0, s)
            torch.cuda.current_stream().synchronize()
            del keep
            sig["ecorr"] = out
        total = dv.zeros((R, N))
        for k in ("rn", "gwb", "wn", "ecorr"):
            if k in sig:
                total += sig[k]
        if pl.det:
            sig["det"] = self.d_det.unsqueeze(0).expand(R, N)
            total += self.d_det
        torch.cuda.current_stream().synchronize()
        if per_signal:
            sig["total"] = total
            return sig
        return total

    def inject(self, r, signal_name="engine_realisation"):
        """Write realisation r into the pulsar objects the way the reference's add_* functions do - record the delay in
        ``added_signals`` / ``added_signals_time``, shift the TOAs, rebuild the residuals - but ONCE per pulsar with the summed
        delay (the PINT sink dominates real runs, so it is paid once, not once per signal)."""
        from ._compat import TimeDelta, u
        row = self.generate(1, r0=r)[0].cpu().numpy()
        for a, psr in enumerate(self.psrs):
            dt = row[self.off[a]:self.off[a + 1]] * u.s
            psr.update_added_signals("{}_{}".format(psr.name, signal_name), {"seed": self.seed, "realisation": int(r)}, dt)
            psr.toas.adjust_TOAs(TimeDelta(dt.to("day")))
            psr.update_residuals()
        return row

    # ---------------------------------------------------------------- timing-model projection ---
    def prepare_timing_projection(self, model="spin"):
        """Design matrices M_a (simulate.timing_design_matrix on the IDEAL TOAs: "spin" = offset, F0, F1; "astrometric" = + position,
        proper motion, parallax) and the fit operators Q_a = (M^T W M)^-1 M^T W of every pulsar, W = 1 / sigma^2, on the device
        (K-major over the concatenated TOAs, like Ft).  Realisation independent: computed once."""
        from .simulate import timing_design_matrix
        dv.require_gpu()
        Mt = Qt = None
        for a in range(self.P):
            M, names = timing_design_matrix(self.mjd[a] * 86400.0, model=model)
            w = 1.0 / self.sigma_s[a] ** 2
            # scale the weights (the projection is invariant): sigma ~ 1e-6 s would put M^T W M at 1e12 and lose digits to nothing
            w = w / w.mean()
            A = M.T @ (M * w[:, None])
            Q = np.linalg.solve(A, M.T * w[None, :])
            if Mt is None:
                m = M.shape[1]
                Mt, Qt = np.zeros((m, self.n_toa)), np.zeros((m, self.n_toa))
            Mt[:, self.off[a]:self.off[a + 1]] = M.T
            Qt[:, self.off[a]:self.off[a + 1]] = Q
        self._tm = dict(model=model, m=Mt.shape[0], names=names, Mt=dv.f64(Mt), Qt=dv.f64(Qt), off=dv.i32(self.off), Mt_host=Mt, Qt_host=Qt)
        return self

    def project_timing_model(self, rows, model="spin"):
        """In place: every realisation of ``rows`` ([R, n_toa] device tensor, e.g. the output of generate()) loses what a linearised
        refit of the pulsars' timing models would absorb, r <- r - M (M^T W M)^-1 M^T W r per (realisation, pulsar) - the batched
        counterpart of the reference's fit + residual rebuild through PINT (simulate.py:40-69) for idealised timing models
        (pta_tm_project, one launch).  The weighted-mean subtraction of PINT's default residuals is the one-column special case."""
        if getattr(self, "_tm", None) is None or self._tm["model"] != model:
            self.prepare_timing_projection(model)
        tm = self._tm
        assert rows.dim() == 2 and rows.shape[1] == self.n_toa and rows.stride(1) == 1
        _lib.call("pta_tm_project", dv.ptr(tm["Qt"]), dv.ptr(tm["Mt"]), self.n_toa, tm["m"], dv.ptr(tm["off"]), self.P,
                  ctypes.c_void_p(rows.data_ptr()), rows.stride(0), int(rows.shape[0]), dv.stream_ptr())
        return rows

    def to_enterprise(self, rows, subtract_mean=True, timing_model="spin", fitted=False):
        """enterprise-style pulsar objects for realisations already generated: ``rows`` is a [R, n_toa] tensor / array (e.g. the
        output of generate() or generate_td()); returns R lists of P ``ArrayEnterprisePulsar`` whose ``toas`` are the ideal TOAs
        shifted by the realisation's delay (what the reference's hand-off carries, simulate.py:91-95) and whose ``residuals`` are the
        injected delays with the weighted mean removed (what PINT's Residuals would report for an idealised pulsar, SURVEY.md §8
        a16) - the hand-off of SURVEY.md §8f rank 4 for whole ensembles, without a par/tim round trip.  Works for array-backed,
        PINT-backed and foreign pulsars alike: everything is read from the engine's own copies of the ideal TOAs / errors and the
        pulsar's flag table.  ``fitted=True``: the residuals are POST-FIT - the columns of ``Mmat`` are projected out of every
        realisation on the device (project_timing_model) before the hand-off; the TOAs still carry the full injected delay."""
        from .simulate import ArrayEnterprisePulsar
        from ._position import ra_dec
        if fitted:   # post-fit residuals: the design matrix handed over (Mmat) is projected out on the device, whole ensemble at once
            dev = rows.detach().clone() if hasattr(rows, "detach") else dv.f64(np.atleast_2d(np.asarray(rows)))
            if dev.dim() == 1:
                dev = dev[None, :].contiguous()
            rows = self.project_timing_model(dev, timing_model)
            subtract_mean = False
        arr = rows.detach().cpu().numpy() if hasattr(rows, "detach") else np.asarray(rows)
        if arr.ndim == 1:
            arr = arr[None, :]
        meta = []
        for a, psr in enumerate(self.psrs):
            toas = psr.toas
            flags = list(toas.table["flags"].data)
            freq = getattr(toas, "freqs_mhz", None)
            if freq is None:
                try:
                    freq = np.asarray(toas.get_freqs().to("MHz").value, dtype=np.float64)
                except AttributeError:
                    freq = 1440.0
            meta.append((flags, np.asarray(freq, dtype=np.float64) * np.ones(int(self.counts[a])), ra_dec(psr, default=(0.0, 0.0))))
        out = []
        for row in arr:
            psrs = []
            for a, psr in enumerate(self.psrs):
                x = row[self.off[a]:self.off[a + 1]]
                res = x
                if subtract_mean:
                    w = 1.0 / self.sigma_s[a] ** 2
                    res = x - np.sum(x * w) / np.sum(w)
                flags, freq, (ra, dec) = meta[a]
                mjd = self.mjd[a].astype(np.longdouble) + (x / 86400.0).astype(np.longdouble)
                psrs.append(ArrayEnterprisePulsar(psr.name, np.asarray(mjd, dtype=np.float64), res, self.sigma_s[a] * 1e6, freq, flags, ra, dec,
                                                  timing_model=timing_model))
            out.append(psrs)
        return out

    def write_tim_ensemble(self, rows, outdir, r0=0):
        """Batched counterpart of SimulatedPulsar.write_partim (simulate.py:71-77) for array-backed pulsars: one directory
        ``real_<r>/`` per realisation with a Tempo2 tim file per pulsar whose TOAs are the ideal TOAs shifted by the injected
        delay - without touching the pulsar objects (no per-realisation adjust_TOAs / residual rebuild) - and, ONCE (``par/``), the par
        file of every pulsar: the timing models are the same in every realisation."""
        import os
        from .simulate import ArrayTOAs, minimal_par, write_tim
        arr = rows.detach().cpu().numpy() if hasattr(rows, "detach") else np.asarray(rows)
        if arr.ndim == 1:
            arr = arr[None, :]
        for psr in self.psrs:
            if not isinstance(psr.toas, ArrayTOAs):
                raise NotImplementedError("write_tim_ensemble needs array-backed pulsars (PINT-backed: use inject() + write_partim)")
        os.makedirs(os.path.join(outdir, "par"), exist_ok=True)
        for psr in self.psrs:
            with open(os.path.join(outdir, "par", f"{psr.name}.par"), "w") as fh:
                fh.write(psr.par_text if getattr(psr, "par_text", None) else minimal_par(psr.name, psr.loc))
        paths = []
        for k, row in enumerate(arr):
            d = os.path.join(outdir, f"real_{r0 + k:06d}")
            os.makedirs(d, exist_ok=True)
            for a, psr in enumerate(self.psrs):
                mjd = psr.toas.mjd0_ld + (row[self.off[a]:self.off[a + 1]] / 86400.0).astype(np.longdouble)
                path = os.path.join(d, f"{psr.name}.tim")
                write_tim(path, psr.name, mjd, psr.toas.errors_us, psr.toas.freqs_mhz, psr.toas.flags)
                paths.append(path)
        return paths

    def split(self, arr):
        """per-pulsar views of an [..., n_toa] array."""
        return [arr[..., self.off[a]:self.off[a + 1]] for a in range(self.P)]
