"""Red noise and GWB injection on the MI355X: ``add_red_noise`` / ``add_gwb`` with the reference's signatures.

Mirrors ``pta_replicator/red_noise.py`` (create_fourier_design_matrix_red :36-103, add_red_noise :106-135,
add_gwb :138-298).  Host Python keeps what is bookkeeping or parity-critical scalar work - seeding and the
NumPy legacy draws in the reference's consumption order, the Nf knife-edge frequency grid (:230-232), the
spectrum C(f) (:243-265) and the PINT sink - and every array operation runs in HIP kernels through the
ctypes C ABI (include/pta_replicator_amd.h): design matrix + synthesis, ORF basis, Cholesky, pruned inverse
DFT on fp64 MFMA, ORF mix, interpolation onto the TOAs.  This is "replay mode": same draws, same algebra,
agreement with the reference at the 1e-12 level (tests/test_gpu_parity.py).
"""
import ctypes

import numpy as np
import torch

from . import _lib, device as dv
from . import spharmORFbasis as anis
from ._compat import TimeDelta, u
from ._position import ra_dec
from .constants import DAY_IN_SEC, YEAR_IN_SEC


# ------------------------------------------------------------------------------------------------
# red noise
# ------------------------------------------------------------------------------------------------
def _fourier_frequencies(toas, nmodes, Tspan, logf, fmin, fmax, modes):
    """sampling frequencies exactly as red_noise.py:61-80 builds them (host, O(nmodes))."""
    T = Tspan if Tspan is not None else toas.max() - toas.min()
    if modes is not None:
        return np.asarray(modes, dtype=np.float64)
    if fmin is None and fmax is None and not logf:
        return 1.0 * np.arange(1, nmodes + 1) / T
    if fmin is None:
        fmin = 1 / T
    if fmax is None:
        fmax = nmodes / T
    return np.logspace(np.log10(fmin), np.log10(fmax), nmodes) if logf else np.linspace(fmin, fmax, nmodes)


def _design_matrix_device(toas, f, ranphase, libstempo_convention):
    """Ft [2*nmodes, N] on the device (pta_rn_basis)."""
    n, nm = len(toas), len(f)
    t_d, f_d = dv.f64(toas), dv.f64(f)
    ph_d = dv.f64(ranphase) if ranphase is not None else None
    Ft = dv.empty((2 * nm, n))
    t_ref = float(toas[0]) if libstempo_convention else 0.0
    _lib.call("pta_rn_basis", dv.ptr(t_d), n, ctypes.c_double(t_ref), dv.ptr(f_d), dv.ptr(ph_d), nm,
              1 if libstempo_convention else 0, dv.ptr(Ft), n, dv.stream_ptr())
    return Ft


def create_fourier_design_matrix_red(toas, nmodes=30, Tspan=None, logf=False, fmin=None, fmax=None, pshift=False,
                                     libstempo_convention=False, modes=None):
    """Fourier design matrix (Lentati et al. 2013, eq. 11); same arguments and return as red_noise.py:36-103.

    Returns (F [N, 2*nmodes], Ffreqs [2*nmodes]) as NumPy arrays; the sin/cos evaluation runs on the GPU.
    """
    toas = np.asarray(toas, dtype=np.float64)
    f = _fourier_frequencies(toas, nmodes, Tspan, logf, fmin, fmax, modes)
    nmodes = len(f)
    ranphase = np.random.uniform(0.0, 2 * np.pi, nmodes) if pshift else None  # red_noise.py:83-84
    Ft = _design_matrix_device(toas, f, ranphase, libstempo_convention)
    return Ft.T.contiguous().cpu().numpy(), np.repeat(f, 2)


def _red_noise_inputs(psr, log10_amplitude, spectral_index, components, modes):
    """toas [s], sampling frequencies and sqrt(prior) of one pulsar, with the reference's expressions (red_noise.py:112-126)."""
    A = 10 ** log10_amplitude
    gamma = spectral_index
    fyr = 1 / YEAR_IN_SEC
    toas = np.array(psr.toas.table["tdbld"], dtype="float64") * DAY_IN_SEC
    Tspan = toas.max() - toas.min()
    f = _fourier_frequencies(toas, components, Tspan, False, None, None, modes)
    freqs = np.repeat(f, 2)
    prior = A ** 2 * (freqs / fyr) ** (-gamma) / (12 * np.pi ** 2 * Tspan) * YEAR_IN_SEC ** 3
    return toas, f, np.sqrt(prior)


def _record_red_noise(psr, log10_amplitude, spectral_index, dt):
    psr.update_added_signals("{}_red_noise".format(psr.name),
                             {"amplitude": log10_amplitude, "spectral_index": spectral_index}, dt)
    psr.toas.adjust_TOAs(TimeDelta(dt.to("day")))
    psr.update_residuals()


def add_red_noise(psr, log10_amplitude, spectral_index, components=30, seed=None, modes=None, Tspan=None,
                  libstempo_convention=False):
    """Add red noise with P(f) = A^2 / (12 pi^2) (f year)^-gamma using `components` Fourier bases
    (red_noise.py:106-135).  The `Tspan` argument is ignored, as in the reference (:124).

    ``psr`` may be a LIST of pulsars with per-pulsar ``log10_amplitude`` / ``spectral_index`` / ``seed`` lists (an amplitude of
    None skips that pulsar): the loop of single calls with one upload, one download and no synchronisation in between."""
    if isinstance(psr, (list, tuple)):
        return _add_red_noise_list(list(psr), log10_amplitude, spectral_index, components, seed, modes, libstempo_convention)
    if seed is not None:
        np.random.seed(seed)
    if modes is not None:
        print("Must use linear spacing.")
    toas, f, amp = _red_noise_inputs(psr, log10_amplitude, spectral_index, components, modes)
    y = amp * np.random.randn(amp.size)
    n, nm, K = len(toas), len(f), amp.size
    t_d, f_d, y_d = dv.upload_packed([toas, f, y])
    Ft = dv.empty((K, n))
    t_ref = float(toas[0]) if libstempo_convention else 0.0
    s = dv.stream_ptr()
    _lib.call("pta_rn_basis", dv.ptr(t_d), n, ctypes.c_double(t_ref), dv.ptr(f_d), None, nm, 1 if libstempo_convention else 0, dv.ptr(Ft), n, s)
    out = dv.empty((1, n))
    _lib.call("pta_rn_synth", dv.ptr(Ft), n, n, K, dv.ptr(y_d), K, 1, dv.ptr(out), n, 0, s)
    _record_red_noise(psr, log10_amplitude, spectral_index, dv.download(out[0]) * u.s)


def _add_red_noise_list(psrs, log10_amplitude, spectral_index, components, seed, modes, libstempo_convention):
    P = len(psrs)
    lA = list(log10_amplitude) if isinstance(log10_amplitude, (list, tuple, np.ndarray)) else [log10_amplitude] * P
    gm = list(spectral_index) if isinstance(spectral_index, (list, tuple, np.ndarray)) else [spectral_index] * P
    seeds = None if seed is None else list(seed)
    if len(lA) != P or len(gm) != P or (seeds is not None and len(seeds) != P):
        raise ValueError("log10_amplitude / spectral_index / seed must be scalars or one entry per pulsar")
    if modes is not None:
        print("Must use linear spacing.")
    live = [a for a in range(P) if lA[a] is not None and gm[a] is not None]
    inp = [_red_noise_inputs(psrs[a], lA[a], gm[a], components, modes) for a in live]
    if seeds is None:
        ys = [amp * np.random.randn(amp.size) for (_, _, amp) in inp]
    else:   # every pulsar re-seeds: its 2 * components draws are the head of the stream np.random.seed(seed_a) starts
        from .white_noise import _legacy_normals  # native restatement of the legacy stream, the pulsars' streams on host threads
        zs = _legacy_normals([seeds[a] for a in live], [[amp.size] for (_, _, amp) in inp]) if live else []
        ys = [amp * z[0] for (_, _, amp), z in zip(inp, zs)]
    counts = [len(t) for (t, _, _) in inp]
    off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    dev = dv.upload_packed([np.concatenate([t for (t, _, _) in inp])] + [x for (_, f, _) in inp for x in (f,)] + ys) if live else []
    t_all = dev[0] if live else None
    out = dv.empty((1, int(off[-1]))) if live else None
    s = dv.stream_ptr()
    keep = []
    for k, a in enumerate(live):
        toas, f, amp = inp[k]
        n, nm, K = len(toas), len(f), amp.size
        f_d, y_d = dev[1 + k], dev[1 + len(live) + k]
        Ft = dv.empty((K, n))
        keep.append(Ft)
        t_ref = float(toas[0]) if libstempo_convention else 0.0
        tp = ctypes.c_void_p(t_all.data_ptr() + 8 * int(off[k]))
        _lib.call("pta_rn_basis", tp, n, ctypes.c_double(t_ref), dv.ptr(f_d), None, nm, 1 if libstempo_convention else 0, dv.ptr(Ft), n, s)
        _lib.call("pta_rn_synth", dv.ptr(Ft), n, n, K, dv.ptr(y_d), K, 1, ctypes.c_void_p(out.data_ptr() + 8 * int(off[k])), n, 0, s)
    if live:
        res = dv.download(out[0])
        for k, a in enumerate(live):
            _record_red_noise(psrs[a], lA[a], gm[a], res[off[k]:off[k + 1]] * u.s)


# ------------------------------------------------------------------------------------------------
# GWB
# ------------------------------------------------------------------------------------------------
def gwb_time_grid(psrs, npts=600, howml=10):
    """start/stop/dur, coarse grid, the quirky dt = dur/npts and the DC..Nyquist frequency grid, computed with
    the very NumPy expressions of red_noise.py:182-197,230-232 - len(f) is a knife-edge."""
    start = float(np.min([psr.toas.first_MJD.value * 86400 for psr in psrs]) - 86400)
    stop = float(np.max([psr.toas.last_MJD.value * 86400 for psr in psrs]) + 86400)
    dur = stop - start
    if npts is None:
        npts = dur / (86400 * 14)
    ut = np.linspace(start, stop, npts)
    dt = dur / npts
    f = np.arange(0, 1 / (2 * dt), 1 / (dur * howml))
    f[0] = f[1]
    return dict(start=start, stop=stop, dur=dur, npts=npts, ut=ut, dt=dt, f=f, Nf=len(f), howml=howml)


def gwb_spectrum(f, dur, howml, log10_amplitude, spectral_index, turnover=False, f0=1e-9, beta=1, power=1,
                 userSpec=None):
    """C(f) = hc(f)^2 / (96 pi^2 f^3) * dur * howml (red_noise.py:243-265)."""
    if userSpec is None:
        Amp = 10 ** log10_amplitude
        gam = spectral_index
        f1yr = 1 / 3.16e7
        alpha = -0.5 * (gam - 3)
        hcf = Amp * (f / f1yr) ** (alpha)
        if turnover:
            si = alpha - beta
            hcf /= (1 + (f / f0) ** (power * si)) ** (1 / power)
    else:
        freqs = userSpec[:, 0]
        if len(userSpec[:, 0]) != len(freqs):
            raise ValueError("Number of supplied spectral points does not match number of frequencies!")
        # log-log linear interpolation, flat outside the supplied band (interp1d + extrap1d, :11-33,261-263);
        # numpy.interp is what scipy's interp1d(kind='linear') delegates to, and clamps at the ends.  interp1d
        # (assume_sorted=False) first sorts the supplied points by frequency (stable mergesort) - numpy.interp would
        # silently return garbage for an unsorted spectrum - so do the same
        order = np.argsort(freqs, kind="mergesort")
        hcf = 10.0 ** np.interp(np.log10(f), np.log10(freqs[order]), np.log10(userSpec[order, 1]))
    return 1 / 96 / np.pi ** 2 * hcf ** 2 / f ** 3 * dur * howml


def gwb_orf_device(psrs, no_correlations=False, clm=(np.sqrt(4.0 * np.pi),), lmax=0):
    """ORF [P, P] as a device tensor (red_noise.py:200-226)."""
    P = len(psrs)
    if no_correlations:
        return dv.f64(np.diag(np.ones(P) * 2))
    psrlocs = np.zeros((P, 2))
    for ii in range(P):
        psrlocs[ii] = ra_dec(psrs[ii], default=(0.0, 0.0))   # no location: stays (0, 0), like red_noise.py:203
    psrlocs[:, 1] = np.pi / 2.0 - psrlocs[:, 1]
    return anis.orf_from_locations(psrlocs, clm, lmax)


def cholesky_device(A, flags=0, auto_substitution=True):
    """lower Cholesky factor of a [n,n] (or [B,n,n]) device tensor, in place; raises numpy's LinAlgError
    like np.linalg.cholesky (red_noise.py:235) when a matrix is not positive definite.  `flags`: extra PTA_POTRF_* bits
    (e.g. _lib.POTRF_VALU for the all-VALU cross-check path).  `auto_substitution=False` leaves the choice of the panel solve to
    `flags` alone (tests of the MFMA product form)."""
    batched = A.dim() == 3
    B, n = (A.shape[0], A.shape[1]) if batched else (1, A.shape[0])
    info = dv.zeros((B,), dtype=torch.int32)
    # P x P ORF factors (a few hundred rows at most): forward-substitution panel solves.  The MFMA product with the inverted
    # diagonal block brings cond(L11) * eps into the backward error - harmless for the well-conditioned HD matrix, but a
    # user-supplied anisotropic ORF (clm, lmax > 0) may be close to singular and must still match np.linalg.cholesky at 1e-10
    # (ADVICE r2); the cost is negligible at this size, the product form stays for the large TD factors
    if auto_substitution and n <= 2048 and not (int(flags) & _lib.POTRF_VALU):
        flags = int(flags) | _lib.POTRF_SUBSTITUTION
    _lib.call("pta_potrf_batched_ex", dv.ptr(A), n, n, n * n, B, dv.ptr(info), _lib.POTRF_ZERO_UPPER | int(flags), dv.stream_ptr())
    bad = info.cpu().numpy()
    if np.any(bad != 0):
        raise np.linalg.LinAlgError("Matrix is not positive definite")
    return A


def pad16(n):
    return (int(n) + 15) // 16 * 16


def add_gwb(psrs, log10_amplitude, spectral_index, no_correlations=False, seed=None, turnover=False,
            clm=[np.sqrt(4.0 * np.pi)], lmax=0, f0=1e-9, beta=1, power=1, userSpec=None, npts=600, howml=10):
    """Inject a stochastic GWB (Chamberlin et al. 2014 construction) into a list of pulsars; same arguments as
    red_noise.py:138-153."""
    if seed is not None:
        np.random.seed(seed)
    Npulsars = len(psrs)
    grid = gwb_time_grid(psrs, npts, howml)
    npts, Nf, dt = grid["npts"], grid["Nf"], grid["dt"]

    ORF = gwb_orf_device(psrs, no_correlations, clm, lmax)
    M = cholesky_device(ORF)

    # draws in the reference's order: per pulsar Nf real parts then Nf imaginary parts (:238-240)
    w = np.empty((Npulsars, Nf, 2))
    for ll in range(Npulsars):
        w[ll, :, 0] = np.random.randn(Nf)
        w[ll, :, 1] = np.random.randn(Nf)

    C = gwb_spectrum(grid["f"], grid["dur"], howml, log10_amplitude, spectral_index, turnover, f0, beta, power, userSpec)

    ldt = pad16(npts)
    T = dv.empty((2 * (Nf - 2), ldt))
    # every operand of the call in ONE pinned staging copy (device.upload_packed), the result in one
    toa_s = [psr.toas.get_mjds().value.astype(float) * 86400 for psr in psrs]
    counts = [len(t) for t in toa_s]
    ntot = int(np.sum(counts))
    sqrtC, w_d, toa_d, ut_d, psr_of = dv.upload_packed([C ** 0.5, w, np.concatenate(toa_s), grid["ut"],
                                                        np.repeat(np.arange(Npulsars, dtype=np.int32), counts)])
    s = dv.stream_ptr()
    _lib.call("pta_gwb_twiddle", dv.ptr(sqrtC), Nf, npts, 10, ctypes.c_double(1.0 / dt), dv.ptr(T), ldt, s)
    G0 = dv.empty((Npulsars, npts))
    _lib.call("pta_gwb_idft", dv.ptr(w_d), 2 * Nf, Npulsars, Nf, dv.ptr(T), ldt, npts, dv.ptr(G0), npts, 1, s)
    G = dv.empty((Npulsars, npts))
    _lib.call("pta_gwb_mix", dv.ptr(M), Npulsars, dv.ptr(G0), 1, npts, npts, dv.ptr(G), 0, s)

    jlo = dv.empty((ntot,), dtype=torch.int32)
    _lib.call("pta_gwb_bracket", dv.ptr(ut_d), npts, dv.ptr(toa_d), ntot, dv.ptr(jlo), s)
    out = dv.empty((1, ntot))
    _lib.call("pta_gwb_interp", dv.ptr(G), npts, Npulsars, npts, dv.ptr(ut_d), dv.ptr(toa_d), dv.ptr(psr_of), dv.ptr(jlo),
              ntot, 1, ctypes.c_double(1.0), dv.ptr(out), ntot, 0, s)
    res_all = dv.download(out[0])
    res_gw = np.split(res_all, np.cumsum(counts)[:-1])

    ct = 0
    for psr in psrs:
        dt_ = res_gw[ct] / 86400.0 * u.day  # the reference stores this signal in days (:292)
        psr.toas.adjust_TOAs(TimeDelta(dt_.to("day")))
        psr.update_added_signals("{}_gwb".format(psr.name),
                                 {"amplitude": log10_amplitude, "spectral_index": spectral_index}, dt_)
        psr.update_residuals()
        ct += 1
