"""Continuous-wave injection on the MI355X: ``add_cgw`` and ``add_catalog_of_cws`` with the reference's signatures.

Mirrors ``pta_replicator/deterministic.py:13-185`` (single source) and ``:188-561`` (catalogue + its two numba
kernels).  For a single source the scalar prefactors (antenna patterns, chirp factors) are computed on the host with
the reference's own expressions and the per-TOA waveform (four pow, two sincos) runs in the ``pta_cgw`` kernel; for a
catalogue everything, prefactors included, runs on the device (``pta_cw_catalog``).  ``add_gwb_plus_outlier_cws``
(``:565-715``) is host bookkeeping around ``add_gwb`` and the catalogue kernel.  ``add_burst`` / ``add_noise_transient`` /
``add_gw_memory`` (``:718-884``) evaluate USER-SUPPLIED Python callables (or a one-line ramp) per TOA: host work by
construction - there is no device path for them to fall back from - kept so that a reference script finds every name.
"""
import ctypes

import numpy as np

from . import _lib, device as dv
from ._compat import TimeDelta, u
from ._position import ra_dec
from .constants import KPC2S, MPC2S, SOLAR2S

CGW_NPAR = 18


def cgw_parameters(ptheta, pphi, gwtheta, gwphi, mc, dist, fgw, phase0, psi, inc, pdist=1.0, pphase=None,
                   psrTerm=True, evolve=True, phase_approx=False, tref=0):
    """The 18 scalars of pta_cgw (include/pta_replicator_amd.h), following deterministic.py:51-109 line by line.
    Returns (par, mc_seconds, dist_seconds, phase0_orbital) - the last three are what the reference records."""
    mc *= SOLAR2S
    dist *= MPC2S
    w0 = np.pi * fgw
    phase0 /= 2
    w053 = w0 ** (-5 / 3)
    cosgwtheta, cosgwphi = np.cos(gwtheta), np.cos(gwphi)
    singwtheta, singwphi = np.sin(gwtheta), np.sin(gwphi)
    sin2psi, cos2psi = np.sin(2 * psi), np.cos(2 * psi)
    incfac1, incfac2 = 0.5 * (3 + np.cos(2 * inc)), 2 * np.cos(inc)
    m = np.array([singwphi, -cosgwphi, 0.0])
    n = np.array([-cosgwtheta * cosgwphi, -cosgwtheta * singwphi, singwtheta])
    omhat = np.array([-singwtheta * cosgwphi, -singwtheta * singwphi, -cosgwtheta])
    fac1 = 256 / 5 * mc ** (5 / 3) * w0 ** (8 / 3)
    fac2 = 1 / 32 / mc ** (5 / 3)
    fac3 = mc ** (5 / 3) / dist
    phat = np.array([np.sin(ptheta) * np.cos(pphi), np.sin(ptheta) * np.sin(pphi), np.cos(ptheta)])
    fplus = 0.5 * (np.dot(m, phat) ** 2 - np.dot(n, phat) ** 2) / (1 + np.dot(omhat, phat))
    fcross = (np.dot(m, phat) * np.dot(n, phat)) / (1 + np.dot(omhat, phat))
    cosMu = -np.dot(omhat, phat)
    if pphase is not None:
        pd = pphase / (2 * np.pi * fgw * (1 - cosMu)) / KPC2S
    else:
        pd = pdist
    pd *= KPC2S
    par = np.zeros(CGW_NPAR)
    par[:14] = [tref, w0, phase0, w053, fac1, fac2, fac3, incfac1, incfac2, cos2psi, sin2psi, fplus, fcross,
                pd * (1 - cosMu)]
    par[14] = 0 if evolve else (1 if phase_approx else 2)
    par[15] = 1 if psrTerm else 0
    if (not evolve) and phase_approx:
        omega_p = w0 * (1 + fac1 * pd * (1 - cosMu)) ** (-3 / 8)
        par[16] = omega_p
        par[17] = phase0 + fac2 * (w053 - omega_p ** (-5 / 3))
    return par, mc, dist, phase0


def cgw_delay_device(mjd, par):
    """res[N] (seconds) as a device tensor for float64 MJDs."""
    mjd_d = dv.f64(mjd)
    n = mjd_d.shape[0]
    out = dv.empty((n,))
    par = np.ascontiguousarray(par, dtype=np.float64)
    _lib.call("pta_cgw", dv.ptr(mjd_d), n, dv.hptr(par), dv.ptr(out), 0, dv.stream_ptr())
    return out


def add_cgw(psr, gwtheta, gwphi, mc, dist, fgw, phase0, psi, inc, pdist=1.0, pphase=None, psrTerm=True, evolve=True,
            phase_approx=False, tref=0, signal_name="cw"):
    """Add continuous-wave residuals (adapted from libstempo.toasim by the reference); arguments as in
    deterministic.py:13-48: angles [rad], mc [Msun], dist [Mpc], fgw [Hz], pdist [kpc], tref [s]."""
    ra, dec = ra_dec(psr)
    ptheta = np.pi / 2 - dec
    pphi = ra
    par, mc_s, dist_s, phase0_orb = cgw_parameters(ptheta, pphi, gwtheta, gwphi, mc, dist, fgw, phase0, psi, inc, pdist,
                                                   pphase, psrTerm, evolve, phase_approx, tref)
    mjd = np.asarray(psr.toas.get_mjds().value, dtype=np.float64)
    res = cgw_delay_device(mjd, par).cpu().numpy()
    dt = res * u.s
    # the reference records the unit-converted values (it rebinds mc, dist, phase0 before this point)
    psr.update_added_signals("{}_".format(psr.name) + signal_name,
                             {"gwtheta": gwtheta, "gwphi": gwphi, "mc": mc_s, "dist": dist_s, "fgw": fgw,
                              "phase0": phase0_orb, "psi": psi, "inc": inc, "pdist": pdist, "pphase": pphase,
                              "psrTerm": psrTerm, "evolve": evolve, "phase_approx": phase_approx, "tref": tref}, dt)
    psr.toas.adjust_TOAs(TimeDelta(dt.to("day")))
    psr.update_residuals()


CW_NPAR = 16              # PTA_CW_NPAR


def _dot3(X, y):
    """row-wise X[i] . y for an [n, 3] array, bit-identical to ``np.dot(X[i], y)`` - the call the reference's loop body makes
    (deterministic.py:364-372).  np.dot on two 3-vectors is OpenBLAS' ddot, which sums fma(x2 y2, fma(x1 y1, x0 y0)); the native
    helper pta_dot3_host reproduces that association without a Python-level loop.  Because the association belongs to the BLAS
    build, it is CHECKED against np.dot on a sample of rows of every call (the first / last 32 and 64 spread between); on a host
    whose BLAS sums differently the rows are evaluated one np.dot at a time, as before."""
    X = np.ascontiguousarray(X, dtype=np.float64)
    y = np.ascontiguousarray(y, dtype=np.float64)
    n = len(X)
    out = np.empty(n)
    _lib.call("pta_dot3_host", dv.hptr(X), n, dv.hptr(y), dv.hptr(out))
    probe = np.unique(np.concatenate([np.arange(min(n, 32)), np.arange(max(n - 32, 0), n), np.linspace(0, n - 1, 64).astype(np.int64)])) if n else []
    for i in probe:
        if out[i] != np.dot(X[i], y) and not (np.isnan(out[i]) and np.isnan(np.dot(X[i], y))):
            return np.array([np.dot(X[i], y) for i in range(n)])
    return out


def _pow(x, y):
    """elementwise x ** y through libm's scalar pow (pta_pow_host): what ``np.float64 ** float`` evaluates to in the reference's
    loop body.  NumPy's array power uses SIMD kernels that differ from libm by an ulp on ~5 % of arguments."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty_like(x)
    _lib.call("pta_pow_host", dv.hptr(x), ctypes.c_double(y), x.size, dv.hptr(out))
    return out


def cw_source_params(lists, phat, pdist=1.0, pphase=None):
    """The realisation- and TOA-independent scalars of every source of a CW catalogue, as deterministic.py:331-383 computes
    them at the top of its numba loops: [ncw, 16] = w0, orbital phase0, w0^(-5/3), fac1, fac2, fac3, incfac1, incfac2, cos 2psi,
    sin 2psi, F+, Fx, pd (1 - cos mu) [s], and for phase_approx omega_p and its phase offset (:395,:399).

    Computed on the HOST, like add_cgw's (cgw_parameters): the pulsar-term phase is omega * pd * (1 - cos mu) with pd ~ 1e11 s, so
    one ulp of cos mu moves a fast binary's phase by up to 4e-11 rad - device sin / cos of the source angles (1-2 ulp) were the
    largest parity error of the catalogue.  ONE code path for every catalogue size (a per-source Python loop up to
    20 000 sources and array expressions beyond made the results jump by 7e-12 at 20 001 and cost 0.7 s per pulsar below it):
    the elementwise expressions are NumPy array operations - the same ufunc inner loops a NumPy float64 scalar goes through in the
    reference's loop body, element for element -, the powers go through libm's scalar pow (_pow) and the three np.dot calls per
    source, whose association is BLAS', through _dot3.  Bit-identical to the per-source scalar loop (tests/test_host_logic.py)."""
    gwtheta, gwphi, mc, dist, fgw, phase0, psi, inc = lists
    ncw = len(mc)
    par = np.zeros((ncw, CW_NPAR))
    if ncw == 0:
        return par
    with np.errstate(all="ignore"):
        mcs, ds = mc * SOLAR2S, dist * MPC2S
        w0 = np.pi * fgw
        w053 = _pow(w0, -5 / 3)
        cgt, cgp, sgt, sgp = np.cos(gwtheta), np.cos(gwphi), np.sin(gwtheta), np.sin(gwphi)
        mp = _dot3(np.stack([sgp, -cgp, np.zeros(ncw)], axis=1), phat)                  # np.dot(m, phat)
        npd = _dot3(np.stack([-cgt * cgp, -cgt * sgp, sgt], axis=1), phat)              # np.dot(n, phat)
        op = _dot3(np.stack([-sgt * cgp, -sgt * sgp, -cgt], axis=1), phat)              # np.dot(omhat, phat)
        mc53 = _pow(mcs, 5 / 3)
        fac1 = 256 / 5 * mc53 * _pow(w0, 8 / 3)
        fac2 = 1 / 32 / mc53
        fac3 = mc53 / ds
        cosMu = -op
        pd = (pphase / (2 * np.pi * fgw * (1 - cosMu)) / KPC2S if pphase is not None else pdist * np.ones(ncw)) * KPC2S
        omega_p = w0 * _pow(1 + fac1 * pd * (1 - cosMu), -3 / 8)
        cols = (w0, phase0 / 2, w053, fac1, fac2, fac3, 0.5 * (3 + np.cos(2 * inc)), 2 * np.cos(inc), np.cos(2 * psi), np.sin(2 * psi),
                0.5 * (_pow(mp, 2.0) - _pow(npd, 2.0)) / (1 + op), (mp * npd) / (1 + op), pd * (1 - cosMu), omega_p,
                phase0 / 2 + fac2 * (w053 - _pow(omega_p, -5 / 3)), np.zeros(ncw))
        for k, c in enumerate(cols):
            par[:, k] = c
    return par


def add_catalog_of_cws(psr, gwtheta_list, gwphi_list, mc_list, dist_list, fgw_list, phase0_list, psi_list, inc_list, pdist=1.0,
                       pphase=None, psrTerm=True, evolve=True, phase_approx=False, tref=0, chunk_size=10_000_000,
                       signal_name="cw_catalog"):
    """Add many SMBHB continuous-wave sources at once; same arguments as deterministic.py:188-229 (arrays of source
    parameters, one entry per binary).  ``chunk_size`` is accepted for compatibility: the device sums the whole
    catalogue in one pass (the reference re-registers the signal for every chunk, which raises on the second one)."""
    ra, dec = ra_dec(psr)
    ptheta, pphi = np.pi / 2 - dec, ra
    phat = np.array([np.sin(ptheta) * np.cos(pphi), np.sin(ptheta) * np.sin(pphi), np.cos(ptheta)], dtype=np.float64)
    lists = [np.asarray(x, dtype=np.float64).ravel() for x in (gwtheta_list, gwphi_list, mc_list, dist_list, fgw_list,
                                                                  phase0_list, psi_list, inc_list)]
    ncw = len(lists[0])
    if any(len(x) != ncw for x in lists):
        raise ValueError("all source parameter lists must have the same length")
    mjd = np.asarray(psr.toas.get_mjds().value, dtype=np.float64)
    n = len(mjd)
    par = cw_source_params(lists, phat, pdist, pphase)
    par_d, mjd_d = dv.f64(par), dv.f64(mjd)
    npart, nchunk = ctypes.c_int64(0), ctypes.c_int(0)
    _lib.call("pta_cw_catalog_workspace", n, ncw, ctypes.byref(npart), ctypes.byref(nchunk))
    part_ws, out = dv.empty((npart.value,)), dv.empty((n,))
    mode = 0 if evolve else (1 if phase_approx else 2)
    _lib.call("pta_cw_catalog", dv.ptr(mjd_d), n, dv.ptr(par_d), ncw, 1 if psrTerm else 0, mode, ctypes.c_double(tref), dv.ptr(part_ws),
              dv.ptr(out), 0, dv.stream_ptr())
    dt = out.cpu().numpy() * u.s
    psr.update_added_signals("{}_".format(psr.name) + signal_name,
                             {"gwtheta_list": gwtheta_list, "gwphi_list": gwphi_list, "mc_list": mc_list, "dist_list": dist_list,
                              "fgw_list": fgw_list, "phase0_list": phase0_list, "psi_list": psi_list, "inc_list": inc_list,
                              "pdist": pdist, "pphase": pphase, "psrTerm": psrTerm, "evolve": evolve, "phase_approx": phase_approx,
                              "tref": tref}, dt)
    psr.toas.adjust_TOAs(TimeDelta(dt.to("day")))
    psr.update_residuals()


def split_population(vals, weights, fobs, T_obs, outlier_per_bin=100, population=None):
    """Host bookkeeping of deterministic.py:617-676: characteristic strain of every binary, the ``outlier_per_bin`` loudest
    of each frequency bin set aside (padded slots stay zero), the remainder summed into the free spectrum.
    Returns (f_centers, free_spec, outlier_hs, outlier_fo, outlier_mc [Msun, observer frame], outlier_dl [Mpc])."""
    from . import _population as pop
    population = population or pop.Population
    vals = np.asarray(vals, dtype=np.float64)
    weights = np.asarray(weights, dtype=np.float64)
    fobs = np.asarray(fobs, dtype=np.float64)

    f_centers = np.array([(fobs[i + 1] + fobs[i]) / 2 for i in range(fobs.size - 1)])
    mc = population.chirp_mass(*population.component_masses(vals[0], vals[1]))   # rest frame
    rz = vals[2, :]
    frst = vals[3] * (1.0 + rz)
    dc = population.comoving_distance_cm(rz)
    dl = np.copy(dc) * (1.0 + rz)
    hs = population.gw_strain_source(mc, dc, frst / 2)
    fo = vals[-1]
    mc = mc * (1.0 + rz)                                                          # observer frame for the injections

    nbin = fobs.shape[0] - 1
    freq_idxs = np.digitize(fo, fobs)
    free_spec = np.ones(nbin) * 1e-100
    outlier_hs, outlier_fo, outlier_mc, outlier_dl = (np.zeros(nbin * outlier_per_bin) for _ in range(4))
    weighted_h_square = weights * hs ** 2 * fo * T_obs
    for k in range(nbin):
        members = np.nonzero((freq_idxs - 1) == k)[0]
        order = members[np.argsort(weighted_h_square[members])[::-1]]            # loudest first, ties as the reference
        top = order[:outlier_per_bin]
        sl = slice(outlier_per_bin * k, outlier_per_bin * k + len(top))
        outlier_hs[sl] = weighted_h_square[top]
        outlier_fo[sl] = fo[top]
        outlier_mc[sl] = mc[top] / pop.MSOL_CGS
        outlier_dl[sl] = dl[top] / pop.PC_CGS / 1e6
        free_spec[k] += np.sum(weighted_h_square[order[outlier_per_bin:]])
    return f_centers, free_spec, outlier_hs, outlier_fo, outlier_mc, outlier_dl


def add_gwb_plus_outlier_cws(psrs, vals, weights, fobs, T_obs, outlier_per_bin=100, seed=None, population=None):
    """Realistic data set from a binned SMBHB population: the ``outlier_per_bin`` loudest binaries of every frequency bin
    are injected one by one (add_catalog_of_cws, one device pass per pulsar), the rest as a GWB with their summed spectrum
    (add_gwb(userSpec=...)).  Arguments and the eleven return values as deterministic.py:565-715; ``population`` swaps the
    holodeck-equivalent helpers (see _population.py)."""
    from .red_noise import add_gwb
    f_centers, free_spec, outlier_hs, outlier_fo, outlier_mc, outlier_dl = split_population(vals, weights, fobs, T_obs,
                                                                                            outlier_per_bin, population)
    FreeSpec = np.array([f_centers, np.sqrt(free_spec)]).T
    add_gwb(psrs, None, None, userSpec=FreeSpec, howml=10, seed=seed)

    outlier_hs = outlier_hs[np.where(outlier_hs > 0)]
    outlier_fo = outlier_fo[np.where(outlier_fo > 0)]
    outlier_mc = outlier_mc[np.where(outlier_mc > 0)]
    outlier_dl = outlier_dl[np.where(outlier_dl > 0)]
    N_CW = outlier_hs.shape[0]
    # the global legacy stream continues where add_gwb left it (:696-700)
    random_gwthetas = np.arccos(np.random.uniform(low=-1.0, high=1.0, size=N_CW))
    random_gwphis = np.random.uniform(low=0.0, high=2 * np.pi, size=N_CW)
    random_phases = np.random.uniform(low=0.0, high=2 * np.pi, size=N_CW)
    random_psis = np.random.uniform(low=0.0, high=np.pi, size=N_CW)
    random_incs = np.arccos(np.random.uniform(low=-1.0, high=1.0, size=N_CW))

    for pulsar in psrs:
        add_catalog_of_cws(pulsar, gwtheta_list=random_gwthetas, gwphi_list=random_gwphis, mc_list=outlier_mc,
                           dist_list=outlier_dl, fgw_list=outlier_fo, phase0_list=random_phases, psi_list=random_psis,
                           inc_list=random_incs, pdist=1.0, pphase=None, psrTerm=True, evolve=True, phase_approx=False,
                           tref=53000 * 86400)
    return (f_centers, free_spec, outlier_fo, outlier_hs, outlier_mc, outlier_dl, random_gwthetas, random_gwphis,
            random_phases, random_psis, random_incs)


def antenna_patterns(psr, gwtheta, gwphi):
    """(F+, Fx, cosMu) of a pulsar for a source at (gwtheta, gwphi) - Sesana et al. 2010 / Ellis et al. 2012 convention
    (deterministic.py:64-95, :732-761)."""
    ct, st, cp, sp = np.cos(gwtheta), np.sin(gwtheta), np.cos(gwphi), np.sin(gwphi)
    m = np.array([sp, -cp, 0.0])
    n = np.array([-ct * cp, -ct * sp, st])
    omhat = np.array([-st * cp, -st * sp, -ct])
    ra, dec = ra_dec(psr)
    ptheta, pphi = np.pi / 2 - dec, ra
    phat = np.array([np.sin(ptheta) * np.cos(pphi), np.sin(ptheta) * np.sin(pphi), np.cos(ptheta)])
    fplus = 0.5 * (np.dot(m, phat) ** 2 - np.dot(n, phat) ** 2) / (1 + np.dot(omhat, phat))
    fcross = (np.dot(m, phat) * np.dot(n, phat)) / (1 + np.dot(omhat, phat))
    return fplus, fcross, -np.dot(omhat, phat)


def _record(psr, signal_name, params, res):
    dt = np.asarray(res, dtype=np.float64) * u.s
    psr.update_added_signals("{}_".format(psr.name) + signal_name, params, dt)
    psr.toas.adjust_TOAs(TimeDelta(dt.to("day")))
    psr.update_residuals()


def add_burst(psr, gwtheta, gwphi, waveform_plus, waveform_cross, psi=0.0, tref=0, remove_quad=False, signal_name="burst"):
    """GW burst of arbitrary (callable) plus/cross waveforms, elliptical polarisation; arguments as deterministic.py:718-731.
    The callables receive ``t - tref`` in seconds.  Host only (see module docstring)."""
    fplus, fcross, _ = antenna_patterns(psr, gwtheta, gwphi)
    toas = psr.toas.get_mjds().value * 86400 - tref
    hplus, hcross = waveform_plus(toas), waveform_cross(toas)
    c2, s2 = np.cos(2 * psi), np.sin(2 * psi)
    res = -fplus * (hplus * c2 - hcross * s2) - fcross * (hplus * s2 + hcross * c2)
    if remove_quad:  # mimic the spin / spin-down fit (:778-780)
        pp = np.polyfit(np.array(toas, dtype=np.double), np.array(res, dtype=np.double), 2)
        res = res - pp[0] * toas ** 2 - pp[1] * toas - pp[2]
    _record(psr, signal_name, {"gwtheta": gwtheta, "gwphi": gwphi, "waveform_plus": waveform_plus, "waveform_cross": waveform_cross,
                               "psi": psi, "tref": tref, "remove_quad": remove_quad}, res)


def add_noise_transient(psr, waveform, tref=0, signal_name="noise_transient"):
    """Incoherent transient of arbitrary (callable) waveform in one pulsar (deterministic.py:796-819).  Host only."""
    toas = psr.toas.get_mjds().value * 86400 - tref
    _record(psr, signal_name, {"waveform": waveform, "tref": tref}, waveform(toas))


def add_gw_memory(psr, strain, gwtheta, gwphi, bwm_pol, t0_mjd, signal_name="gw_memory"):
    """Burst with memory: a ramp ``pol * strain * (t - t0)`` after the burst epoch (deterministic.py:822-884).  Host only."""
    fplus, fcross, _ = antenna_patterns(psr, gwtheta, gwphi)
    pol = np.cos(2 * bwm_pol) * fplus + np.sin(2 * bwm_pol) * fcross
    toas = psr.toas.get_mjds().value * 86400
    t0_sec = t0_mjd * 86400
    res = np.where(toas < t0_sec, 0.0, pol * strain * (toas - t0_sec))
    _record(psr, signal_name, {"strain": strain, "gwtheta": gwtheta, "gwphi": gwphi, "bwm_pol": bwm_pol, "t0_mjd": t0_mjd}, res)
