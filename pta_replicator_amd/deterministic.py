"""Continuous-wave injection on the MI355X: ``add_cgw`` and ``add_catalog_of_cws`` with the reference's signatures.

Mirrors ``pta_replicator/deterministic.py:13-185`` (single source) and ``:188-561`` (catalogue + its two numba
kernels).  For a single source the scalar prefactors (antenna patterns, chirp factors) are computed on the host with
the reference's own expressions and the per-TOA waveform (four pow, two sincos) runs in the ``pta_cgw`` kernel; for a
catalogue everything, prefactors included, runs on the device (``pta_cw_catalog``).  The population / burst / memory
injectors of the reference remain outside this round's scope.
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
    src_d, mjd_d = dv.f64(np.stack(lists, axis=1)), dv.f64(mjd)
    npar, npart, nchunk = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int(0)
    _lib.call("pta_cw_catalog_workspace", n, ncw, ctypes.byref(npar), ctypes.byref(npart), ctypes.byref(nchunk))
    par_ws, part_ws, out = dv.empty((npar.value,)), dv.empty((npart.value,)), dv.empty((n,))
    consts = np.array([SOLAR2S, KPC2S, MPC2S], dtype=np.float64)
    mode = 0 if evolve else (1 if phase_approx else 2)
    _lib.call("pta_cw_catalog", dv.ptr(mjd_d), n, dv.ptr(src_d), ncw, dv.hptr(phat), dv.hptr(consts), ctypes.c_double(pdist),
              0 if pphase is None else 1, ctypes.c_double(0.0 if pphase is None else pphase), 1 if psrTerm else 0, mode,
              ctypes.c_double(tref), dv.ptr(par_ws), dv.ptr(part_ws), dv.ptr(out), 0, dv.stream_ptr())
    dt = out.cpu().numpy() * u.s
    psr.update_added_signals("{}_".format(psr.name) + signal_name,
                             {"gwtheta_list": gwtheta_list, "gwphi_list": gwphi_list, "mc_list": mc_list, "dist_list": dist_list,
                              "fgw_list": fgw_list, "phase0_list": phase0_list, "psi_list": psi_list, "inc_list": inc_list,
                              "pdist": pdist, "pphase": pphase, "psrTerm": psrTerm, "evolve": evolve, "phase_approx": phase_approx,
                              "tref": tref}, dt)
    psr.toas.adjust_TOAs(TimeDelta(dt.to("day")))
    psr.update_residuals()
