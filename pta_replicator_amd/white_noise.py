"""EFAC/EQUAD and ECORR injection on the MI355X: ``add_measurement_noise`` / ``add_jitter``.

Mirrors ``pta_replicator/white_noise.py`` (quantize_fast :7-44, add_measurement_noise :47-125, add_jitter
:128-198), same signatures, same ValueErrors, same ``added_signals`` keys.  Flag -> per-TOA vector expansion and
the (realisation-independent) epoch bucketing stay on the host - the latter in native code
(pta_quantize_epochs) - while the per-TOA arithmetic runs in HIP kernels.  ECORR is a gather through the epoch
map instead of the reference's dense N x E indicator matrix (17 s of its 18.6 s per realisation at 68 x 5000).
``add_efac`` / ``add_equad`` / ``add_ecorr`` are libstempo-style aliases.
"""
import ctypes

import numpy as np

from . import _lib, device as dv
from ._compat import TimeDelta, u


def epoch_map(times, dt=1.0):
    """Greedy bucketing of white_noise.py:21-31 -> (epoch_of int32[N], first_index int32[E]).

    ``epoch_of[i]`` is the column of the reference's U holding TOA i; ``first_index[e]`` is the earliest TOA of
    epoch e (its flag labels the epoch, :35).  Ties are ordered by np.argsort(times) exactly as the reference."""
    times = np.ascontiguousarray(times, dtype=np.float64)
    n = len(times)
    order = np.ascontiguousarray(np.argsort(times), dtype=np.int64)
    epoch_of = np.empty(n, dtype=np.int32)
    first = np.empty(n, dtype=np.int32)
    ne = ctypes.c_int(0)
    _lib.call("pta_quantize_epochs", dv.hptr(times), n, ctypes.c_double(dt), dv.hptr(order), dv.hptr(epoch_of), dv.hptr(first),
              ctypes.byref(ne))
    return epoch_of, first[:ne.value].copy()


def quantize_fast(times, flags=None, dt=1.0):
    """Same returns as the reference's quantize_fast (white_noise.py:7-44): (avetoas, [aveflags,] U).

    The dense U is materialised only for API compatibility; add_jitter itself never builds it."""
    times = np.asarray(times)
    epoch_of, first = epoch_map(times, dt)
    ne = len(first)
    avetoas = np.array([np.mean(times[epoch_of == e]) for e in range(ne)], "d")
    U = np.zeros((len(times), ne), "d")
    U[np.arange(len(times)), epoch_of] = 1
    if flags is not None:
        return avetoas, np.asarray(flags)[first], U
    return avetoas, U


def add_measurement_noise(psr, efac=1.0, log10_equad=None, flagid="f", flags=None, seed=None, tnequad=False):
    """Add EFAC/EQUAD white noise (white_noise.py:47-125): EFAC*(sigma z1 + EQUAD z2) by default,
    EFAC*sigma z1 + EQUAD z2 with ``tnequad``.  z2 is drawn even when EQUAD is zero, like the reference."""
    equad_str = "tnequad" if tnequad else "t2equad"
    if log10_equad is not None:
        equad = 10 ** log10_equad
    else:
        equad = 0.0
    if seed is not None:
        np.random.seed(seed)
    ntoas = psr.toas.ntoas
    efacvec = np.zeros(ntoas)
    equadvec = np.zeros(ntoas)
    if flags is None:
        if not np.isscalar(efac) or not np.isscalar(equad):
            raise ValueError("ERROR: If flags is None, efac and equad must be a scalar")
        efacvec = np.ones(ntoas) * efac
        equadvec = np.ones(ntoas) * equad
    if (flags is not None and not np.isscalar(efac)) or (flags is not None and not np.isscalar(equad)):
        if len(efac) == len(flags) and len(equad) == len(flags):
            toa_flags = np.array([f[flagid] for f in psr.toas.table["flags"].data])
            for ct, flag in enumerate(flags):
                ind = flag == toa_flags
                efacvec[ind] = efac[ct]
                equadvec[ind] = equad[ct]
        else:
            raise ValueError("ERROR: flags must be same length as efac and log10_equad")

    sigma = np.asarray(psr.toas.get_errors().to("s").value, dtype=np.float64)
    z1 = np.random.randn(ntoas)
    z2 = np.random.randn(ntoas)
    sig_d, ef_d, eq_d, z1_d, z2_d = dv.f64(sigma), dv.f64(efacvec), dv.f64(equadvec), dv.f64(z1), dv.f64(z2)
    out = dv.empty((1, ntoas))
    _lib.call("pta_wn", dv.ptr(sig_d), dv.ptr(ef_d), dv.ptr(eq_d), ntoas, 1 if tnequad else 0, dv.ptr(z1_d), dv.ptr(z2_d),
              ntoas, 1, dv.ptr(out), ntoas, 0, dv.stream_ptr())
    dt = out[0].cpu().numpy() * u.s

    if flags is None:
        psr.update_added_signals("{}_measurement_noise".format(psr.name),
                                 {"efac": efac, "log10_" + equad_str: log10_equad}, dt)
    else:
        psr.update_added_signals("{}_measurement_noise".format(psr.name), {}, dt)
        for i, flag in enumerate(flags):
            psr.update_added_signals("{}_{}_measurement_noise".format(psr.name, flag),
                                     {"efac": efac[i], "log10_" + equad_str: log10_equad[i]})
    psr.toas.adjust_TOAs(TimeDelta(dt.to("day")))
    psr.update_residuals()


def add_jitter(psr, log10_ecorr, flagid="f", flags=None, coarsegrain=0.1, seed=None):
    """Add correlated (ECORR) noise of rms ecorr [s] per epoch, epochs = greedy buckets of width ``coarsegrain``
    days (white_noise.py:128-198)."""
    ecorr = 10 ** log10_ecorr
    if seed is not None:
        np.random.seed(seed)
    times = np.asarray(psr.toas.get_mjds().value, dtype=np.float64)
    epoch_of, first = epoch_map(times, coarsegrain)
    ne = len(first)
    ecorrvec = np.zeros(ne)
    if flags is None:
        if not np.isscalar(ecorr):
            raise ValueError("ERROR: If flags is None, jitter must be a scalar")
        ecorrvec = np.ones(ne) * ecorr
    if flags is not None and not np.isscalar(ecorr):
        if len(ecorr) == len(flags):
            aveflags = np.array([f[flagid] for f in psr.toas.table["flags"].data])[first]  # first TOA labels the epoch (:35)
            for ct, flag in enumerate(flags):
                ecorrvec[flag == aveflags] = ecorr[ct]
        else:
            raise ValueError("ERROR: flags must be same length as jitter")

    z = np.random.randn(ne)
    n = len(times)
    ep_d, ec_d, z_d = dv.i32(epoch_of), dv.f64(ecorrvec), dv.f64(z)
    out = dv.empty((1, n))
    _lib.call("pta_ecorr", dv.ptr(ep_d), dv.ptr(ec_d), n, ne, dv.ptr(z_d), ne, 1, dv.ptr(out), n, 0, dv.stream_ptr())
    dt = u.s * out[0].cpu().numpy()

    if flags is None:
        psr.update_added_signals("{}_jitter".format(psr.name), {"log10_ecorr": log10_ecorr}, dt)
    else:
        psr.update_added_signals("{}_jitter".format(psr.name), {}, dt)
        for i, flag in enumerate(flags):
            psr.update_added_signals("{}_{}_jitter".format(psr.name, flag), {"log10_ecorr": log10_ecorr[i]})
    psr.toas.adjust_TOAs(TimeDelta(dt.to("day")))
    psr.update_residuals()


def add_efac(psr, efac=1.0, flagid="f", flags=None, seed=None):
    """libstempo-style alias: EFAC only (BASELINE.json names add_efac/add_ecorr)."""
    return add_measurement_noise(psr, efac=efac, log10_equad=None if flags is None else np.full(len(flags), -np.inf),
                                 flagid=flagid, flags=flags, seed=seed)


def add_equad(psr, log10_equad, flagid="f", flags=None, seed=None):
    """libstempo-style alias: EQUAD only (tnequad convention, EFAC = 0 on the sigma term)."""
    efac = 0.0 if flags is None else np.zeros(len(flags))
    return add_measurement_noise(psr, efac=efac, log10_equad=log10_equad, flagid=flagid, flags=flags, seed=seed, tnequad=True)


add_ecorr = add_jitter
