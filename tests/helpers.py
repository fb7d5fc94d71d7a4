"""Shared helpers for the tests (not product code)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
S_TO_DAY = 1.0 / 86400.0   # astropy's s->day factor: Quantity.to('day') multiplies by this


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def mjd_ld(z, prefix, i):
    return z[f"{prefix}mjd_hi_{i}"].astype(np.longdouble) + z[f"{prefix}mjd_lo_{i}"].astype(np.longdouble)


def shift_s(mjd, dt_s):
    """what adjust_TOAs(TimeDelta(dt.to('day'))) does to a longdouble MJD column."""
    return mjd + (np.asarray(dt_s, dtype=np.float64) * S_TO_DAY).astype(np.longdouble)


def shift_day(mjd, dt_day):
    return mjd + np.asarray(dt_day, dtype=np.float64).astype(np.longdouble)


def relrms(a, b):
    """max |a-b| / rms(b): the two-sided per-pulsar form of the reference's own pass criterion."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    rms = np.sqrt(np.mean(b ** 2))
    if rms == 0:
        return float(np.max(np.abs(a - b)))
    return float(np.max(np.abs(a - b)) / rms)


def oracle_realisation(po, psrs, noise, draws, components=30, coarsegrain=0.1, gw_gamma=13. / 3.):
    """One realisation of bench.headline_array()-style inputs (per-backend EFAC / t2EQUAD / ECORR, per-pulsar RN, HD GWB)
    through the CPU oracle, on the deviates `draws` (shaped like ReplicaEngine.dump_draws()).  Returns one array per pulsar."""
    P = len(psrs)
    mjd = [np.asarray(p.toas.get_mjds().value, dtype=np.float64) for p in psrs]
    grid = po.gwb_grid([float(m.min()) for m in mjd], [float(m.max()) for m in mjd])
    Mchol = np.linalg.cholesky(po.hd_orf_closed_form(po.psr_locs_equatorial([p.loc for p in psrs])))
    C = po.gwb_spectrum(grid["f"], grid["dur"], grid["howml"], noise["gw_log10_A"], gw_gamma)
    res_gw, _ = po.gwb_dt(grid, Mchol, draws["gwb"], C, [m * 86400 for m in mjd])
    out = []
    for a, psr in enumerate(psrs):
        n = len(mjd[a])
        tf = np.array([f["f"] for f in psr.toas.table["flags"].data])
        sig = np.asarray(psr.toas.get_errors().to("s").value, dtype=np.float64)
        efv = po.flag_vector(tf, noise["flags"][a], noise["efac"][a], n)
        eqv = po.flag_vector(tf, noise["flags"][a], 10 ** np.asarray(noise["log10_equad"][a]), n)
        ref = po.measurement_noise_dt(sig, efv, eqv, *draws["wn"][a])
        epoch_of, ne, first, _ = po.quantize(mjd[a], dt=coarsegrain)
        ecv = po.jitter_ecorr_vector(ne, first, noise["log10_ecorr"][a], toa_flags=tf, flags=noise["flags"][a])
        ref = ref + po.jitter_dt(epoch_of, ecv, draws["ecorr"][a])
        if noise["rn_log10_A"][a] is not None:
            ref = ref + po.red_noise_dt(psr.toas.table["tdbld"], noise["rn_log10_A"][a], noise["rn_gamma"][a], draws["rn"][a],
                                        components=components)
        out.append(ref + res_gw[a])
    return out
