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
