"""Sky position of a pulsar from its ``loc`` dict, as the reference derives it.

red_noise.py:204-221 and deterministic.py:76-88: RAJ [hourangle] / DECJ [deg] are used directly; ELONG /
ELAT [deg] go through pyephem, with names containing "B" precessed to epoch 1950.  When pyephem is not
installed the ecliptic branch falls back to a restatement of the standard IAU formulas below - PARITY
UNPINNED (pyephem/libastro is absent from the reference tree and from this image, and the reference's
only test uses RAJ/DECJ pulsars); differences are expected at the arcsecond level at most.
"""
import numpy as np

try:  # pragma: no cover
    import ephem as _ephem
except ImportError:
    _ephem = None

_ARCSEC = np.pi / 180.0 / 3600.0


def _ecliptic_to_equatorial_j2000(lon_deg, lat_deg):
    eps = 84381.448 * _ARCSEC  # IAU 1976/1980 mean obliquity at J2000 (what libastro uses at epoch 2000)
    lam, bet = np.radians(lon_deg), np.radians(lat_deg)
    ra = np.arctan2(np.sin(lam) * np.cos(eps) - np.tan(bet) * np.sin(eps), np.cos(lam))
    dec = np.arcsin(np.sin(bet) * np.cos(eps) + np.cos(bet) * np.sin(eps) * np.sin(lam))
    return ra % (2 * np.pi), dec


def _precess_from_j2000(ra, dec, jd):
    """IAU 1976 (Lieske) precession of equatorial coordinates from J2000.0 to the equinox of `jd`."""
    T = (jd - 2451545.0) / 36525.0
    zeta = (2306.2181 * T + 0.30188 * T ** 2 + 0.017998 * T ** 3) * _ARCSEC
    z = (2306.2181 * T + 1.09468 * T ** 2 + 0.018203 * T ** 3) * _ARCSEC
    theta = (2004.3109 * T - 0.42665 * T ** 2 - 0.041833 * T ** 3) * _ARCSEC
    A = np.cos(dec) * np.sin(ra + zeta)
    B = np.cos(theta) * np.cos(dec) * np.cos(ra + zeta) - np.sin(theta) * np.sin(dec)
    C = np.sin(theta) * np.cos(dec) * np.cos(ra + zeta) + np.cos(theta) * np.sin(dec)
    return (np.arctan2(A, B) + z) % (2 * np.pi), np.arcsin(C)


def ecliptic_to_equatorial(elong_deg, elat_deg, name):
    """(ra, dec) in radians; epoch 1950 when "B" is in the pulsar name (red_noise.py:214-219)."""
    if _ephem is not None:  # pragma: no cover - exact reference behaviour
        epoch = "1950" if "B" in name else "2000"
        coords = _ephem.Equatorial(_ephem.Ecliptic(str(elong_deg), str(elat_deg)), epoch=epoch)
        return float(repr(coords.ra)), float(repr(coords.dec))
    ra, dec = _ecliptic_to_equatorial_j2000(elong_deg, elat_deg)
    if "B" in name:
        ra, dec = _precess_from_j2000(ra, dec, 2433282.5)  # ephem.Date('1950') = 1950-01-01 00:00
    return float(ra), float(dec)


def ra_dec(psr, default=None):
    """(ra, dec) [rad] with the reference's branch order (its `"RAJ" and "DECJ" in loc` tests DECJ only).  A loc with
    neither DECJ nor ELAT: add_gwb silently leaves the pulsar at (0, 0) (red_noise.py:203-221: no else branch), which callers
    on that path request with default=(0.0, 0.0); add_cgw fails on it (deterministic.py:76-91), as this does without a default."""
    loc = psr.loc
    if "DECJ" in loc:
        return float(loc["RAJ"] * np.pi / 12.0), float(loc["DECJ"] * np.pi / 180.0)
    if "ELAT" in loc:
        return ecliptic_to_equatorial(loc["ELONG"], loc["ELAT"], psr.name)
    if default is not None:
        return default
    raise AttributeError("No pulsar location information (RAJ/DECJ or ELONG/ELAT) in psr.loc.")
