"""Sky position of a pulsar from its ``loc`` dict, as the reference derives it.

red_noise.py:204-221 and deterministic.py:76-88: RAJ [hourangle] / DECJ [deg] are used directly; ELONG / ELAT [deg] go through
pyephem - ``ephem.Equatorial(ephem.Ecliptic(str(elong), str(elat)), epoch='1950' if 'B' in name else '2000')`` - i.e. through
libastro.  pyephem is importable neither here nor in the reference tree, so when it is absent the ecliptic branch runs a
RESTATEMENT of the libastro path pyephem 4.1 takes for that expression (restated from the published XEphem / libastro 3.7 sources,
which are not in this image - file and function names are given so that a maintainer with the sources can check them line by line):

  * ``Ecliptic(lon, lat)`` carries epoch J2000; ``Equatorial(other, epoch=e)`` = ``other.to_radec()`` then ``precess(other.epoch, e)``
    (ephem/__init__.py: ``Coordinate.__init__``);
  * ``to_radec`` = ``ecl_eq(mjd, lat, lng)`` (libastro eq_ecl.c: ``ecleq_aux(sw = -1)``) with the MEAN OBLIQUITY OF THE COORDINATE'S EPOCH
    (libastro obliq.c: ``23.4392911 deg + t (-46.8150 + t (-0.00059 + t 0.001813)) / 3600``, t in centuries from J2000 - note the
    constant is 84381.44796", not the IAU's 84381.448");
  * ``precess`` (libastro precess.c: ``precess_hiprec``) goes from_equinox -> 2000.0 -> to_equinox in decimal YEARS (mjd.c:
    ``mjd_year``), skipping a leg that is within 0.02 yr of 2000.0, with the IAU 1976 angles in DEGREES rounded to seven decimals
    (0.6406161, 0.0000839, 0.0000050 / 0.6406161, 0.0003041, 0.0000051 / 0.5567530, -0.0001185, -0.0000116).

PARITY UNPINNED: no pyephem-generated fixture can exist in this image, so ELONG / ELAT pulsars stay OUTSIDE the 1e-10 residual
claim (1 arcsecond = 5e-6 rad of sky position), and add_gwb / add_cgw say so once per process (``warn_ecliptic_unpinned``).  The
restatement is checked against an independent rotation-matrix evaluation of IAU 1976 precession and the textbook obliquity at the
sub-milliarcsecond level (tests/test_host_logic.py), which bounds formula errors, not libastro's last digits.
"""
import math
import warnings

import numpy as np

try:  # pragma: no cover
    import ephem as _ephem
except ImportError:
    _ephem = None

_J2000_MJD = 36525.0          # libastro's MJD epoch is 1899 Dec 31.5: J2000.0 = 36525.0
_warned = False


def warn_ecliptic_unpinned(name):
    """one warning per process: this pulsar's position came from the unpinned ELONG / ELAT restatement."""
    global _warned
    if _ephem is None and not _warned:
        _warned = True
        warnings.warn(f"pulsar {name}: ELONG/ELAT converted by a restatement of pyephem/libastro (pyephem is not installed); positions - and "
                      "therefore the GWB / CGW residuals of ecliptic-coordinate pulsars - are outside the 1e-10 parity claim "
                      "(pta_replicator_amd/_position.py)", RuntimeWarning, stacklevel=3)


def _libastro_obliquity(mjd):
    """libastro obliq.c: mean obliquity [rad] at libastro MJD `mjd`."""
    t = (mjd - _J2000_MJD) / 36525.0
    return math.radians(23.4392911 + t * (-46.8150 + t * (-0.00059 + t * 0.001813)) / 3600.0)


def _libastro_ecl_eq(mjd, lat, lng):
    """libastro eq_ecl.c: ecl_eq() = ecleq_aux(sw = -1, x = lng, y = lat): (ra, dec) [rad] referred to the equinox of `mjd`."""
    eps = _libastro_obliquity(mjd)
    seps, ceps = math.sin(eps), math.cos(eps)
    sy, cy = math.sin(lat), math.cos(lat)
    if abs(cy) < 1e-20:
        cy = 1e-20
    ty = sy / cy
    cx, sx = math.cos(lng), math.sin(lng)
    sq = sy * ceps + cy * seps * sx            # (sy*ceps) - (cy*seps*sx*sw), sw = -1
    sq = max(-1.0, min(1.0, sq))
    q = math.asin(sq)
    p = math.atan((sx * ceps - ty * seps) / cx)  # ((sx*ceps) + (ty*seps*sw)) / cx
    if cx < 0:
        p += math.pi
    p -= 2 * math.pi * math.floor(p / (2 * math.pi))   # range(&p, 2 PI)
    return p, q


def _libastro_precess_from_2000(to_year, ra, dec):
    """libastro precess.c: the second leg of precess_hiprec (2000.0 -> to_equinox), angles in degrees as there."""
    if abs(to_year - 2000.0) <= 0.02:
        return ra, dec
    T = (to_year - 2000.0) / 100.0
    zeta_A = 0.6406161 * T + 0.0000839 * T * T + 0.0000050 * T * T * T
    z_A = 0.6406161 * T + 0.0003041 * T * T + 0.0000051 * T * T * T
    theta_A = 0.5567530 * T - 0.0001185 * T * T - 0.0000116 * T * T * T
    a2000, d2000 = math.degrees(ra), math.degrees(dec)
    ds, dc = (lambda x: math.sin(math.radians(x))), (lambda x: math.cos(math.radians(x)))
    A = ds(a2000 + zeta_A) * dc(d2000)
    B = dc(a2000 + zeta_A) * dc(theta_A) * dc(d2000) - ds(theta_A) * ds(d2000)
    C = dc(a2000 + zeta_A) * ds(theta_A) * dc(d2000) + dc(theta_A) * ds(d2000)
    alpha = math.degrees(math.atan2(A, B)) + z_A
    alpha -= 360.0 * math.floor(alpha / 360.0)
    return math.radians(alpha), math.radians(math.degrees(math.asin(C)))


def ecliptic_to_equatorial(elong_deg, elat_deg, name):
    """(ra, dec) in radians; epoch 1950 when "B" is in the pulsar name (red_noise.py:214-219)."""
    if _ephem is not None:  # pragma: no cover - exact reference behaviour
        epoch = "1950" if "B" in name else "2000"
        coords = _ephem.Equatorial(_ephem.Ecliptic(str(elong_deg), str(elat_deg)), epoch=epoch)
        return float(repr(coords.ra)), float(repr(coords.dec))
    # Ecliptic(str, str): degrees parsed from the decimal strings (exact round trip of the floats), epoch J2000
    lng, lat = math.radians(float(str(elong_deg))), math.radians(float(str(elat_deg)))
    ra, dec = _libastro_ecl_eq(_J2000_MJD, lat, lng)
    if "B" in name:
        # epoch '1950' = 1950/1/1 00:00 = year 1950.0 exactly (mjd_year); the J2000 -> 2000.0 leg is skipped (|2000.00137 - 2000| < 0.02)
        ra, dec = _libastro_precess_from_2000(1950.0, ra, dec)
    return float(ra), float(dec)


def ra_dec(psr, default=None):
    """(ra, dec) [rad] with the reference's branch order (its `"RAJ" and "DECJ" in loc` tests DECJ only).  A loc with
    neither DECJ nor ELAT: add_gwb silently leaves the pulsar at (0, 0) (red_noise.py:203-221: no else branch), which callers
    on that path request with default=(0.0, 0.0); add_cgw fails on it (deterministic.py:76-91), as this does without a default."""
    loc = psr.loc
    if "DEC_RAD" in loc and "RA_RAD" in loc:   # simulate.from_enterprise: the radians an enterprise-style pulsar carried, unrounded
        # ... as long as they still describe the position RAJ / DECJ give: a caller who edits RAJ / DECJ afterwards means those
        if "DECJ" not in loc or (abs(float(loc["RAJ"]) * np.pi / 12.0 - float(loc["RA_RAD"])) < 1e-12
                                 and abs(float(loc["DECJ"]) * np.pi / 180.0 - float(loc["DEC_RAD"])) < 1e-12):
            return float(loc["RA_RAD"]), float(loc["DEC_RAD"])
    if "DECJ" in loc:
        return float(loc["RAJ"] * np.pi / 12.0), float(loc["DECJ"] * np.pi / 180.0)
    if "ELAT" in loc:
        warn_ecliptic_unpinned(psr.name)
        return ecliptic_to_equatorial(loc["ELONG"], loc["ELAT"], psr.name)
    if default is not None:
        return default
    raise AttributeError("No pulsar location information (RAJ/DECJ or ELONG/ELAT) in psr.loc.")
