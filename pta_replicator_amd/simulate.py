"""Pulsar container and loaders: the drop-in boundary of the injection functions.

Mirrors ``pta_replicator/simulate.py`` (SimulatedPulsar :23-95, simulate_pulsar :98-135, load_pulsar
:138-167, load_from_directories :170-190, make_ideal :193-202).  This is host-side bookkeeping and is
deliberately NOT accelerated (BASELINE.json: "Python host code owns the Pulsar/TOA bookkeeping").

Two kinds of pulsar work with every ``add_*`` function of this package:

* PINT-backed: any object with the reference's duck-type surface - e.g. the reference's
  own ``SimulatedPulsar`` holding ``pint.toa.TOAs`` + ``TimingModel``.  Used as is when PINT is installed.
* array-backed: ``ArrayTOAs`` below keeps the TOAs as a longdouble MJD column (the precision of PINT's
  ``tdbld``), implements the same surface, and defines the residual as PINT does for an idealised
  pulsar: accumulated TOA shift minus the weighted mean.  This is what runs on machines without
  PINT/astropy (the GPU box) and what the batched engine is built from.
"""
import glob
import os
from dataclasses import dataclass

import numpy as np

from ._compat import HAVE_ASTROPY, TimeDelta, delta_days, u  # noqa: F401

try:  # pragma: no cover - depends on the environment
    import pint.toa as _pint_toa
    from pint import models as _pint_models
    from pint.residuals import Residuals as _PintResiduals
    from pint.simulation import make_fake_toas_fromMJDs as _make_fake_toas_fromMJDs
    import pint.fitter as _pint_fitter
    HAVE_PINT = True
except ImportError:
    HAVE_PINT = False


class _Column:
    def __init__(self, data):
        self.data = data


class _Scalar:
    def __init__(self, value):
        self.value = value


class ArrayResiduals:
    """time residuals of an idealised pulsar: total injected delay, weighted-mean subtracted (PINT's default).

    A snapshot, like ``pint.residuals.Residuals``: it keeps the accumulated-shift column as it was when the object was built
    (ArrayTOAs.adjust_TOAs REPLACES that array, never mutates it) and evaluates ``resids_value`` on first access - an injection
    loop that rebuilds the residuals after every add_* (simulate.py:40-42) pays for the arithmetic only when somebody looks."""

    def __init__(self, toas):
        # the accumulated shift is tracked on its own: differencing two longdouble MJDs (resolution 2.5e-10 s at
        # MJD 53000) would bury the 1e-10-relative parity this package is tested to
        self._shift_day = toas.shift_day
        self._err_us = toas.errors_us
        self._resids = None

    @property
    def resids_value(self):
        if self._resids is None:
            shift = self._shift_day * 86400.0
            w = 1.0 / self._err_us ** 2
            self._resids = shift - np.sum(shift * w) / np.sum(w)
        return self._resids

    @property
    def time_resids(self):
        return self.resids_value * u.s

    def get_data_error(self):
        return self._err_us * u.us


class ArrayTOAs:
    """TOA table with the surface the injection functions touch, in longdouble MJD."""

    def __init__(self, mjd, errors_us, flags=None, freqs_mhz=1440.0):
        self.mjd0_ld = np.array(mjd, dtype=np.longdouble)  # "ideal" TOAs: zero residual by construction
        self.mjd_ld = self.mjd0_ld.copy()
        # sum of all adjust_TOAs() deltas [day], float64: the delays are ~1e-11 day, their sum is exact to 1e-16 RELATIVE - the
        # longdouble is needed for the MJDs (5e4 day + 1e-11 day), not here, and x87 adds are 7x slower than these
        self.shift_day = np.zeros(len(self.mjd_ld), dtype=np.float64)
        n = len(self.mjd_ld)
        self.errors_us = np.asarray(errors_us, dtype=np.float64) * np.ones(n)
        self.freqs_mhz = np.asarray(freqs_mhz, dtype=np.float64) * np.ones(n)
        self.flags = list(flags) if flags is not None else [dict() for _ in range(n)]
        if len(self.flags) != n or len(self.errors_us) != n:
            raise ValueError("mjd, errors and flags must have the same length")

    @property
    def shift_day_ld(self):
        return self.shift_day.astype(np.longdouble)

    @property
    def table(self):
        return {"tdbld": self.mjd_ld, "flags": _Column(self.flags)}

    @property
    def ntoas(self):
        return len(self.mjd_ld)

    def _mjd_f64(self):
        """the MJDs rounded to float64, converted once per TOA state (adjust_TOAs REPLACES mjd_ld, so the array's identity is the
        state): get_mjds / first_MJD / last_MJD of 68 pulsars cost 2.6 ms per add_gwb as longdouble reductions and conversions."""
        c = getattr(self, "_f64", None)
        if c is None or c[0] is not self.mjd_ld:
            c = self._f64 = (self.mjd_ld, self.mjd_ld.astype(np.float64))
        return c[1]

    def get_mjds(self):
        return self._mjd_f64().copy() * u.day     # a copy: the caller owns what it gets

    @property
    def first_MJD(self):
        return _Scalar(float(self._mjd_f64().min()))   # rounding is monotone: the rounded minimum IS the minimum of the rounded values

    @property
    def last_MJD(self):
        return _Scalar(float(self._mjd_f64().max()))

    def get_errors(self):
        return self.errors_us.copy() * u.us

    def adjust_TOAs(self, delta):
        d = delta_days(delta)
        self.mjd_ld = self.mjd_ld + d.astype(np.longdouble)   # a NEW array each time: residual snapshots keep the one they saw
        self.shift_day = self.shift_day + d

    def reset_ideal(self):
        self.mjd_ld = self.mjd0_ld.copy()
        self.shift_day = np.zeros(len(self.mjd_ld), dtype=np.float64)


class EnterpriseTOAs(ArrayTOAs):
    """ArrayTOAs over the arrays of an enterprise-style pulsar (``toas`` [s], ``toaerrs`` [s]): the MJD column is toas / 86400 in
    longdouble and the uncertainties stay in SECONDS, so ``get_errors().to('s')`` - what the injection functions and the engine read
    (white_noise.py:105) - hands back the caller's numbers bit for bit instead of a us round trip."""

    def __init__(self, toas_s, toaerrs_s, flags=None, freqs_mhz=1440.0):
        toas_s = np.asarray(toas_s, dtype=np.float64)
        super().__init__(toas_s.astype(np.longdouble) / np.longdouble(86400.0), np.asarray(toaerrs_s, dtype=np.float64) * 1e6, flags, freqs_mhz)
        self.errors_s = np.array(toaerrs_s, dtype=np.float64) * np.ones(len(toas_s))   # the caller's numbers, bit for bit

    # ONE error column: the seconds the caller handed over.  ``errors_us`` - what ArrayTOAs' own code (the errors_seconds()
    # cache check, write_tim, to_enterprise) reads and what a caller may rescale - is a VIEW of it: assigning errors_us updates the seconds.
    @property
    def errors_us(self):
        return self.errors_s * 1e6

    @errors_us.setter
    def errors_us(self, value):
        self.errors_s = np.asarray(value, dtype=np.float64) * 1e-6

    def get_errors(self):
        return self.errors_s.copy() * u.s


def is_enterprise_like(psr):
    """an object with enterprise.pulsar.BasePulsar's array surface (``toas`` [s] and ``toaerrs`` [s] as plain arrays) rather than the
    reference's SimulatedPulsar surface (``toas.get_mjds()`` ...)"""
    toas = getattr(psr, "toas", None)
    return toas is not None and not hasattr(toas, "get_mjds") and hasattr(psr, "toaerrs") and np.ndim(toas) == 1


def from_enterprise(epsr):
    """Array-backed ``SimulatedPulsar`` from an ENTERPRISE-style pulsar - the objects BASELINE.json's north_star names and the
    reference's hand-off produces (simulate.py:91-95).  Read, in enterprise's units (enterprise/pulsar.py): ``name``; ``toas`` [s] =
    MJD * 86400; ``toaerrs`` [s]; ``freqs`` [MHz] (optional); ``flags`` (dict flag -> per-TOA array; '' = the TOA lacks the flag) and
    ``backend_flags`` (becomes flag ``f`` when ``flags`` has none); position from ``_raj`` / ``_decj`` [rad] when present (enterprise keeps
    them), else ``phi`` / ``theta``, else the unit vector ``pos``.  The result is ``make_ideal``-ed: every ``add_*`` function and
    ``ReplicaEngine`` take it as is (the engine also accepts the enterprise-style objects directly and wraps them through this).
    Nothing is written back into ``epsr``: hand realisations back with ``SimulatedPulsar.to_enterprise()`` /
    ``ReplicaEngine.to_enterprise()``."""
    toas_s = np.asarray(epsr.toas, dtype=np.float64)
    n = len(toas_s)
    toaerrs = np.asarray(epsr.toaerrs, dtype=np.float64) * np.ones(n)
    freqs = np.asarray(getattr(epsr, "freqs", 1440.0), dtype=np.float64) * np.ones(n)
    fl = getattr(epsr, "flags", None) or {}
    cols = {str(k): np.asarray(v) for k, v in dict(fl).items() if np.ndim(v) == 1 and len(v) == n}
    if "f" not in cols and getattr(epsr, "backend_flags", None) is not None and len(epsr.backend_flags) == n:
        cols["f"] = np.asarray(epsr.backend_flags)
    flags = [{k: str(v[i]) for k, v in cols.items() if str(v[i]) != ""} for i in range(n)]
    if getattr(epsr, "_raj", None) is not None and getattr(epsr, "_decj", None) is not None:
        ra, dec = float(epsr._raj), float(epsr._decj)
    elif getattr(epsr, "phi", None) is not None and getattr(epsr, "theta", None) is not None:
        ra, dec = float(epsr.phi), float(np.pi / 2 - epsr.theta)
    elif getattr(epsr, "pos", None) is not None:
        x, y, z = (float(c) for c in np.asarray(epsr.pos, dtype=np.float64)[:3])
        ra, dec = float(np.arctan2(y, x) % (2 * np.pi)), float(np.arcsin(max(-1.0, min(1.0, z / np.sqrt(x * x + y * y + z * z)))))
    else:
        raise AttributeError(f"enterprise-style pulsar {getattr(epsr, 'name', '?')}: no position (_raj/_decj, phi/theta or pos)")
    # RAJ [hourangle] / DECJ [deg] as the reference's loc dict holds them; RA_RAD / DEC_RAD carry the radians unrounded (_position.ra_dec)
    loc = {"RAJ": ra * 12.0 / np.pi, "DECJ": np.degrees(dec), "RA_RAD": ra, "DEC_RAD": dec}
    psr = SimulatedPulsar(toas=EnterpriseTOAs(toas_s, toaerrs, flags, freqs), name=str(epsr.name), loc=loc)
    make_ideal(psr)
    return psr


def as_simulated(psr):
    """``psr`` itself when it has the reference's SimulatedPulsar surface, else (enterprise-style arrays) its from_enterprise() wrapper"""
    return from_enterprise(psr) if is_enterprise_like(psr) else psr


@dataclass
class SimulatedPulsar:
    """Same fields and methods as the reference's dataclass (simulate.py:23-95)."""
    ephem: str = "DE440"
    model: object = None
    toas: object = None
    residuals: object = None
    name: str = None
    loc: dict = None
    added_signals: dict = None
    added_signals_time: dict = None
    par_text: str = None     # array-backed pulsars: the par file the pulsar was loaded from, verbatim (injection never changes the model)

    def __repr__(self):
        return f"SimulatedPulsar({self.name})"

    def update_residuals(self):
        """Rebuild the residuals from the current TOAs (simulate.py:40-42)."""
        if isinstance(self.toas, ArrayTOAs):
            self.residuals = ArrayResiduals(self.toas)
        else:
            self.residuals = _PintResiduals(self.toas, self.model)

    def fit(self, fitter="auto", **fitter_kwargs):
        """Refit the timing model (simulate.py:44-69); needs PINT."""
        if isinstance(self.toas, ArrayTOAs) or not HAVE_PINT:
            raise NotImplementedError("fit() needs a PINT-backed pulsar (pint-pulsar is not installed here)")
        if fitter == "wls":
            self.f = _pint_fitter.WLSFitter(self.toas, self.model)
        elif fitter == "gls":
            self.f = _pint_fitter.GLSFitter(self.toas, self.model)
        elif fitter == "downhill":
            self.f = _pint_fitter.DownhillGLSFitter(self.toas, self.model)
        elif fitter == "auto":
            self.f = _pint_fitter.Fitter.auto(self.toas, self.model)
        else:
            raise ValueError(f"{fitter=} must be one of 'wls', 'gls', 'downhill' or 'auto'")
        self.f.fit_toas(**fitter_kwargs)
        self.model = self.f.model
        self.update_residuals()

    def write_partim(self, outpar, outtim, tempo2=False):
        """Write par/tim (simulate.py:71-77).  Array-backed pulsars: a Tempo2 tim file of the shifted TOAs, and the par file the pulsar
        was loaded from, verbatim - injection shifts TOAs, it never touches the timing model (the reference writes its unchanged
        ``model`` back out unless fit() was called).  A pulsar built in code has no par source: a minimal par with its name and
        position is written, marked as such."""
        if isinstance(self.toas, ArrayTOAs):
            write_tim(outtim, self.name, self.toas.mjd_ld, self.toas.errors_us, self.toas.freqs_mhz, self.toas.flags)
            with open(outpar, "w") as fh:
                fh.write(self.par_text if self.par_text else minimal_par(self.name, self.loc))
            return
        self.model.write_parfile(outpar)
        self.toas.write_TOA_file(outtim, format="Tempo2") if tempo2 else self.toas.write_TOA_file(outtim)

    def update_added_signals(self, signal_name, param_dict, dt=None):
        """Record an injected signal; same checks and messages as simulate.py:79-89."""
        if self.added_signals is None:
            raise ValueError("make_ideal() must be called on SimulatedPulsar before adding new signals.")
        if signal_name in self.added_signals:
            raise ValueError(f"{signal_name} already exists in the model.")
        self.added_signals[signal_name] = param_dict
        if dt is not None:
            self.added_signals_time[signal_name] = dt

    def to_enterprise(self, ephem="DE440"):
        """Hand-off to enterprise (simulate.py:91-95).  PINT-backed pulsars: the reference's own call,
        ``enterprise.pulsar.Pulsar(toas, model, ephem=ephem, timing_package="pint")``.  Array-backed pulsars: an
        ``ArrayEnterprisePulsar`` carrying the attribute surface enterprise's signal models read (no par/tim round trip)."""
        if isinstance(self.toas, ArrayTOAs):
            if self.residuals is None:
                self.update_residuals()
            return ArrayEnterprisePulsar.from_simulated(self)
        from enterprise.pulsar import Pulsar
        return Pulsar(self.toas, self.model, ephem=ephem, timing_package="pint")


def timing_design_matrix(toas_s, ra=0.0, dec=0.0, model="spin"):
    """Design matrix of an idealised isolated pulsar's timing model on the TOAs ``toas_s`` [s]: the columns a
    timing fit would project out of the injected delays.  ``model``:
      "spin"         offset, F0 (t), F1 (t^2)                                                                    3 columns
      "astrometric"  + position (annual cos / sin), proper motion (t x annual cos / sin), parallax (semi-annual)  9 columns
    Columns are scaled to unit maximum (what enterprise's TimingModel does with ``normed=True``); the projection
    r - M (M^T W M)^-1 M^T W r is invariant under column scaling.  This is NOT PINT's design matrix of the pulsar's par file (binary,
    DM and JUMP columns are absent - PINT is not installable here, f4 is unpinned); it spans the same subspace for an isolated pulsar
    with a quadratic spin-down and linearised astrometry."""
    t = np.asarray(toas_s, dtype=np.float64)
    t = t - t.mean()
    scale = max(float(np.max(np.abs(t))), 1.0)
    x = t / scale
    cols, names = [np.ones_like(x), x, x ** 2], ["Offset", "F0", "F1"]
    if model == "astrometric":
        w = 2.0 * np.pi / (365.25 * 86400.0)
        c, s = np.cos(w * t), np.sin(w * t)
        cols += [c, s, x * c, x * s, np.cos(2 * w * t), np.sin(2 * w * t)]
        names += ["RAJ", "DECJ", "PMRA", "PMDEC", "PX", "PX_quad"]
    elif model != "spin":
        raise ValueError(f"{model=} must be 'spin' or 'astrometric'")
    return np.stack(cols, axis=1), names


class ArrayEnterprisePulsar:
    """The part of ``enterprise.pulsar.BasePulsar`` that enterprise's signal classes read, filled straight from arrays
    (lets an analysis consume GPU-generated realisations without writing and re-reading par/tim files).

    Attributes, in enterprise's units: ``name``; ``toas`` [s, MJD * 86400]; ``residuals`` [s]; ``toaerrs`` [s]; ``freqs`` [MHz];
    ``flags`` (dict flag -> array of str, one entry per TOA, '' where a TOA lacks the flag); ``backend_flags`` (the 'f' flag, else
    'group'/'be' as enterprise falls back); ``Mmat`` the timing-model design matrix of an idealised pulsar (timing_design_matrix:
    "spin" = offset, t, t^2 - the columns that absorb the mean and the spin-down a fit would remove -, or "astrometric" = 9
    columns); ``pos`` unit vector, ``theta`` / ``phi`` [rad]; ``pdist`` (1.0, 0.2) kpc as enterprise defaults it.  UNPINNED:
    enterprise is not installed here and the reference has no test of its hand-off; the attribute list follows
    enterprise/pulsar.py."""

    def __init__(self, name, mjd, residuals_s, toaerrs_us, freqs_mhz, flags, ra, dec, timing_model="spin"):
        order = np.argsort(np.asarray(mjd, dtype=np.float64), kind="mergesort")   # enterprise sorts TOAs by default (sort=True)
        self.name = name
        self._isort = order
        self.toas = (np.asarray(mjd, dtype=np.float64) * 86400.0)[order]
        self.stoas = self.toas.copy()
        self.residuals = np.asarray(residuals_s, dtype=np.float64)[order]
        self.toaerrs = (np.asarray(toaerrs_us, dtype=np.float64) * 1e-6)[order]
        self.freqs = np.asarray(freqs_mhz, dtype=np.float64)[order]
        keys = sorted({k for f in flags for k in f})
        self.flags = {k: np.array([str(flags[i].get(k, "")) for i in order]) for k in keys}
        for k in ("f", "group", "be"):
            if k in self.flags:
                self.backend_flags = self.flags[k]
                break
        else:
            self.backend_flags = np.array([""] * len(order))
        self.Mmat, self.fitpars = timing_design_matrix(self.toas, ra, dec, timing_model)
        self._raj, self._decj = float(ra), float(dec)   # enterprise keeps the radians it derived pos / theta / phi from
        self.theta, self.phi = float(np.pi / 2 - dec), float(ra)
        self.pos = np.array([np.cos(ra) * np.cos(dec), np.sin(ra) * np.cos(dec), np.sin(dec)])
        self.pdist = (1.0, 0.2)
        self.dm = None

    @classmethod
    def from_simulated(cls, psr, residuals_s=None, toa_shift_s=None, timing_model="spin"):
        """From any pulsar object with the duck-type surface of SURVEY.md §8b (array-backed, PINT-backed or foreign): TOAs, errors and
        flags are read through get_mjds() / get_errors() / table['flags'], the way the injection functions read them.  The TOAs
        handed over are the CURRENT (shifted) ones, like the reference's hand-off (simulate.py:91-95 passes the adjusted TOAs);
        ``toa_shift_s`` adds a further per-TOA shift [s] (ReplicaEngine.to_enterprise: ideal TOAs + the realisation's delay)."""
        from ._position import ra_dec
        ra, dec = ra_dec(psr, default=(0.0, 0.0))
        toas = psr.toas
        if isinstance(toas, ArrayTOAs):
            mjd, err, freq, flags = toas.mjd_ld, toas.errors_us, toas.freqs_mhz, toas.flags
        else:
            mjd = np.asarray(toas.get_mjds().value, dtype=np.longdouble)
            err = np.asarray(toas.get_errors().to("us").value, dtype=np.float64)
            flags = list(toas.table["flags"].data)
            try:
                freq = np.asarray(toas.get_freqs().to("MHz").value, dtype=np.float64)
            except AttributeError:
                freq = np.asarray(getattr(toas, "freqs_mhz", 1440.0), dtype=np.float64) * np.ones(len(mjd))
        if toa_shift_s is not None:
            mjd = np.asarray(mjd, dtype=np.longdouble) + (np.asarray(toa_shift_s, dtype=np.float64) / 86400.0).astype(np.longdouble)
        if residuals_s is None:
            if psr.residuals is None:
                psr.update_residuals()
            res = getattr(psr.residuals, "resids_value", None)
            if res is None:
                res = np.asarray(psr.residuals.time_resids.to("s").value, dtype=np.float64)
        else:
            res = residuals_s
        return cls(psr.name, np.asarray(mjd, dtype=np.float64), res, err, freq, flags, ra, dec, timing_model=timing_model)

    def __repr__(self):
        return f"ArrayEnterprisePulsar({self.name}, {len(self.toas)} TOAs)"


# ----------------------------------------------------------------------------------------------
# par / tim readers for the array-backed path
# ----------------------------------------------------------------------------------------------
def _sexagesimal(text):
    text = text.strip()
    sign = -1.0 if text.startswith("-") else 1.0
    parts = [np.longdouble(p) for p in text.lstrip("+-").split(":")]
    parts += [np.longdouble(0)] * (3 - len(parts))
    return float(sign * (parts[0] + parts[1] / 60 + parts[2] / 3600))


def read_par_location(parfile):
    """(name, loc): loc = {'RAJ' [hourangle], 'DECJ' [deg]} or {'ELONG','ELAT'} [deg], as simulate.py:127-132."""
    if not os.path.isfile(parfile):
        raise FileNotFoundError("par file does not exist.")
    vals = {}
    with open(parfile) as fh:
        for line in fh:
            tok = line.split()
            if len(tok) >= 2 and tok[0] in ("PSR", "PSRJ", "PSRB", "RAJ", "DECJ", "ELONG", "ELAT", "LAMBDA", "BETA"):
                vals.setdefault(tok[0], tok[1])
    name = vals.get("PSR", vals.get("PSRJ", vals.get("PSRB")))
    if "RAJ" in vals and "DECJ" in vals:
        loc = {"RAJ": _sexagesimal(vals["RAJ"]), "DECJ": _sexagesimal(vals["DECJ"])}
    elif ("ELONG" in vals or "LAMBDA" in vals) and ("ELAT" in vals or "BETA" in vals):
        loc = {"ELONG": float(vals.get("ELONG", vals.get("LAMBDA"))), "ELAT": float(vals.get("ELAT", vals.get("BETA")))}
    else:
        raise AttributeError("No pulsar location information (RAJ/DECJ or ELONG/ELAT) in parfile.")
    return name, loc


def write_tim(path, name, mjd_ld, errors_us, freqs_mhz, flags):
    """Tempo2 "FORMAT 1" tim file from arrays (19 decimals of MJD: sub-ns at longdouble precision)."""
    with open(path, "w") as fh:
        fh.write("FORMAT 1\n")
        for i in range(len(mjd_ld)):
            fl = " ".join(f"-{k} {v}" for k, v in flags[i].items())
            fh.write(f" {name} {freqs_mhz[i]:.8f} {np.format_float_positional(mjd_ld[i], precision=19)} {errors_us[i]:.5f} AXIS {fl}\n")


def _sexagesimal_str(value, sign=False):
    s = "-" if value < 0 else ("+" if sign else "")
    v = abs(float(value))
    d = int(v)
    m = int((v - d) * 60)
    sec = (v - d - m / 60.0) * 3600.0
    return f"{s}{d:02d}:{m:02d}:{sec:011.8f}"


def minimal_par(name, loc):
    """par text for a pulsar that was built in code (no par file to copy): name and position only, flagged as a stub - enough for
    tools that read the sky position; not a timing solution."""
    lines = ["# written by pta_replicator_amd for an array-backed pulsar without a par source: name and position only", f"PSR {name}"]
    if loc and "RAJ" in loc and "DECJ" in loc:
        lines += [f"RAJ {_sexagesimal_str(loc['RAJ'])}", f"DECJ {_sexagesimal_str(loc['DECJ'], sign=True)}"]
    elif loc and "ELONG" in loc and "ELAT" in loc:
        lines += [f"ELONG {float(loc['ELONG']):.12f}", f"ELAT {float(loc['ELAT']):.12f}"]
    return "\n".join(lines) + "\n"


def read_tim(timfile):
    """Tempo2-format tim file -> (mjd longdouble[N], error_us[N], freq_mhz[N], flags list[dict])."""
    if not os.path.isfile(timfile):
        raise FileNotFoundError("tim file does not exist.")
    mjd, err, freq, flags = [], [], [], []
    with open(timfile) as fh:
        for line in fh:
            tok = line.split()
            if len(tok) < 5 or tok[0] in ("FORMAT", "MODE", "C", "#", "JUMP", "SKIP", "NOSKIP", "TIME", "INCLUDE", "EFAC", "EQUAD"):
                continue
            try:
                f_, m_, e_ = float(tok[1]), np.longdouble(tok[2]), float(tok[3])
            except ValueError:
                continue
            fl, rest, i = {}, tok[5:], 0
            while i + 1 < len(rest):
                key = rest[i]
                is_num = key.lstrip("-").replace(".", "", 1).replace("e", "", 1).isdigit()
                if key.startswith("-") and not is_num:
                    fl[key[1:]] = rest[i + 1]
                    i += 2
                else:
                    i += 1
            mjd.append(m_); err.append(e_); freq.append(f_); flags.append(fl)
    return np.array(mjd, dtype=np.longdouble), np.array(err), np.array(freq), flags


def simulate_pulsar(parfile, obstimes, toaerr, freq=1440.0, observatory="AXIS", flags=None, ephem="DE440"):
    """SimulatedPulsar from a par file and observation times [MJD] (simulate.py:98-135)."""
    if not os.path.isfile(parfile):
        raise FileNotFoundError("par file does not exist.")
    if HAVE_PINT:  # pragma: no cover
        model = _pint_models.get_model(parfile)
        toas = _make_fake_toas_fromMJDs(obstimes, model, freq=freq * u.MHz, obs=observatory, flags=flags, error=toaerr * u.us)
        name, loc = read_par_location(parfile)
        return SimulatedPulsar(ephem=ephem, model=model, toas=toas, residuals=_PintResiduals(toas, model), name=model.PSR.value, loc=loc)
    name, loc = read_par_location(parfile)
    n = len(obstimes)
    fl = [dict(flags) for _ in range(n)] if isinstance(flags, dict) else flags
    toas = ArrayTOAs(obstimes, toaerr, fl, freq)
    with open(parfile) as fh:
        par_text = fh.read()
    psr = SimulatedPulsar(ephem=ephem, model=None, toas=toas, name=name, loc=loc, par_text=par_text)
    psr.update_residuals()
    return psr


def load_pulsar(parfile, timfile, ephem="DE440"):
    """SimulatedPulsar from par + tim files (simulate.py:138-167)."""
    if not os.path.isfile(parfile):
        raise FileNotFoundError("par file does not exist.")
    if not os.path.isfile(timfile):
        raise FileNotFoundError("tim file does not exist.")
    name, loc = read_par_location(parfile)
    if HAVE_PINT:  # pragma: no cover
        model = _pint_models.get_model(parfile)
        toas = _pint_toa.get_TOAs(timfile, ephem=ephem, planets=True)
        return SimulatedPulsar(ephem=ephem, model=model, toas=toas, residuals=_PintResiduals(toas, model), name=model.PSR.value, loc=loc)
    mjd, err, freq, flags = read_tim(timfile)
    with open(parfile) as fh:
        par_text = fh.read()
    psr = SimulatedPulsar(ephem=ephem, model=None, toas=ArrayTOAs(mjd, err, flags, freq), name=name, loc=loc, par_text=par_text)
    psr.update_residuals()
    return psr


def load_from_directories(pardir, timdir, ephem="DE440", num_psrs=None, debug=False):
    """Pair sorted par and tim files of two directories (simulate.py:170-190)."""
    if not os.path.isdir(pardir):
        raise FileNotFoundError("par directory does not exist.")
    if not os.path.isdir(timdir):
        raise FileNotFoundError("tim directory does not exist.")
    pars = [p for p in sorted(glob.glob(pardir + "/*.par")) if ".t2" not in p]
    tims = sorted(glob.glob(timdir + "/*.tim"))
    psrs = []
    for par, tim in zip(pars, tims):
        if num_psrs and len(psrs) >= num_psrs:
            break
        if debug:
            print(f"loading {par=}, {tim=}")
        psrs.append(load_pulsar(par, tim, ephem=ephem))
    return psrs


def make_ideal(psr, iterations=2):
    """Zero the residuals and open the added-signal registry (simulate.py:193-202)."""
    if isinstance(psr.toas, ArrayTOAs):
        psr.toas.reset_ideal()  # array-backed TOAs carry their ideal column: exact in one step
    else:  # pragma: no cover - PINT path, same iteration as the reference
        for _ in range(iterations):
            residuals = _PintResiduals(psr.toas, psr.model)
            psr.toas.adjust_TOAs(TimeDelta(-1.0 * residuals.time_resids))
    psr.added_signals = {}
    psr.added_signals_time = {}
    psr.update_residuals()
