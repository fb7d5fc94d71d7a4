"""Run the UNMODIFIED reference (``/root/reference/pta_replicator``) under stubs.

TEST / MEASUREMENT INFRASTRUCTURE ONLY — this file is imported by ``oracle/gen_golden.py`` (fixtures) and by
``oracle/cpu_baseline.py`` (the ``cpu_baseline`` leg of bench.py, kind "reference"), never by the product.  It
only works where ``/root/reference`` is mounted (the build container); on the GPU box the baseline falls back
to the NumPy port.

What it provides
----------------
* ``load_reference()``  – puts ``oracle/_stubs`` (astropy/pint/enterprise/ephem/numba/
  holodeck stand-ins) and ``/root/reference`` on ``sys.path`` and imports the reference's
  own ``red_noise``, ``white_noise``, ``deterministic``, ``spharmORFbasis`` and
  ``simulate`` modules, byte-for-byte as shipped.
* ``MockTOAs``          – the duck-typed TOA container the reference functions touch
  (SURVEY.md §8b): ``table['tdbld']``, ``table['flags'].data``, ``get_mjds()``,
  ``first_MJD``/``last_MJD``, ``ntoas``, ``get_errors()``, ``adjust_TOAs()``.
  Times are held in x87 ``longdouble`` (as PINT's ``tdbld`` column); every
  ``adjust_TOAs`` shift is recorded so that the residual oracle is
  ``sum(shifts) - mean`` (PINT's default mean-subtracted residual).
* ``make_pulsar()``     – builds a reference ``SimulatedPulsar`` around a ``MockTOAs``
  with ``make_ideal`` already applied (``added_signals = {}``).
"""
import os
import sys
import types

import numpy as np

REFERENCE_ROOT = "/root/reference"
_HERE = os.path.dirname(os.path.abspath(__file__))
_STUBS = os.path.join(_HERE, "_stubs")


def load_reference():
    """Import the reference modules under the stubs; returns a namespace."""
    if not os.path.isdir(REFERENCE_ROOT):
        raise RuntimeError(f"{REFERENCE_ROOT} is not mounted: the reference can only be run in the build container")
    for p in (_STUBS, REFERENCE_ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    import pta_replicator.simulate as simulate
    import pta_replicator.red_noise as red_noise
    import pta_replicator.white_noise as white_noise
    import pta_replicator.deterministic as deterministic
    import pta_replicator.spharmORFbasis as spharm
    import pta_replicator.constants as constants

    # PINT is absent: residual rebuild is a no-op, the oracle residual is sum(shifts)-mean
    simulate.SimulatedPulsar.update_residuals = lambda self: None
    return types.SimpleNamespace(simulate=simulate, red_noise=red_noise, white_noise=white_noise,
                                 deterministic=deterministic, spharm=spharm, constants=constants)


class _Column:
    def __init__(self, data):
        self.data = data


class _Scalar:
    def __init__(self, value):
        self.value = value


class MockTOAs:
    """float64/longdouble TOA container with the surface listed in SURVEY.md §8b."""

    def __init__(self, mjd, errors_us, flags=None):
        self.mjd_ld = np.array(mjd, dtype=np.longdouble)
        self.errors_us = np.array(errors_us, dtype=np.float64) * np.ones(len(self.mjd_ld))
        self.flags = flags if flags is not None else [dict() for _ in range(len(self.mjd_ld))]
        self.shifts_day = []  # every adjust_TOAs() argument, in call order

    # -- columns ------------------------------------------------------------------
    @property
    def table(self):
        return {"tdbld": self.mjd_ld, "flags": _Column(self.flags)}

    @property
    def ntoas(self):
        return len(self.mjd_ld)

    def get_mjds(self):
        from astropy import units as u  # the stub
        return u.Quantity(self.mjd_ld.astype(np.float64), "day")

    @property
    def first_MJD(self):
        return _Scalar(float(self.mjd_ld.min()))

    @property
    def last_MJD(self):
        return _Scalar(float(self.mjd_ld.max()))

    def get_errors(self):
        from astropy import units as u
        return u.Quantity(self.errors_us.copy(), "us")

    # -- the sink -----------------------------------------------------------------
    def adjust_TOAs(self, delta):
        d = np.asarray(delta.quantity.to("day").value, dtype=np.float64)
        self.shifts_day.append(d.copy())
        self.mjd_ld = self.mjd_ld + d.astype(np.longdouble)

    def residuals_s(self):
        """sum of all shifts in seconds, weighted-mean removed (equal weights when errors are equal)."""
        tot = np.zeros(self.ntoas)
        for d in self.shifts_day:
            tot = tot + d * 86400.0
        w = 1.0 / self.errors_us ** 2
        return tot - np.sum(tot * w) / np.sum(w)


def make_pulsar(ref, name, mjd, errors_us, loc, flags=None):
    psr = ref.simulate.SimulatedPulsar(model=None, toas=MockTOAs(mjd, errors_us, flags), name=name, loc=dict(loc))
    psr.added_signals = {}          # what make_ideal() does (simulate.py:200-201)
    psr.added_signals_time = {}
    return psr


# ---------------------------------------------------------------------------------
# par / tim readers (just enough for the reference's own fixtures)
# ---------------------------------------------------------------------------------
def _sexagesimal(s):
    sign = -1.0 if s.strip().startswith("-") else 1.0
    parts = [np.longdouble(p) for p in s.strip().lstrip("+-").split(":")]
    while len(parts) < 3:
        parts.append(np.longdouble(0))
    return float(sign * (parts[0] + parts[1] / 60 + parts[2] / 3600))


def read_par_loc(parfile):
    """Returns (name, loc) with loc as simulate.py:127-132 builds it (RAJ hours / DECJ deg, or ELONG/ELAT deg)."""
    vals = {}
    with open(parfile) as fh:
        for line in fh:
            tok = line.split()
            if len(tok) >= 2 and tok[0] in ("PSR", "PSRJ", "RAJ", "DECJ", "ELONG", "ELAT", "LAMBDA", "BETA"):
                vals[tok[0]] = tok[1]
    name = vals.get("PSR", vals.get("PSRJ"))
    if "RAJ" in vals and "DECJ" in vals:
        loc = {"RAJ": _sexagesimal(vals["RAJ"]), "DECJ": _sexagesimal(vals["DECJ"])}
    else:
        loc = {"ELONG": float(vals.get("ELONG", vals.get("LAMBDA"))), "ELAT": float(vals.get("ELAT", vals.get("BETA")))}
    return name, loc


def read_tim(timfile):
    """Returns (mjd longdouble[N], err_us float64[N], flags list[dict])."""
    mjd, err, flags = [], [], []
    with open(timfile) as fh:
        for line in fh:
            tok = line.split()
            if len(tok) < 5 or tok[0] in ("FORMAT", "MODE", "C", "#", "JUMP", "SKIP", "NOSKIP", "TIME", "INCLUDE"):
                continue
            try:
                m = np.longdouble(tok[2])
                e = float(tok[3])
            except ValueError:
                continue
            fl = {}
            rest = tok[5:]
            i = 0
            while i + 1 < len(rest):
                if rest[i].startswith("-") and not _is_number(rest[i]):
                    fl[rest[i][1:]] = rest[i + 1]
                    i += 2
                else:
                    i += 1
            mjd.append(m)
            err.append(e)
            flags.append(fl)
    return np.array(mjd, dtype=np.longdouble), np.array(err), flags


def _is_number(s):
    try:
        float(s)
        return True
    except ValueError:
        return False
