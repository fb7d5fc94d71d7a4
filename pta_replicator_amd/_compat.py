"""astropy units / TimeDelta when astropy is installed, otherwise a minimal stand-in.

The reference hands ``astropy`` quantities to PINT (``dt * u.s``, ``TimeDelta(dt.to('day'))``:
red_noise.py:128-134, white_noise.py:105-124).  With PINT-backed pulsars astropy is necessarily
present and is used as is.  On machines without astropy (this project's build and GPU boxes) the
array-backed pulsars of ``pta_replicator_amd.simulate`` accept the small Quantity/TimeDelta below,
which implements the same conversions (``to('day')`` multiplies by 1/86400, as astropy does).
"""
import numpy as np

try:  # pragma: no cover - depends on the environment
    import astropy.units as u
    from astropy.time import TimeDelta
    HAVE_ASTROPY = True
except ImportError:
    HAVE_ASTROPY = False

    _TO_S = {"s": 1.0, "day": 86400.0, "us": 1e-6, "d": 86400.0}

    class _Unit:
        __array_ufunc__ = None

        def __init__(self, name):
            self.name = name

        def __rmul__(self, other):
            return Quantity(np.asarray(other, dtype=np.float64), self.name)

        __mul__ = __rmul__

        def __repr__(self):
            return self.name

    class Quantity:
        __array_ufunc__ = None

        def __init__(self, value, unit):
            self.value = value
            self.unit = unit if isinstance(unit, str) else unit.name

        def to(self, unit):
            name = unit if isinstance(unit, str) else unit.name
            if name == self.unit:
                return Quantity(self.value, name)
            return Quantity(self.value * (_TO_S[self.unit] / _TO_S[name]), name)

        def to_value(self, unit):
            return self.to(unit).value

        def __add__(self, other):
            return Quantity(self.value + other.to(self.unit).value, self.unit)

        def __mul__(self, other):
            return Quantity(self.value * np.asarray(other), self.unit)

        __rmul__ = __mul__

        def __neg__(self):
            return Quantity(-self.value, self.unit)

        def __len__(self):
            return len(self.value)

        def __getitem__(self, idx):
            return Quantity(self.value[idx], self.unit)

        def __repr__(self):
            return f"<Quantity {self.value!r} {self.unit}>"

    class _Units:
        s = _Unit("s")
        day = _Unit("day")
        us = _Unit("us")
        Quantity = Quantity

    u = _Units()

    class TimeDelta:
        def __init__(self, quantity):
            self.quantity = quantity

        def to_value(self, unit):
            return self.quantity.to_value(unit)


def delta_days(td):
    """float64 day values of a TimeDelta (astropy's or the stand-in's)."""
    if hasattr(td, "quantity"):
        return np.asarray(td.quantity.to_value("day"), dtype=np.float64)
    return np.asarray(td.to_value("day"), dtype=np.float64)
