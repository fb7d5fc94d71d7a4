"""Minimal stand-in for ``astropy.units`` — TEST INFRASTRUCTURE ONLY.

Exists so that the *unmodified* reference modules under /root/reference can be
imported and executed in this container (astropy is not installed and there is
no network).  It implements exactly the surface those modules touch:
``ndarray * u.s``, ``u.s * ndarray``, ``Quantity.to()``, ``.value``, ``+``/``+=``
between quantities and scalar/array scaling.  Nothing in the product package
imports this file.
"""
import numpy as np

_TO_SECONDS = {"s": 1.0, "day": 86400.0, "us": 1e-6, "MHz": 1.0}


class Unit:
    # make ``ndarray * Unit`` dispatch to Unit.__rmul__ instead of broadcasting
    __array_ufunc__ = None

    def __init__(self, name):
        self.name = name

    def __rmul__(self, other):
        return Quantity(np.asarray(other, dtype=np.float64), self.name)

    def __mul__(self, other):
        return Quantity(np.asarray(other, dtype=np.float64), self.name)

    def __repr__(self):
        return f"Unit({self.name})"


class Quantity:
    __array_ufunc__ = None

    def __init__(self, value, unit):
        self.value = value
        self.unit = unit

    def to(self, unit):
        name = unit.name if isinstance(unit, Unit) else str(unit)
        if name == self.unit:
            return Quantity(self.value, name)
        # same association as astropy: value * (from/to) computed as one factor
        return Quantity(self.value * (_TO_SECONDS[self.unit] / _TO_SECONDS[name]), name)

    def to_value(self, unit):
        return self.to(unit).value

    def _other(self, other):
        if isinstance(other, Quantity):
            return other.to(self.unit).value
        raise TypeError("can only add Quantity to Quantity")

    def __add__(self, other):
        return Quantity(self.value + self._other(other), self.unit)

    def __iadd__(self, other):
        self.value = self.value + self._other(other)
        return self

    def __mul__(self, other):
        if isinstance(other, Unit):
            raise TypeError("compound units not needed")
        return Quantity(self.value * np.asarray(other), self.unit)

    __rmul__ = __mul__

    def __neg__(self):
        return Quantity(-self.value, self.unit)

    def __truediv__(self, other):
        return Quantity(self.value / np.asarray(other), self.unit)

    def __len__(self):
        return len(self.value)

    def __repr__(self):
        return f"Quantity({self.value!r}, {self.unit})"


s = Unit("s")
day = Unit("day")
us = Unit("us")
MHz = Unit("MHz")
