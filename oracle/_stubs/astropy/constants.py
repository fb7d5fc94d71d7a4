"""Stand-in for ``astropy.constants`` — TEST INFRASTRUCTURE ONLY.  deterministic.py:617-618 reads
``ap.constants.pc.cgs.value`` and ``ap.constants.M_sun.cgs.value``; the values are astropy's (CODATA 2018 / IAU 2015)."""


class _Value:
    def __init__(self, v):
        self.value = v


class _Const:
    def __init__(self, cgs):
        self.cgs = _Value(cgs)


pc = _Const(3.0856775814913674e18)       # cm
M_sun = _Const(1.988409870698051e33)     # g
