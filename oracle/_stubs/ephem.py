"""Stand-in for pyephem — TEST INFRASTRUCTURE ONLY.

The reference only touches ephem in its ELONG/ELAT branch (red_noise.py:210-221,
deterministic.py:79-88).  pyephem (libastro) is not installed, so that branch
cannot be run here; the goldens use RAJ/DECJ pulsars (the same as the
reference's own test) and the ecliptic branch is "parity unpinned".
"""


class Ecliptic:
    def __init__(self, *a, **k):
        raise RuntimeError("pyephem is not available: ecliptic branch is unpinned")


class Equatorial:
    def __init__(self, *a, **k):
        raise RuntimeError("pyephem is not available: ecliptic branch is unpinned")
