"""Stand-in for ``holodeck.utils`` — TEST INFRASTRUCTURE ONLY.

holodeck is a third-party dependency that is neither under /root/reference nor installed here, and the reference does
not pin a version of it (it is not even listed in pyproject.toml).  The three functions deterministic.py:623-631 calls
are restated from their published definitions (holodeck/utils.py, Kelley et al.; cgs units):
    m1m2_from_mtmr : m1 = mt / (1 + q), m2 = mt - m1
    chirp_mass     : (m1 m2)^(3/5) / (m1 + m2)^(1/5)
    gw_strain_source : h_s = 8/sqrt(10) (G Mc)^(5/3) (2 pi f_orb)^(2/3) / (c^4 d_c)   (the reference's own comment at
                       deterministic.py:633-638 spells the same formula)
PARITY UNPINNED for these three (no holodeck to compare with); everything downstream of them is pinned by running the
unmodified reference function on top of this stub.
"""
import numpy as np

NWTG = 6.6743e-8          # cm^3 g^-1 s^-2
SPLC = 2.99792458e10      # cm s^-1


def m1m2_from_mtmr(mt, mr):
    mt = np.asarray(mt)
    mr = np.asarray(mr)
    m1 = mt / (1.0 + mr)
    m2 = mt - m1
    return np.array([m1, m2])


def chirp_mass(m1, m2):
    return np.power(m1 * m2, 3.0 / 5.0) / np.power(m1 + m2, 1.0 / 5.0)


def gw_strain_source(mchirp, dcom, freq_rest_orb):
    const = 8.0 * np.power(NWTG, 5.0 / 3.0) * np.power(np.pi, 2.0 / 3.0) / np.sqrt(10.0) / np.power(SPLC, 4.0)
    return const * mchirp * np.power(2.0 * mchirp * freq_rest_orb, 2.0 / 3.0) / dcom
