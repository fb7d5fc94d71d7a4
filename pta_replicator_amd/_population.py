"""Source-population helpers of add_gwb_plus_outlier_cws (deterministic.py:565-715).

The reference takes these from holodeck (``utils.chirp_mass``, ``utils.m1m2_from_mtmr``, ``utils.gw_strain_source``,
``cosmo.z_to_dcom``; deterministic.py:623-631) - a dependency it does not pin and that is not installed here - so they are
written from their published definitions in cgs units.  Parity with holodeck itself is UNPINNED; a caller
who has holodeck can pass its functions through the ``population`` argument of add_gwb_plus_outlier_cws.
"""
import numpy as np

G_CGS = 6.6743e-8                  # cm^3 g^-1 s^-2 (CODATA 2018)
C_CGS = 2.99792458e10              # cm / s
PC_CGS = 3.0856775814913674e18     # astropy.constants.pc.cgs (deterministic.py:617)
MSOL_CGS = 1.988409870698051e33    # astropy.constants.M_sun.cgs (deterministic.py:618)

# holodeck's default cosmology: flat LCDM, WMAP9, no radiation
H0_KM_S_MPC = 69.32
OMEGA_M = 0.2865

_GL_X, _GL_W = np.polynomial.legendre.leggauss(48)


def component_masses(mtot, mrat):
    """m1 (primary), m2 from total mass and mass ratio q = m2/m1 <= 1."""
    mtot = np.asarray(mtot, dtype=np.float64)
    m1 = mtot / (1.0 + np.asarray(mrat, dtype=np.float64))
    return m1, mtot - m1


def chirp_mass(m1, m2):
    return np.power(m1 * m2, 3.0 / 5.0) / np.power(m1 + m2, 1.0 / 5.0)


def comoving_distance_cm(z):
    """d_c(z) = c/H0 int_0^z dz'/E(z'), E^2 = Om (1+z)^3 + 1 - Om; 48-point Gauss-Legendre per element (relative error
    below 1e-13 for z < 20)."""
    z = np.atleast_1d(np.asarray(z, dtype=np.float64))
    x = 0.5 * z[:, None] * (_GL_X[None, :] + 1.0)
    integrand = 1.0 / np.sqrt(OMEGA_M * (1.0 + x) ** 3 + 1.0 - OMEGA_M)
    integral = 0.5 * z * (integrand @ _GL_W)
    return integral * (C_CGS * 1e-5 / H0_KM_S_MPC) * (PC_CGS * 1e6)


def gw_strain_source(mchirp, dcom, freq_rest_orb):
    """sky- and polarisation-averaged strain of a circular binary: 8/sqrt(10) (G Mc)^(5/3) (2 pi f_orb)^(2/3) / (c^4 d_c)."""
    const = 8.0 / np.sqrt(10.0) * G_CGS ** (5.0 / 3.0) * np.pi ** (2.0 / 3.0) / C_CGS ** 4
    return const * mchirp * np.power(2.0 * mchirp * freq_rest_orb, 2.0 / 3.0) / dcom


class Population:
    """bundle handed to add_gwb_plus_outlier_cws; replace any member with holodeck's function to use holodeck's own numbers"""
    component_masses = staticmethod(component_masses)
    chirp_mass = staticmethod(chirp_mass)
    comoving_distance_cm = staticmethod(comoving_distance_cm)
    gw_strain_source = staticmethod(gw_strain_source)
