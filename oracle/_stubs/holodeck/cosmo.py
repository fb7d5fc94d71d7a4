"""Stand-in for ``holodeck.cosmo`` — TEST INFRASTRUCTURE ONLY (see utils.py: parity unpinned).

holodeck's default cosmology is astropy's FlatLambdaCDM with the WMAP9 parameters h = 0.6932, Om0 = 0.2865 and no
radiation term (Tcmb0 = 0); ``z_to_dcom`` returns the line-of-sight comoving distance in cm,
    d_c(z) = (c / H0) int_0^z dz' / sqrt(Om0 (1+z')^3 + 1 - Om0).
Evaluated here with adaptive quadrature, element by element (the product uses a fixed Gauss-Legendre rule)."""
import numpy as np
from scipy import integrate

H0_KM_S_MPC = 69.32
OM0 = 0.2865
_MPC_CM = 3.0856775814913674e24
_C_KM_S = 299792.458


def z_to_dcom(z):
    z = np.atleast_1d(np.asarray(z, dtype=np.float64))
    out = np.empty_like(z)
    for i, zi in enumerate(z):
        out[i] = integrate.quad(lambda x: 1.0 / np.sqrt(OM0 * (1.0 + x) ** 3 + 1.0 - OM0), 0.0, zi, epsabs=0, epsrel=1e-13)[0]
    return out * (_C_KM_S / H0_KM_S_MPC) * _MPC_CM
