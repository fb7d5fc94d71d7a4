"""Overlap-reduction-function basis on the GPU.

Mirrors the public entry point of ``pta_replicator/spharmORFbasis.py``: ``correlated_basis(psr_locs, lmax)``
(:385-434).  The pair x (l, m) loops - pure Python in the reference, 0.27 ms per pair at lmax = 0 and ~40 ms at
lmax = 4 - are one kernel launch here (csrc/pta_orf.h, csrc/pta_orf_kernels.hip): one thread per (pair, l).
"""
import numpy as np

from . import _lib, device as dv

LMAX_SUPPORTED = 8


def correlated_basis_device(psr_locs, lmax):
    """[(lmax+1)^2, P, P] device tensor, l-major, m = -l..l; psr_locs[:,0] = phi, psr_locs[:,1] = colatitude."""
    psr_locs = np.ascontiguousarray(psr_locs, dtype=np.float64)
    P = len(psr_locs)
    if lmax < 0 or lmax > LMAX_SUPPORTED:
        raise ValueError(f"lmax must be in 0..{LMAX_SUPPORTED}")
    locs_d = dv.f64(psr_locs)
    zc_d = dv.f64(pair_zeta_cos(psr_locs))
    basis = dv.zeros(((lmax + 1) ** 2, P, P))
    _lib.call("pta_orf_basis", dv.ptr(locs_d), dv.ptr(zc_d), P, int(lmax), dv.ptr(basis), dv.stream_ptr())
    return basis


def calczeta(phi1, phi2, theta1, theta2):
    """angular separation of two sky positions with the reference's exact-equality and clamping rules
    (spharmORFbasis.py:14-35): identical positions -> 0; arguments outside [-1, 1] clamp to pi / 0."""
    if phi1 == phi2 and theta1 == theta2:
        return 0.0
    argument = np.sin(theta1) * np.sin(theta2) * np.cos(phi1 - phi2) + np.cos(theta1) * np.cos(theta2)
    if argument < -1:
        return np.pi
    if argument > 1:
        return 0.0
    return float(np.arccos(argument))


def pair_zeta_cos(psr_locs):
    """[P, P, 2]: (zeta_ab, cos zeta_ab) for a <= b, evaluated pair by pair on NumPy float64 scalars like the reference's loop
    (spharmORFbasis.py:400-408,166) - C libm underneath - so the ill-conditioned l >= 3 sums on the device start from the very
    numbers the reference uses.  O(P^2) scalar work on the host (0.1 s at 200 pulsars), realisation independent."""
    P = len(psr_locs)
    zc = np.zeros((P, P, 2))
    for a in range(P):
        for b in range(a, P):
            z = calczeta(psr_locs[a][0], psr_locs[b][0], psr_locs[a][1], psr_locs[b][1])
            zc[a, b] = zc[b, a] = (z, np.cos(z))
    return zc


def correlated_basis(psr_locs, lmax):
    """Same return type as the reference: a list of (lmax+1)^2 NumPy [P, P] matrices."""
    return list(correlated_basis_device(psr_locs, lmax).cpu().numpy())


def orf_from_locations(psr_locs, clm=(np.sqrt(4.0 * np.pi),), lmax=0):
    """ORF = 2 * sum_k clm[k] basis[k] (red_noise.py:224-226) as a [P, P] device tensor.

    The default isotropic case (lmax = 0, clm = [sqrt(4 pi)]) takes the closed-form Hellings-Downs kernel."""
    psr_locs = np.ascontiguousarray(psr_locs, dtype=np.float64)
    P = len(psr_locs)
    clm = np.asarray(clm, dtype=np.float64).ravel()
    orf = dv.empty((P, P))
    if lmax == 0 and len(clm) == 1 and clm[0] == np.sqrt(4.0 * np.pi):
        locs_d = dv.f64(psr_locs)
        _lib.call("pta_orf_hd", dv.ptr(locs_d), P, dv.ptr(orf), dv.stream_ptr())
        return orf
    basis = correlated_basis_device(psr_locs, lmax)
    nb = basis.shape[0]
    if len(clm) < nb:
        raise IndexError(f"clm has {len(clm)} coefficients but lmax={lmax} needs {nb}")
    clm_d = dv.f64(clm[:nb])
    _lib.call("pta_orf_combine", dv.ptr(basis), dv.ptr(clm_d), nb, P, dv.ptr(orf), dv.stream_ptr())
    return orf
