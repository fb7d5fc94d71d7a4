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


def pair_zeta_cos_loop(psr_locs):
    """[P, P, 2]: (zeta_ab, cos zeta_ab) for a <= b, pair by pair on NumPy float64 scalars exactly like the reference's loop
    (spharmORFbasis.py:400-408,166).  The definition pair_zeta_cos is checked against - and its fallback; 61 ms at 200 pulsars."""
    P = len(psr_locs)
    zc = np.zeros((P, P, 2))
    for a in range(P):
        for b in range(a, P):
            z = calczeta(psr_locs[a][0], psr_locs[b][0], psr_locs[a][1], psr_locs[b][1])
            zc[a, b] = zc[b, a] = (z, np.cos(z))
    return zc


_PAIR_CHECK_SAMPLES = 48
_native_pairs_ok = True     # cleared (once, with a warning) on a host where the native evaluation is not the scalar loop's bit for bit


def pair_zeta_cos(psr_locs):
    """[P, P, 2]: (zeta_ab, cos zeta_ab), the numbers the reference's per-pair loop produces (spharmORFbasis.py:14-35,400-408,166) -
    the ill-conditioned l >= 3 sums on the device start from the very doubles the reference uses - without its 20 100 Python-level
    iterations at 200 pulsars (61 ms of loop in front of 0.38 ms of kernel):

    * the arccos argument of every pair comes from native host code (pta_orf_pair_arguments: libm sin / cos, the reference's
      left-to-right association, no contraction) - NumPy routes float64 scalar sin / cos to the same libm;
    * zeta = arccos(argument) and cos(zeta) are ONE NumPy call each over the pair array: NumPy's float64 arccos is NOT libm's on AVX512
      hosts (SVML: 9 % of arguments differ by an ulp here), but its array loop is the inner loop its scalars go through;
    * the reference's branches (identical positions -> 0, argument < -1 -> pi, > 1 -> 0) are applied on the arrays;
    * every call re-derives a deterministic sample of pairs (all of them up to 48, the antipodal-most and closest included) with the
      scalar loop and compares bit for bit; a host whose NumPy scalars do not route that way gets the loop for everything, with one warning.
    """
    global _native_pairs_ok
    psr_locs = np.ascontiguousarray(psr_locs, dtype=np.float64)
    P = len(psr_locs)
    if not _native_pairs_ok or P == 0:
        return pair_zeta_cos_loop(psr_locs)
    arg = np.empty((P, P))
    same = np.empty((P, P), dtype=np.uint8)
    _lib.call("pta_orf_pair_arguments", dv.hptr(psr_locs), P, dv.hptr(arg), dv.hptr(same))
    zeta = np.arccos(np.clip(arg, -1.0, 1.0))            # clip only guards the call: the branches below overwrite what it touched
    zeta[arg < -1] = np.pi
    zeta[arg > 1] = 0.0
    zeta[same != 0] = 0.0
    zc = np.stack([zeta, np.cos(zeta)], axis=2)
    iu = np.triu_indices(P)
    if len(iu[0]) > _PAIR_CHECK_SAMPLES:
        au = arg[iu]
        ends = np.argpartition(au, (3, len(au) - 4))
        pick = np.unique(np.concatenate([ends[:4], ends[-4:], np.random.default_rng(P).choice(len(au), _PAIR_CHECK_SAMPLES - 8, replace=False)]))
    else:
        pick = np.arange(len(iu[0]))
    for a, b in zip(iu[0][pick], iu[1][pick]):
        z = calczeta(psr_locs[a][0], psr_locs[b][0], psr_locs[a][1], psr_locs[b][1])
        if not (z == zc[a, b, 0] and np.cos(z) == zc[a, b, 1]):
            import warnings
            warnings.warn("pta_orf_pair_arguments does not reproduce NumPy's scalar sin / cos bit for bit on this host: the pair separations "
                          "fall back to the per-pair Python loop", RuntimeWarning)
            _native_pairs_ok = False
            return pair_zeta_cos_loop(psr_locs)
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
