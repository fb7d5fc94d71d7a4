"""CPU oracle: a NumPy restatement of the reference's stochastic-injection algebra.

TEST INFRASTRUCTURE ONLY.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this module, and only as the *checker*:
nothing in ``pta_replicator_amd/`` (the product) imports it, and the product raises if
its HIP library is missing rather than falling back to anything in ``oracle/``.

Pinning: every function below is checked by
``tests/test_oracle_golden.py`` against fixtures in ``tests/golden/`` that were produced
by running the UNMODIFIED reference (``/root/reference/pta_replicator``) under dependency
stubs with ``oracle/gen_golden.py``; that stub run itself reproduces the reference's one
shipped golden vector (``tests/libstempo_test_residuals_efac_ecorr_rn_gwb_cgw.npz``) to
5.6e-5 / 1.9e-4 / 3.8e-4 of RMS, inside the reference's own 1e-3 bar
(``tests/test_against_libstempo.py:66``).  Unpinned: the ELONG/ELAT -> RA/DEC branch
(pyephem absent: ``oracle/ecliptic.py`` says "parity unpinned").

All arithmetic is float64, like the reference (it casts PINT's longdouble columns to
float64 before any arithmetic: ``red_noise.py:123``, ``:287``).  Functions take plain
arrays - pulsar/TOA bookkeeping is the caller's business.

Citations are ``file:line`` under ``/root/reference/pta_replicator/``.
"""
import math

import numpy as np

# constants.py:3-8 (scipy.constants values spelled out: G, c, parsec are CODATA-2018 / IAU exact)
DAY_IN_SEC = 86400
YEAR_IN_SEC = 365.25 * DAY_IN_SEC
_G = 6.6743e-11
_C = 299792458.0
_PARSEC = 3.085677581491367e16
SOLAR2S = _G / _C ** 3 * 1.98855e30
KPC2S = _PARSEC / _C * 1e3
MPC2S = _PARSEC / _C * 1e6


# ----------------------------------------------------------------------------------------
# draws: the reference consumes NumPy's legacy global stream
# ----------------------------------------------------------------------------------------
def legacy_normals(seed, counts):
    """``np.random.seed(seed)`` then one ``randn(n)`` per entry of ``counts`` (in order).

    Mirrors red_noise.py:118-119,127 / :175-176,238-240 and white_noise.py:79-80,105-109 /
    :154-155,182.  ``seed=None`` continues the current global stream, like the reference.
    """
    if seed is not None:
        np.random.seed(seed)
    return [np.random.randn(int(n)) for n in counts]


def gwb_draws(seed, npsr, nf):
    """w[ll,:] = randn(Nf) + 1j*randn(Nf), ll ascending  (red_noise.py:238-240)."""
    if seed is not None:
        np.random.seed(seed)
    w = np.zeros((npsr, nf), complex)
    for a in range(npsr):
        re = np.random.randn(nf)
        im = np.random.randn(nf)
        w[a] = re + 1j * im
    return w


# ----------------------------------------------------------------------------------------
# red noise  (red_noise.py:36-103, 106-135)
# ----------------------------------------------------------------------------------------
def fourier_frequencies(toas_s, nmodes=30, Tspan=None, logf=False, fmin=None, fmax=None, modes=None):
    """Sampling frequencies, red_noise.py:61-80."""
    T = Tspan if Tspan is not None else toas_s.max() - toas_s.min()
    if modes is not None:
        return np.asarray(modes), T
    if fmin is None and fmax is None and not logf:
        return 1.0 * np.arange(1, nmodes + 1) / T, T
    lo = 1 / T if fmin is None else fmin
    hi = nmodes / T if fmax is None else fmax
    if logf:
        return np.logspace(np.log10(lo), np.log10(hi), nmodes), T
    return np.linspace(lo, hi, nmodes), T


def fourier_design_matrix(toas_s, nmodes=30, Tspan=None, libstempo_convention=False, modes=None,
                          logf=False, fmin=None, fmax=None, ranphase=None):
    """F[N, 2*nmodes] and the repeated frequency vector (red_noise.py:86-103).

    default convention: even columns sin(2*pi*t*f), odd columns cos; ``libstempo_convention``:
    even columns cos(2*pi*(t-t[0])*f), odd columns sin (red_noise.py:92-101).  ``ranphase`` is
    the optional per-mode phase (red_noise.py:83-84; zeros when ``pshift`` is off).
    """
    f, _ = fourier_frequencies(toas_s, nmodes, Tspan, logf, fmin, fmax, modes)
    nm = len(f)
    ph = np.zeros(nm) if ranphase is None else np.asarray(ranphase)
    t = toas_s - toas_s[0] if libstempo_convention else toas_s
    # same association as the reference: ((2*pi) * t) * f + phase
    arg = 2 * np.pi * t[:, None] * f[None, :] + ph[None, :]
    F = np.empty((len(toas_s), 2 * nm))
    if libstempo_convention:
        F[:, 0::2] = np.cos(arg)
        F[:, 1::2] = np.sin(arg)
    else:
        F[:, 0::2] = np.sin(arg)
        F[:, 1::2] = np.cos(arg)
    return F, np.repeat(f, 2)


def red_noise_prior(freqs2, log10_amplitude, spectral_index, Tspan):
    """prior = A^2 (f/fyr)^-gamma / (12 pi^2 Tspan) * yr^3   (red_noise.py:112-116,126)."""
    A = 10 ** log10_amplitude
    fyr = 1 / YEAR_IN_SEC
    return A ** 2 * (freqs2 / fyr) ** (-spectral_index) / (12 * np.pi ** 2 * Tspan) * YEAR_IN_SEC ** 3


def red_noise_dt(tdbld_mjd, log10_amplitude, spectral_index, z, components=30,
                 libstempo_convention=False, modes=None):
    """dt[N] seconds for one pulsar given the 2*components normals ``z`` (red_noise.py:123-128).

    The ``Tspan`` argument of ``add_red_noise`` is ignored by the reference (:124 overwrites it).
    """
    toas = np.array(tdbld_mjd, dtype="float64") * DAY_IN_SEC
    Tspan = toas.max() - toas.min()
    F, freqs = fourier_design_matrix(toas, nmodes=components, Tspan=Tspan, modes=modes,
                                     libstempo_convention=libstempo_convention)
    y = np.sqrt(red_noise_prior(freqs, log10_amplitude, spectral_index, Tspan)) * z
    return F @ y


# ----------------------------------------------------------------------------------------
# white noise  (white_noise.py:7-44, 47-125, 128-198)
# ----------------------------------------------------------------------------------------
def flag_vector(toa_flags, flags, values, n):
    """per-TOA vector: values[ct] where the TOA's flag == flags[ct], else 0 (white_noise.py:95-101)."""
    vec = np.zeros(n)
    toa_flags = np.asarray(toa_flags)
    for ct, flag in enumerate(flags):
        vec[flag == toa_flags] = values[ct]
    return vec


def measurement_noise_dt(sigma_s, efacvec, equadvec, z1, z2, tnequad=False):
    """dt = efac*sigma*z1 + (efac*equad | equad)*z2   (white_noise.py:105-109).

    z2 is always drawn, even for zero EQUAD (white_noise.py:75-78,107-109)."""
    dt = efacvec * sigma_s * z1
    if tnequad:
        dt = dt + equadvec * z2
    else:
        dt = dt + efacvec * equadvec * z2
    return dt


def quantize(times, flags=None, dt=1.0):
    """Greedy epoch bucketing of white_noise.py:21-35 without the dense U.

    Returns (epoch_of[N] int, n_epochs, first_index[E], avetoas[E]); epoch e's label flag is
    ``flags[first_index[e]]`` (white_noise.py:35) and U[i, e] == (epoch_of[i] == e) (:37-39).
    A new bucket opens when ``t - bucket_ref >= dt`` with bucket_ref the bucket's first TOA (:26-31).
    """
    times = np.asarray(times)
    order = np.argsort(times)
    epoch_of = np.empty(len(times), dtype=np.int64)
    first_index = []
    ref = None
    e = -1
    for i in order:
        if ref is None or not (times[i] - ref < dt):
            ref = times[i]
            e += 1
            first_index.append(i)
        epoch_of[i] = e
    ne = e + 1
    first_index = np.array(first_index, dtype=np.int64)
    avetoas = np.bincount(epoch_of, weights=times, minlength=ne) / np.bincount(epoch_of, minlength=ne)
    return epoch_of, ne, first_index, avetoas


def jitter_dt(epoch_of, ecorr_epoch, z):
    """dt = (U*ecorrvec) @ z  ==  ecorr[e(i)] * z[e(i)]   (white_noise.py:182)."""
    return ecorr_epoch[epoch_of] * z[epoch_of]


def jitter_ecorr_vector(n_epochs, first_index, log10_ecorr, toa_flags=None, flags=None):
    """per-epoch ecorr (white_noise.py:150,165-180): scalar -> all epochs; per-flag -> by first-TOA flag."""
    if flags is None:
        return np.ones(n_epochs) * 10 ** log10_ecorr
    ecorr = 10 ** np.asarray(log10_ecorr, dtype=float)
    aveflags = np.asarray(toa_flags)[first_index]
    vec = np.zeros(n_epochs)
    for ct, flag in enumerate(flags):
        vec[flag == aveflags] = ecorr[ct]
    return vec


# ----------------------------------------------------------------------------------------
# ORF  (spharmORFbasis.py, red_noise.py:200-226)
# ----------------------------------------------------------------------------------------
NORM = 3.0 / (8 * np.pi)          # spharmORFbasis.py:11
_fact = math.factorial


def calczeta(phi1, phi2, theta1, theta2):
    """angular separation with the reference's exact-equality and clamp rules (spharmORFbasis.py:14-35)."""
    if phi1 == phi2 and theta1 == theta2:
        return 0.0
    arg = np.sin(theta1) * np.sin(theta2) * np.cos(phi1 - phi2) + np.cos(theta1) * np.cos(theta2)
    if arg < -1:
        return np.pi
    if arg > 1:
        return 0.0
    return float(np.arccos(arg))


def _T(ii, jj, qq, mm, ll):
    # common factorial weight of the four finite sums (spharmORFbasis.py:49-64 etc.)
    return (2.0 ** (ii - jj)) * (float(_fact(qq)) * float(_fact(ll + jj))) / (
        float(_fact(ii)) * float(_fact(qq - ii)) * float(_fact(jj)) * float(_fact(ll - jj)) * float(_fact(jj - mm)))


def _Fsum(qq, mm, ll, base, extra_p, sign_l, imax):
    """sum_{i<=imax} sum_{j=m..l} (-1)^s T(i,j) (2^p - base^p)/p with p = q-i+j-m+extra_p.

    s = q-i+j+m for the 'minus' family (sign_l False), s = l+q-i+j for the 'plus' family."""
    tot = 0.0
    for ii in range(0, imax + 1):
        for jj in range(mm, ll + 1):
            p = qq - ii + jj - mm + extra_p
            s = (ll + qq - ii + jj) if sign_l else (qq - ii + jj + mm)
            tot += ((2.0 ** (ii - jj)) * (-1.0) ** s) * (
                float(_fact(qq)) * float(_fact(ll + jj)) * (2.0 ** p - base ** p)) / (
                float(_fact(ii)) * float(_fact(qq - ii)) * float(_fact(jj)) * float(_fact(ll - jj))
                * float(_fact(jj - mm)) * p)
    return tot


def Fminus00(qq, mm, ll, zeta):   # spharmORFbasis.py:43-67
    return _Fsum(qq, mm, ll, 1.0 + np.cos(zeta), 1, False, qq)


def Fminus01(qq, mm, ll, zeta):   # spharmORFbasis.py:70-94
    return _Fsum(qq, mm, ll, 1.0 + np.cos(zeta), 2, False, qq)


def Fplus00(qq, mm, ll, zeta):    # spharmORFbasis.py:137-161
    return _Fsum(qq, mm, ll, 1.0 - np.cos(zeta), 1, True, qq)


def Fplus01(qq, mm, ll, zeta):    # spharmORFbasis.py:97-134
    omc = 1.0 - np.cos(zeta)
    tot = _Fsum(qq, mm, ll, omc, 0, True, qq - 1)
    for jj in range(mm + 1, ll + 1):
        tot += ((2.0 ** (qq - jj)) * (-1.0) ** (ll + jj)) * (
            float(_fact(ll + jj)) * (2.0 ** (jj - mm) - omc ** (jj - mm))) / (
            float(_fact(jj)) * float(_fact(ll - jj)) * float(_fact(jj - mm)) * (jj - mm))
    tot += ((-1.0) ** (ll + mm) * 2.0 ** (qq - mm) * float(_fact(ll + mm)) * np.log(2.0 / omc)) / (
        1.0 * float(_fact(mm)) * float(_fact(ll - mm)))
    return tot


def arbORF(mm, ll, zeta):
    """computational-frame Gamma_lm for 0 < zeta <= pi (spharmORFbasis.py:164-248)."""
    c = np.cos(zeta)
    pre = np.sqrt((2.0 * ll + 1.0) * np.pi)
    if mm == 0:
        body = -(1.0 + c) * Fminus00(0, 0, ll, zeta)
        if 0 <= ll <= 2:
            delta = [1.0 + c / 3.0, -(1.0 + c) / 3.0, 2.0 * c / 15.0]
            body = delta[ll] - (1.0 + c) * Fminus00(0, 0, ll, zeta)
        if zeta != 0.0:
            body = body - (1.0 - c) * Fplus01(1, 0, ll, zeta)
        return NORM * 0.5 * pre * body
    if mm == 1:
        body = (-((1.0 + c) ** 1.5 / (1.0 - c) ** 0.5) * Fminus00(1, 1, ll, zeta)
                - ((1.0 - c) ** 1.5 / (1.0 + c) ** 0.5) * Fplus01(2, 1, ll, zeta))
        if ll in (1, 2):
            delta = [2.0 * np.sin(zeta) / 3.0, -2.0 * np.sin(zeta) / 5.0]
            body = (delta[ll - 1]
                    - ((1.0 + c) ** 1.5 / (1.0 - c) ** 0.5) * Fminus00(1, 1, ll, zeta)
                    - ((1.0 - c) ** 1.5 / (1.0 + c) ** 0.5) * Fplus01(2, 1, ll, zeta))
        return NORM * 0.25 * pre * np.sqrt((1.0 * _fact(ll - 1)) / (1.0 * _fact(ll + 1))) * body
    h = mm / 2.0
    body = (((1.0 + c) ** (h + 1) / (1.0 - c) ** h) * Fminus00(mm, mm, ll, zeta)
            - ((1.0 + c) ** h / (1.0 - c) ** (h - 1.0)) * Fminus01(mm - 1, mm, ll, zeta)
            + ((1.0 - c) ** (h + 1) / (1.0 + c) ** h) * Fplus01(mm + 1, mm, ll, zeta)
            - ((1.0 - c) ** h / (1.0 + c) ** (h - 1.0)) * Fplus00(mm, mm, ll, zeta))
    return -NORM * 0.25 * pre * np.sqrt((1.0 * _fact(ll - mm)) / (1.0 * _fact(ll + mm))) * body


def arbCompFrame_ORF(mm, ll, zeta):
    """zeta==0 closed forms with pulsar-term doubling, zeta==pi special cases (spharmORFbasis.py:309-344)."""
    if zeta == 0.0:
        if ll == 0:
            return 2.0 * NORM * 0.25 * np.sqrt(np.pi * 4) * (1 + (np.cos(zeta) / 3.0))
        if ll == 1 and mm == 0:
            return -2 * 0.5 * NORM * (np.sqrt(np.pi / 3.0)) * (1.0 + np.cos(zeta))
        if ll == 2 and mm == 0:
            return 2 * 0.25 * NORM * (4.0 / 3) * (np.sqrt(np.pi / 5)) * np.cos(zeta)
        return 0.0
    if zeta == np.pi:
        if ll > 2 or (ll in (1, 2) and mm != 0):
            return 0.0
        return arbORF(mm, ll, zeta)
    return arbORF(mm, ll, zeta)


def _hyp2f1_terminating(a, b, c, z):
    """2F1(a,b;c;z) for a a non-positive integer: the finite series (what scipy.special.hyp2f1
    evaluates at spharmORFbasis.py:262, where a = m-l <= 0 and b = -k-l <= 0)."""
    n_terms = min(-a, -b) if b <= 0 and float(b).is_integer() else -a
    term = 1.0
    tot = 1.0
    for n in range(int(n_terms)):
        term *= (a + n) * (b + n) / ((c + n) * (n + 1.0)) * z
        tot += term
    return tot


def dlmk(l, m, k, theta1):
    """Wigner small-d as the reference defines it (spharmORFbasis.py:251-268)."""
    if m >= k:
        factor = np.sqrt(float(_fact(l - k)) * float(_fact(l + m)) / float(_fact(l + k)) / float(_fact(l - m)))
        part2 = (np.cos(theta1 / 2)) ** (2 * l + k - m) * (-np.sin(theta1 / 2)) ** (m - k) / float(_fact(m - k))
        part3 = _hyp2f1_terminating(m - l, -k - l, m - k + 1, -((np.tan(theta1 / 2)) ** 2))
        return factor * part2 * part3
    return (-1) ** (m - k) * dlmk(l, k, m, theta1)


def third_euler_angle(phi1, phi2, theta1, theta2):
    """gamma of spharmORFbasis.py:282-306 (arctan + sign fix)."""
    if phi1 == phi2 and theta1 == theta2:
        g = 0
    else:
        g = np.arctan(np.sin(theta2) * np.sin(phi2 - phi1)
                      / (np.cos(theta1) * np.sin(theta2) * np.cos(phi1 - phi2) - np.sin(theta1) * np.cos(theta2)))
    dummy = (np.cos(g) * np.cos(theta1) * np.sin(theta2) * np.cos(phi1 - phi2)
             + np.sin(g) * np.sin(theta2) * np.sin(phi2 - phi1)
             - np.cos(g) * np.sin(theta1) * np.cos(theta2))
    return g if dummy >= 0 else np.pi + g


def _rotated(m, l, phi1, phi2, theta1, theta2, gamma_ml):
    # spharmORFbasis.py:271-279, 347-359
    g = third_euler_angle(phi1, phi2, theta1, theta2)
    tot = 0
    for ii in range(2 * l + 1):
        k = ii - l
        D = np.exp(complex(0.0, -m * phi1)) * dlmk(l, m, k, theta1) * np.exp(complex(0.0, -k * g))
        tot += D.conjugate() * gamma_ml[ii]
    return tot


def real_rotated_Gammas(m, l, phi1, phi2, theta1, theta2, gamma_ml):
    """real-valued cosmic-frame ORF (spharmORFbasis.py:362-382)."""
    if m > 0:
        ans = (1.0 / np.sqrt(2)) * (_rotated(m, l, phi1, phi2, theta1, theta2, gamma_ml)
                                    + (-1) ** m * _rotated(-m, l, phi1, phi2, theta1, theta2, gamma_ml))
        return ans.real
    if m == 0:
        return _rotated(0, l, phi1, phi2, theta1, theta2, gamma_ml).real
    ans = (1.0 / np.sqrt(2) / complex(0.0, 1)) * (_rotated(-m, l, phi1, phi2, theta1, theta2, gamma_ml)
                                                  - (-1) ** m * _rotated(m, l, phi1, phi2, theta1, theta2, gamma_ml))
    return ans.real


def correlated_basis(psr_locs, lmax):
    """list of (lmax+1)^2 PxP matrices, l-major, m=-l..l (spharmORFbasis.py:385-434).

    ``psr_locs[:,0]`` = phi (RA), ``psr_locs[:,1]`` = theta (colatitude)."""
    P = len(psr_locs)
    out = []
    for ll in range(lmax + 1):
        mats = [np.zeros((P, P)) for _ in range(2 * ll + 1)]
        for aa in range(P):
            for bb in range(aa, P):
                p1, p2, t1, t2 = psr_locs[aa, 0], psr_locs[bb, 0], psr_locs[aa, 1], psr_locs[bb, 1]
                zeta = calczeta(p1, p2, t1, t2)
                plus = [arbCompFrame_ORF(mm, ll, zeta) for mm in range(ll + 1)]
                neg = [(-1) ** mm * plus[mm] for mm in range(1, ll + 1)]
                gamma_ml = neg[::-1] + plus
                for mi in range(2 * ll + 1):
                    v = real_rotated_Gammas(mi - ll, ll, p1, p2, t1, t2, gamma_ml)
                    mats[mi][aa, bb] = v
                    mats[mi][bb, aa] = v
        out.extend(mats)
    return out


def hd_orf_closed_form(psr_locs):
    """lmax=0, clm=[sqrt(4 pi)] fast path: ORF_ab = 2*HD(zeta), diagonal 2 (derivation in DESIGN.md;
    equals 2*sqrt(4pi)*correlated_basis(.,0)[0] to rounding - checked in tests)."""
    phi, th = psr_locs[:, 0], psr_locs[:, 1]
    arg = (np.sin(th)[:, None] * np.sin(th)[None, :] * np.cos(phi[:, None] - phi[None, :])
           + np.cos(th)[:, None] * np.cos(th)[None, :])
    zeta = np.arccos(np.clip(arg, -1.0, 1.0))
    same = (phi[:, None] == phi[None, :]) & (th[:, None] == th[None, :])
    zeta[same] = 0.0
    x = (1.0 - np.cos(zeta)) / 2.0
    with np.errstate(divide="ignore", invalid="ignore"):
        orf = 2.0 * (0.5 - x / 4.0 + 1.5 * x * np.log(x))
    orf[zeta == 0.0] = 2.0    # zeta == 0 branch with pulsar-term doubling (spharmORFbasis.py:311,327-329)
    return orf


def psr_locs_equatorial(locs):
    """(phi, colatitude) rows from RAJ[hours]/DECJ[deg] dicts (red_noise.py:205-207,223)."""
    out = np.zeros((len(locs), 2))
    for ii, loc in enumerate(locs):
        out[ii] = float(loc["RAJ"] * np.pi / 12.0), float(loc["DECJ"] * np.pi / 180.0)
    out[:, 1] = np.pi / 2.0 - out[:, 1]
    return out


def gwb_orf(psr_locs, clm=(np.sqrt(4.0 * np.pi),), lmax=0, no_correlations=False):
    """ORF = 2 * sum_k clm[k] basis[k]  (red_noise.py:200-226)."""
    if no_correlations:
        return np.diag(np.ones(len(psr_locs)) * 2)
    basis = np.array(correlated_basis(psr_locs, lmax))
    ORF = sum(clm[kk] * basis[kk] for kk in range(len(basis)))
    return ORF * 2.0


# ----------------------------------------------------------------------------------------
# GWB  (red_noise.py:138-298)
# ----------------------------------------------------------------------------------------
def gwb_grid(first_mjds, last_mjds, npts=600, howml=10):
    """start/stop/dur, the coarse grid ``ut``, the quirky dt = dur/npts and the frequency grid
    (red_noise.py:182-197, 230-232).  Nf is a knife-edge: computed exactly as the
    reference does and never re-derived elsewhere."""
    start = float(np.min([m * 86400 for m in first_mjds]) - 86400)
    stop = float(np.max([m * 86400 for m in last_mjds]) + 86400)
    dur = stop - start
    ut = np.linspace(start, stop, npts)
    dt = dur / npts
    f = np.arange(0, 1 / (2 * dt), 1 / (dur * howml))
    f[0] = f[1]
    return dict(start=start, stop=stop, dur=dur, ut=ut, dt=dt, f=f, Nf=len(f), npts=npts, howml=howml)


def lerp_sorted(xp, fp, x):
    """numpy.interp's two-point formula (what scipy.interpolate.interp1d(kind='linear') delegates to
    for 1-D float64 data): j = last index with xp[j] <= x;  slope*(x - xp[j]) + fp[j]."""
    j = np.searchsorted(xp, x, side="right") - 1
    j = np.clip(j, 0, len(xp) - 2)
    slope = (fp[j + 1] - fp[j]) / (xp[j + 1] - xp[j])
    res = slope * (x - xp[j]) + fp[j]
    res = np.where(x == xp[j], fp[j], res)
    res = np.where(x == xp[-1], fp[-1], res)
    return res


def gwb_spectrum(f, dur, howml, log10_amplitude, spectral_index, turnover=False, f0=1e-9, beta=1,
                 power=1, userSpec=None):
    """C(f) = hc^2 / (96 pi^2 f^3) * dur * howml  (red_noise.py:243-265); f1yr = 1/3.16e7 (:248)."""
    if userSpec is None:
        Amp = 10 ** log10_amplitude
        f1yr = 1 / 3.16e7
        alpha = -0.5 * (spectral_index - 3)
        hcf = Amp * (f / f1yr) ** alpha
        if turnover:
            si = alpha - beta
            hcf = hcf / (1 + (f / f0) ** (power * si)) ** (1 / power)
    else:
        lx = np.log10(userSpec[:, 0])
        ly = np.log10(userSpec[:, 1])
        lf = np.log10(f)
        v = lerp_sorted(lx, ly, np.clip(lf, lx[0], lx[-1]))
        v = np.where(lf < lx[0], ly[0], v)       # flat extrapolation (red_noise.py:23-26)
        v = np.where(lf > lx[-1], ly[-1], v)
        hcf = 10.0 ** v
    return 1 / 96 / np.pi ** 2 * hcf ** 2 / f ** 3 * dur * howml


def gwb_freq_series(M, w, C):
    """Res_f = (M @ w) * sqrt(C), DC and Nyquist zeroed (red_noise.py:268-272)."""
    Res_f = np.dot(M, w)
    Res_f = Res_f * C ** 0.5
    Res_f[:, 0] = 0
    Res_f[:, -1] = 0
    return Res_f


def gwb_time_series(Res_f, dt):
    """Hermitian packing to n = 2Nf-2 and real(ifft)/dt (red_noise.py:275-279)."""
    P, Nf = Res_f.shape
    full = np.zeros((P, 2 * Nf - 2), complex)
    full[:, :Nf] = Res_f
    full[:, Nf:] = np.conj(Res_f[:, Nf - 2:0:-1])
    return np.real(np.fft.ifft(full) / dt)


def gwb_grid_series(Res_t, npts):
    """the 600 samples actually used: Res_t[:, 10:npts+10] (red_noise.py:285)."""
    return Res_t[:, 10:npts + 10]


def gwb_dt(grid, M, w, C, toa_s_list):
    """per-pulsar GWB delays in SECONDS on each pulsar's TOAs (red_noise.py:268-287).

    (The reference stores the added signal in days, :292; callers divide by 86400.)"""
    Res_t = gwb_time_series(gwb_freq_series(M, w, C), grid["dt"])
    Res = gwb_grid_series(Res_t, grid["npts"])
    return [lerp_sorted(grid["ut"], Res[a], np.asarray(t, dtype=float)) for a, t in enumerate(toa_s_list)], Res


# ----------------------------------------------------------------------------------------
# continuous wave  (deterministic.py:13-185)
# ----------------------------------------------------------------------------------------
def cr_pow(x, y):
    """x ** y rounded ONCE from an 80-bit evaluation: the correctly rounded float64 power (up to x87 double rounding, 2^-11 of
    the cases).  NumPy's own float64 ``**`` is not correctly rounded - its array loops are SIMD kernels with errors of a few
    ulp, its scalar path is libm - which matters where the reference subtracts two nearly equal powers (deterministic.py:118)."""
    return np.asarray(np.power(np.asarray(x, dtype=np.longdouble), np.longdouble(np.float64(y))), dtype=np.float64)


def cgw_dt(mjd, ptheta, pphi, gwtheta, gwphi, mc, dist, fgw, phase0, psi, inc, pdist=1.0, pphase=None,
           psrTerm=True, evolve=True, phase_approx=False, tref=0, power=np.power):
    """CW residual on float64 MJDs for a pulsar at (ptheta colatitude, pphi) (deterministic.py:50-163).  ``power`` is the
    float64 pow used for the per-TOA frequency / phase evolution (default: NumPy's, like the reference; ``cr_pow`` = correctly
    rounded) - every other operation is IEEE basic arithmetic and identical on any machine."""
    mc = mc * SOLAR2S
    dist = dist * MPC2S
    w0 = np.pi * fgw
    phase0 = phase0 / 2
    w053 = w0 ** (-5 / 3)
    cgt, cgp, sgt, sgp = np.cos(gwtheta), np.cos(gwphi), np.sin(gwtheta), np.sin(gwphi)
    s2p, c2p = np.sin(2 * psi), np.cos(2 * psi)
    incfac1, incfac2 = 0.5 * (3 + np.cos(2 * inc)), 2 * np.cos(inc)
    m = np.array([sgp, -cgp, 0.0])
    n = np.array([-cgt * cgp, -cgt * sgp, sgt])
    omhat = np.array([-sgt * cgp, -sgt * sgp, -cgt])
    fac1 = 256 / 5 * mc ** (5 / 3) * w0 ** (8 / 3)
    fac2 = 1 / 32 / mc ** (5 / 3)
    fac3 = mc ** (5 / 3) / dist
    phat = np.array([np.sin(ptheta) * np.cos(pphi), np.sin(ptheta) * np.sin(pphi), np.cos(ptheta)])
    fplus = 0.5 * (np.dot(m, phat) ** 2 - np.dot(n, phat) ** 2) / (1 + np.dot(omhat, phat))
    fcross = (np.dot(m, phat) * np.dot(n, phat)) / (1 + np.dot(omhat, phat))
    cosMu = -np.dot(omhat, phat)
    toas = np.asarray(mjd, dtype=float) * 86400 - tref
    pd = pphase / (2 * np.pi * fgw * (1 - cosMu)) / KPC2S if pphase is not None else pdist
    pd = pd * KPC2S
    tp = toas - pd * (1 - cosMu)
    if evolve:
        omega = w0 * power(1 - fac1 * toas, -3 / 8)
        omega_p = w0 * power(1 - fac1 * tp, -3 / 8)
        phase = phase0 + fac2 * (w053 - power(omega, -5 / 3))
        phase_p = phase0 + fac2 * (w053 - power(omega_p, -5 / 3))
    elif phase_approx:
        omega = w0
        omega_p = w0 * (1 + fac1 * pd * (1 - cosMu)) ** (-3 / 8)
        phase = phase0 + omega * toas
        phase_p = phase0 + fac2 * (w053 - omega_p ** (-5 / 3)) + omega_p * toas
    else:
        omega = w0
        omega_p = omega
        phase = phase0 + omega * toas
        phase_p = phase0 + omega * tp
    At, Bt = np.sin(2 * phase) * incfac1, np.cos(2 * phase) * incfac2
    At_p, Bt_p = np.sin(2 * phase_p) * incfac1, np.cos(2 * phase_p) * incfac2
    alpha = fac3 / omega ** (1 / 3)
    alpha_p = fac3 / omega_p ** (1 / 3)
    rplus = alpha * (At * c2p + Bt * s2p)
    rcross = alpha * (-At * s2p + Bt * c2p)
    rplus_p = alpha_p * (At_p * c2p + Bt_p * s2p)
    rcross_p = alpha_p * (-At_p * s2p + Bt_p * c2p)
    if psrTerm:
        return fplus * (rplus_p - rplus) + fcross * (rcross_p - rcross)
    return -fplus * rplus - fcross * rcross


def cw_catalog_dt(mjd, ptheta, pphi, gwtheta_list, gwphi_list, mc_list, dist_list, fgw_list, phase0_list, psi_list, inc_list,
                  pdist=1.0, pphase=None, psrTerm=True, evolve=True, phase_approx=False, tref=0, power=np.power):
    """sum over sources of the single-source waveform with NaN -> 0, in catalogue order (add_catalog_of_cws and
    loop_over_CWs, deterministic.py:188-318, :443-561; the parallel variant :321-440 sums the same terms)."""
    res = np.zeros(len(mjd))
    for i in range(len(mc_list)):
        with np.errstate(invalid="ignore"):
            rrr = cgw_dt(mjd, ptheta, pphi, gwtheta_list[i], gwphi_list[i], mc_list[i], dist_list[i], fgw_list[i], phase0_list[i],
                         psi_list[i], inc_list[i], pdist=pdist, pphase=pphase, psrTerm=psrTerm, evolve=evolve,
                         phase_approx=phase_approx, tref=tref, power=power)
        res += np.where(np.isnan(rrr), 0.0, rrr)
    return res


# ----------------------------------------------------------------------------------------
# time-domain ("TD") mode oracle: dense covariance -> Cholesky -> L z   
# ----------------------------------------------------------------------------------------
def td_covariance(toas_s, log10_amplitude, spectral_index, components, sigma2_wn, epoch_of, ecorr_epoch):
    """C = F diag(phi) F^T + diag(sigma_wn^2) + sum_e ecorr_e^2 1_e 1_e^T, the covariance implied by the
    reference's RN synthesis (red_noise.py:98-101,126-128) plus its white/ECORR terms
    (white_noise.py:105-109,182).  F uses the default (sin-first, absolute-t) convention."""
    Tspan = toas_s.max() - toas_s.min()
    F, freqs = fourier_design_matrix(toas_s, nmodes=components, Tspan=Tspan)
    phi = red_noise_prior(freqs, log10_amplitude, spectral_index, Tspan)
    Cm = (F * phi[None, :]) @ F.T
    Cm[np.diag_indices_from(Cm)] += sigma2_wn
    if epoch_of is not None:
        e2 = ecorr_epoch[epoch_of] ** 2
        same = epoch_of[:, None] == epoch_of[None, :]
        Cm += same * e2[:, None]
    return Cm


def td_covariance_rows(toas_s, log10_amplitude, spectral_index, components, sigma2_wn, epoch_of, ecorr_epoch, rows):
    """rows `rows` of td_covariance() without forming the N x N matrix (N = 35 037 for BASELINE.json config 2's largest pulsar:
    9.8 GB): [len(rows), N].  log10_amplitude None = no red noise."""
    toas_s = np.asarray(toas_s, dtype=np.float64)
    rows = np.asarray(rows)
    out = np.zeros((len(rows), len(toas_s)))
    if log10_amplitude is not None:
        Tspan = toas_s.max() - toas_s.min()
        F, freqs = fourier_design_matrix(toas_s, nmodes=components, Tspan=Tspan)
        phi = red_noise_prior(freqs, log10_amplitude, spectral_index, Tspan)
        out += (F[rows] * phi[None, :]) @ F.T
    out[np.arange(len(rows)), rows] += np.asarray(sigma2_wn)[rows]
    if epoch_of is not None:
        e2 = ecorr_epoch[epoch_of] ** 2
        out += (epoch_of[rows][:, None] == epoch_of[None, :]) * e2[rows][:, None]
    return out


def td_draw(Cm, z):
    """L z with L = cholesky(C) lower; z is [N] or [N, R]."""
    return np.linalg.cholesky(Cm) @ z


def td_gwb_grid_covariance(grid, C, i0=10):
    """Covariance of the npts grid samples Res_t[a, i0:i0+npts] of ONE pulsar with unit ORF, implied by the reference's
    frequency-domain synthesis (red_noise.py:265-285):
        Sigma[p, q] = 4 / (dt^2 n^2) * sum_{k=1..Nf-2} C_k cos(2 pi k (p - q) / n),   n = 2 Nf - 2
    (w_k = x + i y with x, y ~ N(0,1); DC and Nyquist bins zeroed, :271-272).  Toeplitz in (p - q); the crop offset i0 drops
    out.  Between pulsars a, b the covariance is ORF_ab * Sigma (the mix with M = cholesky(ORF), :235,268)."""
    Nf, npts, dt = grid["Nf"], grid["npts"], grid["dt"]
    n = 2 * Nf - 2
    k = np.arange(1, Nf - 1)
    lag = np.arange(npts)
    col = 4.0 / (dt ** 2 * n ** 2) * (np.cos(2 * np.pi * np.outer(lag, k) / n) @ C[1:Nf - 1])
    idx = np.abs(lag[:, None] - lag[None, :])
    return col[idx]


def cholesky_longdouble(S):
    """reference-quality lower Cholesky factor in x87 extended precision (64-bit mantissa), outer-product form: the yardstick
    against which the forward error of a float64 factorisation (LAPACK's or the device's) is measured when the matrix is
    too ill-conditioned for two float64 factorisations to agree with each other."""
    A = np.array(S, dtype=np.longdouble)
    n = A.shape[0]
    L = np.zeros_like(A)
    for j in range(n):
        d = A[j, j]
        if not d > 0:
            raise np.linalg.LinAlgError(f"leading minor {j + 1} not positive definite")
        r = np.sqrt(d)
        L[j, j] = r
        col = A[j + 1:, j] / r
        L[j + 1:, j] = col
        A[j + 1:, j + 1:] -= np.outer(col, col)
    return L


def td_realisation(toa_s_list, td_cov_list, z_td, grid=None, Lg=None, M=None, z_gw=None, det=None):
    """one TD-mode realisation on the CPU: per pulsar cholesky(C_a) @ z_a (+ the GWB term: grid series Lg @ z_gw[a] per
    pulsar, mixed with M = cholesky(ORF) and interpolated onto the TOAs like red_noise.py:268,286-287) (+ det)."""
    P = len(td_cov_list)
    out = [td_draw(td_cov_list[a], z_td[a]) for a in range(P)]
    if Lg is not None:
        G0 = z_gw @ Lg.T                     # [P, npts]: row a = Lg @ z_gw[a]
        G = M @ G0
        for a in range(P):
            out[a] = out[a] + lerp_sorted(grid["ut"], G[a], np.asarray(toa_s_list[a], dtype=float))
    if det is not None:
        out = [o + d for o, d in zip(out, det)]
    return out
