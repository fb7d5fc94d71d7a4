"""CPU checks of the device math: the __host__ __device__ headers under pta_replicator_amd/csrc are
compiled with g++ (tests/hostcheck/hostcheck.cpp) and compared with known answers, the oracle and the
reference-generated golden vectors.  The same formulas run on the GPU in the -m gpu tests."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from helpers import load
from oracle import philox_ref
from oracle import pta_oracle as po

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def hc(tmp_path_factory):
    out = tmp_path_factory.mktemp("hostcheck") / "libhostcheck.so"
    src = os.path.join(HERE, "hostcheck", "hostcheck.cpp")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-shared", "-fPIC", src, "-o", str(out)])
    lib = ctypes.CDLL(str(out))
    lib.hc_stream_id.restype = ctypes.c_uint32
    return lib


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


# Random123 known-answer vectors for philox4x32-10 (kat_vectors of the Random123 distribution)
KAT = [
    ([0, 0, 0, 0], [0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
    ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
    ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0], [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]),
]


@pytest.mark.parametrize("ctr,key,expect", KAT)
def test_philox_known_answers(hc, ctr, key, expect):
    c = np.array(ctr, dtype=np.uint32); k = np.array(key, dtype=np.uint32); o = np.zeros(4, dtype=np.uint32)
    hc.hc_philox(_p(c, ctypes.c_uint32), _p(k, ctypes.c_uint32), _p(o, ctypes.c_uint32))
    assert [int(x) for x in o] == expect
    assert [int(x) for x in philox_ref.philox4x32_10(c[None, :], k)[0]] == expect


def test_normals_match_numpy_twin(hc):
    """device formula (compiled for the host) == the NumPy twin used to replay throughput-mode draws."""
    seed, r, npairs = 0x1234567890ABCDEF, 7, 4096
    stream = hc.hc_stream_id(3, 11)
    assert stream == philox_ref.stream_id(3, 11)
    out = np.zeros(2 * npairs)
    hc.hc_normal_pairs(ctypes.c_uint64(seed), ctypes.c_uint64(r), ctypes.c_uint32(stream), npairs, _p(out, ctypes.c_double))
    z0, z1 = philox_ref.normal_pairs(seed, r, stream, npairs)
    assert np.max(np.abs(out[0::2] - z0)) < 1e-14 and np.max(np.abs(out[1::2] - z1)) < 1e-14
    u = np.zeros(2 * npairs)
    hc.hc_uniform_pairs(ctypes.c_uint64(seed), ctypes.c_uint64(r), ctypes.c_uint32(stream), npairs, _p(u, ctypes.c_double))
    u1, u2 = philox_ref.uniform_pairs(seed, r, stream, npairs)
    assert np.array_equal(u[0::2], u1) and np.array_equal(u[1::2], u2)
    assert u[0::2].min() > 0.0 and u[0::2].max() <= 1.0 and u[1::2].min() >= 0.0 and u[1::2].max() < 1.0


def _ulps(got, ref):
    """|got - ref| in units of the spacing of doubles at ref (ref: longdouble)"""
    L = np.longdouble
    r64 = ref.astype(np.float64)
    sp = np.spacing(np.where(r64 == 0, np.finfo(np.float64).tiny, np.abs(r64))).astype(L)
    return (np.abs(got.astype(L) - ref) / sp).astype(np.float64)


def test_table_driven_neg2log_accuracy(hc):
    """-2 ln u of the product path (128-entry table + degree-5 remainder, pta_rng.h) against an 80-bit evaluation: < 1.5 ulp
    everywhere, including the interval boundaries of the table, the re-centring boundary (j = 52 | 53) and u -> 1, where the
    result must stay accurate RELATIVE to itself (the Box-Muller radius goes to zero there)."""
    L = np.longdouble
    if np.finfo(L).nmant < 63:
        pytest.skip("no 80-bit long double on this host")
    rng = np.random.default_rng(5)
    u = np.concatenate([
        1.0 - rng.random(400000),                                   # (0, 1]
        np.exp(-rng.uniform(0, 36, 100000)),                        # down to 2^-52
        1.0 - np.exp(-rng.uniform(0, 36, 100000)),                  # up against 1
        np.array([1.0, 1.0 - 2.0 ** -53, 1.0 - 2.0 ** -52, 0.5, 0.25, 2.0 ** -52, 2.0 ** -51, 0.70710678118654752, 0.7071067811865476]),
        np.ldexp(1.0 + np.arange(0, 129) / 128.0, -1)[:-1],         # every table boundary, and one ulp either side of it
        np.nextafter(np.ldexp(1.0 + np.arange(1, 128) / 128.0, -1), 0.0), np.nextafter(np.ldexp(1.0 + np.arange(0, 128) / 128.0, -1), 1.0),
    ])
    u = u[(u > 0) & (u <= 1)]
    out = np.zeros_like(u); pol = np.zeros_like(u)
    hc.hc_neg2log(_p(u, ctypes.c_double), len(u), 0, _p(out, ctypes.c_double))
    hc.hc_neg2log(_p(u, ctypes.c_double), len(u), 1, _p(pol, ctypes.c_double))
    ref = L(-2) * np.log(u.astype(L))
    nz = u < 1.0
    assert _ulps(out[nz], ref[nz]).max() < 1.5
    assert _ulps(pol[nz], ref[nz]).max() < 1.5          # the polynomial cross-check is held to the same bar
    assert abs(out[~nz]).max() < 1e-18                  # u = 1: zero up to the rounding of one table entry; the caller clamps before sqrt
    assert np.all(out[nz] > 0)


def test_table_driven_sincos_accuracy(hc):
    """sin / cos(2 pi u) of the product path (32-entry table + Taylor remainder + rotation): within 4.5 ulp (the rotation sums two rounded products), |error| <= 2.3e-16
    absolute, and exact zeros / ones at the quarter turns."""
    L = np.longdouble
    if np.finfo(L).nmant < 63:
        pytest.skip("no 80-bit long double on this host")
    rng = np.random.default_rng(6)
    u = np.concatenate([rng.random(500000), np.arange(0, 64) / 64.0, np.nextafter(np.arange(1, 64) / 64.0, 0.0), np.nextafter(np.arange(0, 64) / 64.0, 1.0),
                        np.array([1.0 - 2.0 ** -52, 2.0 ** -52, 0.5 - 2.0 ** -53, 0.25 + 2.0 ** -54])])
    sn, cs = np.zeros_like(u), np.zeros_like(u)
    hc.hc_sincos_2pi(_p(u, ctypes.c_double), len(u), 0, _p(sn, ctypes.c_double), _p(cs, ctypes.c_double))
    # reference: exact reduction to the nearest quarter turn in double, remainder through the 80-bit libm
    q = np.rint(4 * u); x = (u.astype(L) - L(0.25) * q.astype(L)) * (L(2) * np.arccos(L(-1)))
    sr, cr = np.sin(x), np.cos(x); k = q.astype(np.int64) & 3
    rs = np.choose(k, [sr, cr, -sr, -cr]); rc = np.choose(k, [cr, -sr, -cr, sr])
    assert max(_ulps(sn, rs).max(), _ulps(cs, rc).max()) < 4.5
    assert max(np.abs(sn.astype(L) - rs).max(), np.abs(cs.astype(L) - rc).max()) < 2.3e-16
    for uq, s0, c0 in ((0.0, 0.0, 1.0), (0.25, 1.0, 0.0), (0.5, 0.0, -1.0), (0.75, -1.0, 0.0)):
        i = int(np.where(u == uq)[0][0])
        assert sn[i] == s0 and cs[i] == c0


def test_deviates_within_a_few_ulp_of_an_80_bit_box_muller(hc):
    L = np.longdouble
    if np.finfo(L).nmant < 63:
        pytest.skip("no 80-bit long double on this host")
    seed, r, stream, npairs = 77, 5, philox_ref.stream_id(3, 11), 1 << 18
    out = np.zeros(2 * npairs)
    hc.hc_normal_pairs(ctypes.c_uint64(seed), ctypes.c_uint64(r), ctypes.c_uint32(stream), npairs, _p(out, ctypes.c_double))
    u1, u2 = philox_ref.uniform_pairs(seed, r, stream, npairs)
    rad = np.sqrt(L(-2) * np.log(u1.astype(L)))
    q = np.rint(4 * u2); x = (u2.astype(L) - L(0.25) * q.astype(L)) * (L(2) * np.arccos(L(-1)))
    sr, cr = np.sin(x), np.cos(x); k = q.astype(np.int64) & 3
    s = np.choose(k, [sr, cr, -sr, -cr]); c = np.choose(k, [cr, -sr, -cr, sr])
    e = np.maximum(_ulps(out[0::2], rad * c), _ulps(out[1::2], rad * s))
    assert e.max() < 6.0 and np.quantile(e, 0.999) < 3.5


def test_normals_are_standard_normal(hc):
    npairs = 200000
    out = np.zeros(2 * npairs)
    hc.hc_normal_pairs(ctypes.c_uint64(99), ctypes.c_uint64(0), ctypes.c_uint32(1 << 24), npairs, _p(out, ctypes.c_double))
    n = out.size
    assert abs(out.mean()) < 5 / np.sqrt(n)
    assert abs(out.var() - 1) < 5 * np.sqrt(2 / n)
    assert abs(np.mean(out ** 4) - 3) < 5 * np.sqrt(96 / n)
    assert abs(np.corrcoef(out[0::2], out[1::2])[0, 1]) < 5 / np.sqrt(npairs)
    assert abs(np.corrcoef(out[:-1], out[1:])[0, 1]) < 5 / np.sqrt(n)


def test_orf_basis_device_math_vs_reference_golden(hc):
    z = load("orf_basis.npz")
    locs = np.ascontiguousarray(z["psr_locs"], dtype=np.float64)
    P, lmax = len(locs), int(z["lmax"])
    basis = np.zeros(((lmax + 1) ** 2, P, P))
    hc.hc_orf_basis(_p(locs, ctypes.c_double), P, lmax, _p(basis, ctypes.c_double))
    ref = z["basis"]
    scale = np.max(np.abs(ref), axis=(1, 2), keepdims=True)
    assert np.max(np.abs(basis - ref) / scale) < 1e-11
    orf = np.zeros((P, P))
    hc.hc_orf_hd(_p(locs, ctypes.c_double), P, _p(orf, ctypes.c_double))
    assert np.max(np.abs(orf - 2 * np.sqrt(4 * np.pi) * ref[0])) < 1e-14
    assert np.max(np.abs(orf - po.hd_orf_closed_form(locs))) < 1e-14


def test_orf_config5_all_pairs_device_math_vs_reference_fixture(hc):
    """BASELINE.json config 5 at full P: the device ORF formulas (host twin) on all 20 100 pairs against the UNMODIFIED reference's
    values (tests/golden/c5_orf_lmax4.npz), degree by degree.  l <= 3 at 1e-12 absolute; l = 4 at 2e-12 up to 170 degrees of
    separation and 2e-10 on the near-antipodal pairs (next test)."""
    z = load("c5_orf_lmax4.npz")
    P, lmax, clm = 200, int(z["lmax"]), z["clm"]
    locs = np.ascontiguousarray(po.psr_locs_equatorial([{"RAJ": z["raj"][a], "DECJ": z["decj"][a]} for a in range(P)]))
    basis = np.zeros(((lmax + 1) ** 2, P, P))
    hc.hc_orf_basis(_p(locs, ctypes.c_double), P, lmax, _p(basis, ctypes.c_double))
    zeta = np.array([[po.calczeta(locs[a, 0], locs[b, 0], locs[a, 1], locs[b, 1]) for b in range(P)] for a in range(P)])
    for ll in range(lmax + 1):
        err = np.abs(2 * np.tensordot(clm[ll * ll:(ll + 1) ** 2], basis[ll * ll:(ll + 1) ** 2], axes=1) - z["orf_l"][ll])
        if ll < 4:
            assert err.max() < 1e-12, (ll, err.max())
        else:
            assert err[np.degrees(zeta) <= 170.0].max() < 2e-12 and err.max() < 2e-10
    assert np.max(np.abs(2 * np.tensordot(clm, basis, axes=1) - z["orf"])) < 2e-10


def test_orf_l4_near_antipodal_pairs_are_ill_conditioned_in_the_reference(hc):
    """Why the l = 4 basis is held to 2e-10 (not 1e-12) on near-antipodal pairs: at the worst pair of config 5 (pulsars 25 / 33,
    174.86 degrees apart) the reference's computational-frame sums (spharmORFbasis.py:43-248) cancel ~6 digits.  Evaluated with
    the reference's formula in float64 - the restatement is bit-identical to the reference there - and in longdouble, the
    reference's OWN result is 1e-10 from the better one; the device formulas land 2.3e-10 from it, 1.4e-10 from the reference."""
    z = load("c5_orf_lmax4.npz")
    locs = np.ascontiguousarray(po.psr_locs_equatorial([{"RAJ": z["raj"][a], "DECJ": z["decj"][a]} for a in (25, 33)]))
    p1, p2, t1, t2 = locs[0, 0], locs[1, 0], locs[0, 1], locs[1, 1]
    zeta, ll = po.calczeta(p1, p2, t1, t2), 4
    assert abs(np.degrees(zeta) - 174.857) < 1e-3

    def row(zz):
        plus = [po.arbCompFrame_ORF(mm, ll, zz) for mm in range(ll + 1)]
        gamma_ml = [(-1) ** mm * plus[mm] for mm in range(1, ll + 1)][::-1] + plus
        return np.array([float(po.real_rotated_Gammas(mi - ll, ll, p1, p2, t1, t2, gamma_ml)) for mi in range(2 * ll + 1)])
    b64, b80 = row(zeta), row(np.longdouble(zeta))
    basis = np.zeros((25, 2, 2))
    hc.hc_orf_basis(_p(locs, ctypes.c_double), 2, 4, _p(basis, ctypes.c_double))
    dev = basis[16:25, 0, 1]
    ref_noise, dev_err = np.max(np.abs(b64 - b80)), np.max(np.abs(dev - b80))
    assert 5e-11 < ref_noise < 3e-10, ref_noise           # the reference's own float64 rounding at this pair
    assert dev_err < 4e-10 and np.max(np.abs(dev - b64)) < 2e-10


def test_orf_hd_headline_array(hc):
    """68 isotropic pulsars (config 3 geometry): closed form == general basis, and is positive definite."""
    rng = np.random.default_rng(68)
    locs = np.stack([rng.uniform(0, 24, 68) * np.pi / 12, np.pi / 2 - np.arcsin(rng.uniform(-1, 1, 68))], axis=1)
    orf = np.zeros((68, 68))
    hc.hc_orf_hd(_p(locs, ctypes.c_double), 68, _p(orf, ctypes.c_double))
    assert np.max(np.abs(orf - po.hd_orf_closed_form(locs))) < 1e-14
    basis = np.zeros((1, 68, 68))
    hc.hc_orf_basis(_p(locs, ctypes.c_double), 68, 0, _p(basis, ctypes.c_double))
    assert np.max(np.abs(2 * np.sqrt(4 * np.pi) * basis[0] - orf)) < 1e-13
    np.linalg.cholesky(orf)


def _digit_reverse8(k):
    k = np.asarray(k)
    return ((k & 7) << 9) | (((k >> 3) & 7) << 6) | (((k >> 6) & 7) << 3) | ((k >> 9) & 7)


def test_fft4096_passes_against_numpy(hc):
    """the in-LDS radix-8 FFT used by the chirp-z GWB kernel: forward = numpy fft in digit-reversed order,
    inverse(forward(x)) = 4096 x, linear convolution through it = numpy's."""
    rng = np.random.default_rng(8)
    n, plane = 4096, hc.hc_fft_plane()
    phys = np.array([hc.hc_fft_phys(i) for i in range(n)])
    m = np.arange(n)
    tw = np.stack([np.cos(2 * np.pi * m / n), -np.sin(2 * np.pi * m / n)], axis=1).ravel().copy()
    x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    re, im = np.zeros(plane), np.zeros(plane)
    re[phys], im[phys] = x.real, x.imag
    hc.hc_fft_forward(_p(re, ctypes.c_double), _p(im, ctypes.c_double), _p(tw, ctypes.c_double))
    X = np.fft.fft(x)
    got = re[phys] + 1j * im[phys]                    # logical order of the buffer = digit-reversed frequencies
    assert np.max(np.abs(got[_digit_reverse8(m)] - X)) < 1e-11 * np.max(np.abs(X))
    hc.hc_fft_inverse(_p(re, ctypes.c_double), _p(im, ctypes.c_double), _p(tw, ctypes.c_double))
    back = (re[phys] + 1j * im[phys]) / n
    assert np.max(np.abs(back - x)) < 1e-13
