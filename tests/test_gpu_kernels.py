"""GPU tests of the C-ABI building blocks (run with -m gpu on an MI355X).

Every test calls the HIP library through the ctypes boundary and compares with NumPy / the oracle."""
import ctypes

import numpy as np
import pytest

from helpers import load
from oracle import philox_ref
from oracle import pta_oracle as po

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch
    from pta_replicator_amd import _lib, device as dv
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return dict(torch=torch, lib=_lib, dv=dv, s=dv.stream_ptr())


def test_device_and_mfma_layout(gpu):
    info = gpu["dv"].device_info()
    assert info["wavefront"] == 64 and info["arch"].startswith("gfx950"), info
    err = ctypes.c_double(-1.0)
    gpu["lib"].call("pta_selftest_mfma_f64", ctypes.byref(err))
    assert err.value < 1e-12


def test_philox_known_answers_on_device(gpu):
    dv, lib, torch = gpu["dv"], gpu["lib"], gpu["torch"]
    kat = [([0, 0, 0, 0], [0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
           ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
           ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0], [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1])]
    for ctr, key, expect in kat:
        c = torch.tensor(np.array(ctr, dtype=np.uint32).view(np.int32), device="cuda")
        k = torch.tensor(np.array(key, dtype=np.uint32).view(np.int32), device="cuda")
        o = torch.zeros(4, dtype=torch.int32, device="cuda")
        lib.call("pta_rng_philox_raw", dv.ptr(c), dv.ptr(k), 1, dv.ptr(o), gpu["s"])
        assert [int(x) for x in o.cpu().numpy().view(np.uint32)] == expect


@pytest.mark.parametrize("interleave", [0, 1])
def test_rng_fill_matches_numpy_twin(gpu, interleave):
    dv, lib = gpu["dv"], gpu["lib"]
    seed, r0, R, npairs = 0xDEADBEEF12345678, (1 << 33) + 5, 3, 1000
    sid = philox_ref.stream_id(philox_ref.STREAM_WN, 17)
    if interleave:
        z0 = dv.empty((R, 2 * npairs)); z1 = None
        lib.call("pta_rng_fill_normal", seed, r0, R, sid, npairs, 1, dv.ptr(z0), None, 2 * npairs, 0, gpu["s"])
        got0, got1 = z0.cpu().numpy()[:, 0::2], z0.cpu().numpy()[:, 1::2]
    else:
        z0, z1 = dv.empty((R, npairs)), dv.empty((R, npairs))
        lib.call("pta_rng_fill_normal", seed, r0, R, sid, npairs, 0, dv.ptr(z0), dv.ptr(z1), npairs, 0, gpu["s"])
        got0, got1 = z0.cpu().numpy(), z1.cpu().numpy()
    for r in range(R):
        e0, e1 = philox_ref.normal_pairs(seed, r0 + r, sid, npairs)
        assert np.max(np.abs(got0[r] - e0)) < 1e-13 and np.max(np.abs(got1[r] - e1)) < 1e-13


def test_rng_fill_blocks_equals_the_per_block_fill(gpu):
    """pta_rng_fill_normal_blocks: every block of a TD plan in one launch - bit-identical to pta_rng_fill_normal per block (stream
    (kind, b)), odd orders (the pair is written whole), nothing written outside a block's columns."""
    dv, lib, torch = gpu["dv"], gpu["lib"], gpu["torch"]
    seed, r0, R, kind = 0x1234ABCD, (1 << 32) + 7, 5, 5
    ns = [7, 300, 1, 1025, 64]
    zoff = np.concatenate([[0], np.cumsum([(n + 1) // 2 * 2 + 4 for n in ns])]).astype(np.int64)   # 4 spare columns behind every block
    ld = int(zoff[-1]) + 16
    z = dv.empty((R, ld)); z.fill_(-7.0)
    d_n, d_off = dv.i32(ns), dv.i32(zoff[:-1])
    lib.call("pta_rng_fill_normal_blocks", seed, r0, R, kind, len(ns), dv.ptr(d_n), dv.ptr(d_off), max(ns), dv.ptr(z), ld, 0, gpu["s"])
    got = z.cpu().numpy()
    want = np.full((R, ld), -7.0)
    for b, n in enumerate(ns):
        npair = (n + 1) // 2
        zb = dv.empty((R, 2 * npair))
        lib.call("pta_rng_fill_normal", seed, r0, R, philox_ref.stream_id(kind, b), npair, 1, dv.ptr(zb), None, 2 * npair, 0, gpu["s"])
        want[:, zoff[b]:zoff[b] + 2 * npair] = zb.cpu().numpy()
    assert np.array_equal(got, want)


def test_device_deviates_within_a_few_ulp_of_an_80_bit_box_muller(gpu):
    """the table-driven fp64 transform on the device (tables staged in LDS): every deviate within 6 ulp of a long double
    evaluation of the same uniforms, 99.9 % within 3.5 (scripts/gpu_rng_accuracy.py prints the measured figures)."""
    dv, lib = gpu["dv"], gpu["lib"]
    L = np.longdouble
    if np.finfo(L).nmant < 63:
        pytest.skip("no 80-bit long double on this host")
    seed, real, npairs = 77, 5, 1 << 18
    sid = philox_ref.stream_id(3, 11)
    z = dv.empty((2 * npairs,))
    lib.call("pta_rng_fill_normal", seed, real, 1, sid, npairs, 1, dv.ptr(z), None, 2 * npairs, 0, gpu["s"])
    got = z.cpu().numpy()
    u1, u2 = philox_ref.uniform_pairs(seed, real, sid, npairs)
    rad = np.sqrt(L(-2) * np.log(u1.astype(L)))
    q = np.rint(4 * u2); x = (u2.astype(L) - L(0.25) * q.astype(L)) * (L(2) * np.arccos(L(-1)))
    sr, cr = np.sin(x), np.cos(x); k = q.astype(np.int64) & 3
    refs = (rad * np.choose(k, [cr, -sr, -cr, sr]), rad * np.choose(k, [sr, cr, -sr, -cr]))
    worst = []
    for g, r in zip((got[0::2], got[1::2]), refs):
        e = (np.abs(g.astype(L) - r) / np.spacing(np.abs(r.astype(np.float64))).astype(L)).astype(np.float64)
        worst.append(e)
    e = np.maximum(*worst)
    assert e.max() < 6.0 and np.quantile(e, 0.999) < 3.5


def test_fast_rng_math_mode(gpu):
    """opt-in fp32-transcendental Gaussian transform: same uniforms, deviates within ~1e-6 of the fp64 ones, unit variance."""
    dv, lib = gpu["dv"], gpu["lib"]
    seed, npairs = 4242, 200000
    sid = philox_ref.stream_id(philox_ref.STREAM_WN, 3)
    acc, fast = dv.empty((2 * npairs,)), dv.empty((2 * npairs,))
    lib.call("pta_rng_fill_normal", seed, 9, 1, sid, npairs, 1, dv.ptr(acc), None, 2 * npairs, 0, gpu["s"])
    lib.call("pta_rng_fill_normal", seed, 9, 1, sid, npairs, 1, dv.ptr(fast), None, 2 * npairs, 1, gpu["s"])   # rng_fast = 1, per call
    a, f = acc.cpu().numpy(), fast.cpu().numpy()
    assert np.max(np.abs(a - f)) < 2e-5 and np.sqrt(np.mean((a - f) ** 2)) < 2e-6
    z0, z1 = philox_ref.normal_pairs(seed, 9, sid, npairs, fast=True)
    assert np.max(np.abs(f[0::2] - z0)) < 2e-5 and np.max(np.abs(f[1::2] - z1)) < 2e-5
    assert abs(f.var() - 1) < 5 * np.sqrt(2 / f.size) and abs(np.mean(f ** 4) - 3) < 5 * np.sqrt(96 / f.size)


@pytest.mark.parametrize("shape", [(1, 1, 1), (5, 7, 3), (64, 64, 16), (70, 130, 37), (200, 600, 301), (3, 600, 2998), (300, 257, 75), (515, 400, 64)])
@pytest.mark.parametrize("transB", [0, 1])
def test_dgemm_mfma_and_valu(gpu, shape, transB):
    dv, lib = gpu["dv"], gpu["lib"]
    M, N, K = shape
    rng = np.random.default_rng(M * 1000 + N + K)
    A = rng.standard_normal((M, K)); B = rng.standard_normal((N, K) if transB else (K, N)); C0 = rng.standard_normal((M, N))
    ref = 0.7 * A @ (B.T if transB else B) - 1.3 * C0
    for algo in (0, 1):
        Ad, Bd, Cd = dv.f64(A), dv.f64(B), dv.f64(C0)
        lib.call("pta_dgemm", transB, M, N, K, 0.7, dv.ptr(Ad), K, 1, dv.ptr(Bd), B.shape[1], -1.3, dv.ptr(Cd), N, 0, 1, 0, 0, 0,
                 algo, gpu["s"])
        assert np.max(np.abs(Cd.cpu().numpy() - ref)) < 1e-11 * max(1.0, np.max(np.abs(ref))), (algo, shape)


@pytest.mark.parametrize("shape", [(256, 128, 32), (300, 257, 76), (515, 400, 64), (1000, 520, 1032), (640, 640, 250), (768, 384, 16), (512, 256, 8)])
@pytest.mark.parametrize("lower", [0, 1])
def test_dgemm_lds_dma_operand_staging(gpu, shape, lower):
    """pta_dgemm algo 2 (k_dgemm_glds128: operand slabs by global_load_lds, swizzled LDS image, permuted k slots) against NumPy and
    against algo 1, on operands that are SUB-BLOCKS of wider arrays whose other columns hold NaN - a K tail (K % 16 != 0) must not
    pick up what lies behind column K, rows past M / N are clamped duplicates that are never stored; batch of 2, square lower_only."""
    dv, lib = gpu["dv"], gpu["lib"]
    M, N, K = shape
    if lower:
        N = M
    rng = np.random.default_rng(M + 7 * N + 13 * K)
    lda, ldb, Bt = K + 22, K + 6, 2
    A = np.full((Bt, M, lda), np.nan); Bm = np.full((Bt, N, ldb), np.nan)
    A[:, :, 4:4 + K] = rng.standard_normal((Bt, M, K)); Bm[:, :, 2:2 + K] = rng.standard_normal((Bt, N, K))
    C0 = rng.standard_normal((Bt, M, N))
    ref = 0.7 * A[:, :, 4:4 + K] @ Bm[:, :, 2:2 + K].transpose(0, 2, 1) - 1.3 * C0
    got = {}
    for algo in (1, 2, 3):        # 3 = the same kernel with the C-tile prefetch epilogue (A/B form, PTA_POTRF_EPI1)
        Ad, Bd, Cd = dv.f64(A), dv.f64(Bm), dv.f64(C0)
        lib.call("pta_dgemm", 1, M, N, K, 0.7, ctypes.c_void_p(Ad.data_ptr() + 8 * 4), lda, 1, ctypes.c_void_p(Bd.data_ptr() + 8 * 2), ldb, -1.3,
                 dv.ptr(Cd), N, lower, Bt, M * lda, N * ldb, M * N, algo, gpu["s"])
        got[algo] = Cd.cpu().numpy()
        sel = np.tril(np.ones((M, N), bool)) if lower else np.ones((M, N), bool)
        assert np.all(np.isfinite(got[algo]))
        assert np.max(np.abs(got[algo][:, sel] - ref[:, sel])) < 1e-11 * max(1.0, np.max(np.abs(ref))), (algo, shape)
        if lower:
            assert np.array_equal(got[algo][:, ~sel], C0[:, ~sel])


def test_dgemm_batched_strided_lower(gpu):
    """batch + element stride on A (planes of interleaved complex) + SYRK-style lower_only."""
    dv, lib = gpu["dv"], gpu["lib"]
    rng = np.random.default_rng(9)
    Bt, M, K = 3, 97, 50
    W = rng.standard_normal((Bt, M, K, 2))
    C0 = rng.standard_normal((Bt, M, M))
    for algo in (0, 1):
        Wd, Cd, Wre = dv.f64(W), dv.f64(C0), dv.f64(W[..., 0])
        # C -= Re(W) Re(W)^T on the lower triangle
        lib.call("pta_dgemm", 1, M, M, K, -1.0, dv.ptr(Wd), 2 * K, 2, dv.ptr(Wre), K, 1.0, dv.ptr(Cd), M, 1, Bt,
                 M * K * 2, M * K, M * M, algo, gpu["s"])
        got = Cd.cpu().numpy()
        for b in range(Bt):
            ref = C0[b] - W[b, :, :, 0] @ W[b, :, :, 0].T
            lo = np.tril_indices(M)
            up = np.triu_indices(M, 1)
            assert np.max(np.abs(got[b][lo] - ref[lo])) < 1e-11
            assert np.array_equal(got[b][up], C0[b][up])


def test_orf_kernels_vs_reference_golden(gpu):
    from pta_replicator_amd import spharmORFbasis as anis
    z = load("orf_basis.npz")
    locs, lmax, ref = z["psr_locs"], int(z["lmax"]), z["basis"]
    basis = np.array(anis.correlated_basis(locs, lmax))
    scale = np.max(np.abs(ref), axis=(1, 2), keepdims=True)
    err = np.abs(basis - ref) / scale
    # every l at 1e-12 of the matrix scale (measured 3e-15 at l = 4).  The l >= 3 sums are ill-conditioned (~1e5-1e6) in cos(zeta)
    # and in the integer powers: with zeta / cos(zeta) taken from the host exactly as the reference computes them and the powers
    # correctly rounded (double-double product chain) the device reproduces the reference's own rounding; round 1 (device acos /
    # cos / pow, 1-2 ulp each) sat at 7e-10 there.
    assert err.max() < 1e-12, [float(err[l * l:(l + 1) ** 2].max()) for l in range(lmax + 1)]
    orf = anis.orf_from_locations(locs).cpu().numpy()
    assert np.max(np.abs(orf - 2 * np.sqrt(4 * np.pi) * ref[0])) < 1e-14
    clm = np.array([np.sqrt(4 * np.pi), 0.3, -0.2, 0.25])
    orf2 = anis.orf_from_locations(locs, clm, 1).cpu().numpy()
    assert np.max(np.abs(orf2 - 2 * sum(clm[k] * ref[k] for k in range(4)))) < 1e-13


@pytest.mark.parametrize("n,batch", [(1, 1), (3, 2), (64, 1), (65, 2), (68, 1), (130, 4), (131, 1), (200, 2), (515, 1), (512, 2), (1000, 1), (1001, 1), (1025, 2),
                                     (1338, 3)])   # 65 / 1025: a first block 1 column wide; 131 / 1001: odd widths (3, 41)
def test_potrf_batched_vs_numpy(gpu, n, batch):
    from pta_replicator_amd import red_noise as rn
    dv, lib = gpu["dv"], gpu["lib"]
    rng = np.random.default_rng(n)
    X = rng.standard_normal((batch, n, n + 5))
    A = X @ X.transpose(0, 2, 1) + 0.1 * np.eye(n)
    # VALU cross-check, default (MFMA product solves), substitution solves, one chain, 256-wide panels on 3 / 4 chains, and the
    # register-staged operand slabs of the 128-tile updates (the default stages them by LDS DMA)
    for flags in (lib.POTRF_VALU, 0, lib.POTRF_SUBSTITUTION, lib.POTRF_NO_LOOKAHEAD,
                  lib.POTRF_CHAINS(3) | lib.POTRF_NB(1), lib.POTRF_CHAINS(4) | lib.POTRF_NB(1), lib.POTRF_REG_STAGING, lib.POTRF_REG_STAGING | lib.POTRF_NB(1)):
        L = rn.cholesky_device(dv.f64(A), flags, auto_substitution=False).cpu().numpy()
        ref = np.linalg.cholesky(A)
        assert np.max(np.abs(L - ref)) < 1e-10 * np.max(np.abs(ref)), (n, flags)
        assert np.all(np.triu(L, 1) == 0)


@pytest.mark.parametrize("n,batch,flags", [(1025, 2, 0), (1338, 3, 0), (2500, 2, 0), (2501, 1, 0), (700, 3, "NB1"), (1338, 3, "NB1"), (2500, 2, "NB1C3"),
                                           (3000, 5, "C4"), (3000, 2, "LA"), (3000, 2, "LAEPI1"), (2500, 2, "NB1LA"), (2501, 3, "NB1C3LA"), (4200, 3, "NB2LA1"), (2500, 2, "D64"), (2501, 2, "NB1D64LA"), (1200, 3, "LA"), (1290, 2, 0),
                                           # left-looking panel order (round 6): pure and with the run-ahead split, 2 .. 11 panels, one / two / three chains, odd order
                                           (3000, 2, "LEFT"), (3000, 2, "LEFTS"), (4200, 3, "LEFTS"), (2500, 2, "NB1LEFT"), (2900, 3, "NB1LEFTSC3"), (2501, 2, "NB1LEFTS"),
                                           (1200, 3, "LEFTS"), (4300, 1, "NB2LEFTS1"),
                                           # left-looking with the diagonal phases run ahead on the high-priority side stream (U_top / U_rest)
                                           (3000, 2, "LEFTLA"), (4200, 3, "LEFTLA"), (2900, 3, "NB1LEFTLAC3"), (2501, 2, "NB1LEFTLA"), (1200, 3, "LEFTLA"), (4300, 1, "NB2LEFTLA1"),
                                           # the substitution as ONE launch per panel (a workgroup per 128-row tile walks the panel's blocks, reading back its own
                                           # stores through L2): left- and right-looking, narrow first blocks (n mod 128 = 8, 56, 104), an odd order (falls back)
                                           (3000, 2, "LEFTROWS"), (4200, 9, "LEFTROWS"), (5000, 3, "LEFTROWS"), (2900, 3, "NB1LEFTROWSC3"), (2501, 2, "NB1LEFTROWS"),
                                           (4200, 3, "LAROWS"), (4328, 2, "NB2LEFTROWS"), (1290, 2, "LEFTROWS")])
def test_potrf_workspace_scheme_vs_numpy(gpu, n, batch, flags):
    """pta_potrf_batched_ws: panels factored on their diagonal block, explicit inverse W = L11^-1 in a caller-owned workspace (handed
    over full of NaN), rows below solved as X = B W^T right to left - against LAPACK, with leading dimension / stride slack, odd
    orders (non-vector operand path), narrow panels (256 columns: many steps, first panel 256 + n % 128 wide) and 3 / 4 chains."""
    dv, lib, torch = gpu["dv"], gpu["lib"], gpu["torch"]
    fl = {0: 0, "NB1": lib.POTRF_NB(1), "NB1C3": lib.POTRF_NB(1) | lib.POTRF_CHAINS(3), "C4": lib.POTRF_CHAINS(4),
          # look-ahead of the next panel's diagonal phase on a side stream (used while >= 1536 rows remain below it)
          "LA": lib.POTRF_DIAG_AHEAD, "LAEPI1": lib.POTRF_DIAG_AHEAD | lib.POTRF_EPI1, "NB1LA": lib.POTRF_NB(1) | lib.POTRF_DIAG_AHEAD, "NB1C3LA": lib.POTRF_NB(1) | lib.POTRF_CHAINS(3) | lib.POTRF_DIAG_AHEAD,
          "NB2LA1": lib.POTRF_NB(2) | lib.POTRF_DIAG_AHEAD | lib.POTRF_CHAINS(1),
          # A/B path: 64-column recursion on the diagonal block + inversion pass
          "D64": lib.POTRF_DIAG64, "NB1D64LA": lib.POTRF_NB(1) | lib.POTRF_DIAG64 | lib.POTRF_DIAG_AHEAD,
          "LEFT": lib.POTRF_LEFT, "LEFTS": lib.POTRF_LEFT | lib.POTRF_LEFT_SPLIT, "NB1LEFT": lib.POTRF_NB(1) | lib.POTRF_LEFT,
          "NB1LEFTS": lib.POTRF_NB(1) | lib.POTRF_LEFT | lib.POTRF_LEFT_SPLIT,
          "NB1LEFTSC3": lib.POTRF_NB(1) | lib.POTRF_LEFT | lib.POTRF_LEFT_SPLIT | lib.POTRF_CHAINS(3),
          "NB2LEFTS1": lib.POTRF_NB(2) | lib.POTRF_LEFT | lib.POTRF_LEFT_SPLIT | lib.POTRF_CHAINS(1),
          "LEFTLA": lib.POTRF_LEFT | lib.POTRF_DIAG_AHEAD, "NB1LEFTLA": lib.POTRF_NB(1) | lib.POTRF_LEFT | lib.POTRF_DIAG_AHEAD,
          "NB1LEFTLAC3": lib.POTRF_NB(1) | lib.POTRF_LEFT | lib.POTRF_DIAG_AHEAD | lib.POTRF_CHAINS(3),
          "NB2LEFTLA1": lib.POTRF_NB(2) | lib.POTRF_LEFT | lib.POTRF_DIAG_AHEAD | lib.POTRF_CHAINS(1),
          "LEFTROWS": lib.POTRF_LEFT | lib.POTRF_SOLVE_ROWS, "NB1LEFTROWS": lib.POTRF_NB(1) | lib.POTRF_LEFT | lib.POTRF_SOLVE_ROWS,
          "NB1LEFTROWSC3": lib.POTRF_NB(1) | lib.POTRF_LEFT | lib.POTRF_SOLVE_ROWS | lib.POTRF_CHAINS(3),
          "NB2LEFTROWS": lib.POTRF_NB(2) | lib.POTRF_LEFT | lib.POTRF_SOLVE_ROWS, "LAROWS": lib.POTRF_DIAG_AHEAD | lib.POTRF_SOLVE_ROWS}[flags]
    rng = np.random.default_rng(n + batch)
    X = rng.standard_normal((batch, n, n + 5))
    A = X @ X.transpose(0, 2, 1) + 0.1 * np.eye(n)
    ld = n + (n & 1) + 2
    buf = np.full((batch, n + 1, ld), np.nan)
    buf[:, :n, :n] = np.tril(A) + np.triu(np.full((n, n), np.nan), 1)     # only the lower triangle is read
    Ad = dv.f64(buf)
    info = dv.zeros((batch,), dtype=torch.int32)
    need = int(lib.lib.pta_potrf_workspace_doubles(n, batch, fl))
    assert need > 0
    work = dv.empty((need,))
    work.fill_(float("nan"))
    lib.call("pta_potrf_batched_ws", dv.ptr(Ad), n, ld, (n + 1) * ld, batch, dv.ptr(info), fl, dv.ptr(work), need, gpu["s"])
    assert int(info.abs().sum().item()) == 0
    L = np.tril(Ad.cpu().numpy()[:, :n, :n])
    ref = np.linalg.cholesky(A)
    assert np.all(np.isfinite(L))
    assert np.max(np.abs(L - ref)) < 1e-10 * np.max(np.abs(ref)), (n, flags)
    # too small a workspace / none: the workspace-free schedule, same factor
    Ad2 = dv.f64(buf)
    lib.call("pta_potrf_batched_ws", dv.ptr(Ad2), n, ld, (n + 1) * ld, batch, dv.ptr(info), fl, dv.ptr(work), need - 1, gpu["s"])
    L2 = np.tril(Ad2.cpu().numpy()[:, :n, :n])
    assert np.max(np.abs(L2 - ref)) < 1e-10 * np.max(np.abs(ref))
    assert int(lib.lib.pta_potrf_workspace_doubles(n, batch, fl | lib.POTRF_SUBSTITUTION)) == 0


@pytest.mark.parametrize("seed", range(10))
def test_potrf_uniform_schedules_random_shapes(gpu, seed):
    """round 6: random uniform batches (1 ... 11 matrices, orders 1025 ... 4500 - even and odd, any n mod 128 -, leading-dimension and stride
    slack, NaN above the diagonals, NaN workspace) under a random choice among the schedules of the workspace scheme - left-looking
    (the engine's default), left-looking with the run-ahead split / run-ahead diagonal phases / row-resident substitution, right-looking with
    look-ahead, 256 ... 2048-column panels, 1 ... 4 chains: every factor against LAPACK, and a second run bit-identical to the first."""
    dv, lib, torch = gpu["dv"], gpu["lib"], gpu["torch"]
    rng = np.random.default_rng(6000 + seed)
    n = int(rng.integers(1025, 4501))
    if seed % 2 == 0:
        n += n & 1                                                     # the DMA kernels' path (even order); odd orders take the scalar-operand kernels
    B = int(rng.integers(1, 12))
    L, S, LA, RW = lib.POTRF_LEFT, lib.POTRF_LEFT_SPLIT, lib.POTRF_DIAG_AHEAD, lib.POTRF_SOLVE_ROWS
    sched = [L, L | S, L | LA, L | RW, LA, L | RW | LA, LA | RW, L, L | S, L | RW][seed]
    fl = sched | lib.POTRF_NB(int(rng.choice([1, 2, 3, 4, 4, 6, 8]))) | lib.POTRF_CHAINS(int(rng.integers(1, 5)))
    X = rng.standard_normal((B, n, n + 3))
    A = X @ X.transpose(0, 2, 1) + 0.1 * np.eye(n)
    ld = n + (n & 1) + 2 * int(rng.integers(0, 3))
    rows = n + int(rng.integers(0, 2))
    buf = np.full((B, rows, ld), np.nan)
    buf[:, :n, :n] = np.tril(A) + np.triu(np.full((n, n), np.nan), 1)
    need = int(lib.lib.pta_potrf_workspace_doubles(n, B, fl))
    outs = []
    for rep in range(2):
        Ad = dv.f64(buf)
        info = dv.zeros((B,), dtype=torch.int32)
        work = dv.empty((max(need, 1),))
        work.fill_(float("nan"))
        lib.call("pta_potrf_batched_ws", dv.ptr(Ad), n, ld, rows * ld, B, dv.ptr(info), fl, dv.ptr(work) if need else None, need, gpu["s"])
        assert int(info.abs().sum().item()) == 0, (n, B, hex(fl))
        outs.append(np.tril(Ad.cpu().numpy()[:, :n, :n]))
    ref = np.linalg.cholesky(A)
    assert np.all(np.isfinite(outs[0]))
    assert np.max(np.abs(outs[0] - ref)) < 1e-10 * np.max(np.abs(ref)), (n, B, hex(fl))
    assert np.array_equal(outs[0], outs[1]), (n, B, hex(fl))


def _ragged_factor(gpu, mats, flags, nan_upper=True):
    """pta_potrf_ragged on a list of SPD matrices (any even orders): returns the lower factors and info."""
    dv, lib, torch = gpu["dv"], gpu["lib"], gpu["torch"]
    B = len(mats)
    n = np.array([m.shape[0] for m in mats], dtype=np.int32)
    ld = ((n.astype(np.int64) + 15) // 16 * 16 + 16 * (np.arange(B) % 2)).astype(np.int64)     # some slack in every other matrix
    off = np.concatenate([[0], np.cumsum(n.astype(np.int64) * ld + 32)])[:-1].astype(np.int64)    # and between matrices
    buf = np.full(int(off[-1] + n[-1] * ld[-1] + 32), np.nan)
    for b, m in enumerate(mats):
        v = buf[off[b]:off[b] + n[b] * ld[b]].reshape(n[b], ld[b])
        v[:, :n[b]] = np.tril(m) + (np.triu(np.full(m.shape, np.nan), 1) if nan_upper else np.triu(m, 1))
    Ad = dv.f64(buf)
    words = int(lib.lib.pta_potrf_ragged_plan_words(B))
    plan = np.zeros(words, dtype=np.int64)
    need = ctypes.c_int64(0)
    lib.call("pta_potrf_ragged_plan", dv.hptr(n), dv.hptr(off), dv.hptr(ld), B, flags, dv.hptr(plan), ctypes.byref(need))
    work = dv.empty((need.value,))
    work.fill_(float("nan"))
    info = dv.zeros((B,), dtype=torch.int32)
    plan_d = dv.i64(plan)
    lib.call("pta_potrf_ragged", dv.ptr(Ad), dv.hptr(plan), dv.ptr(plan_d), dv.ptr(info), dv.ptr(work), need.value, gpu["s"])
    out = Ad.cpu().numpy()
    Ls = [np.tril(out[off[b]:off[b] + n[b] * ld[b]].reshape(n[b], ld[b])[:, :n[b]]) for b in range(B)]
    return Ls, info.cpu().numpy()


def _spd(rng, n):
    X = rng.standard_normal((n, n + 5))
    return X @ X.T + 0.1 * np.eye(n)


@pytest.mark.parametrize("orders,flags", [
    ((2500, 130, 1024, 1026, 3000, 64, 2, 1152, 2048, 900), 0),          # two chains, look-ahead, 1024-column panels: entries at every kind of cut
    ((2500, 130, 1024, 1026, 3000, 64, 2, 1152, 2048, 900), "C1"),
    ((2500, 130, 1024, 1026, 3000, 64, 2, 1152, 2048, 900), "NOLA"),
    ((1500, 258, 700, 256, 254, 1280, 1282, 66, 1024, 512, 130), "NB1"),  # 256-column panels: many time steps, matrices entering at most of them
    ((1500, 258, 700, 256, 254, 1280, 1282, 66, 1024, 512, 130), "NB1C3"),
    ((2200, 2200, 2200), "NB1"),                                           # a uniform batch is the special case front = const
    ((4200, 3100, 600), "NB2"),
    ((4200, 3100, 600, 2050), "NB8"),                                       # 2048-column panels (the default for arrays of large matrices)
    ((5000,), 0),                                                           # a batch of one
    ((778, 90, 1026, 2602), 0),                                             # the TD test array's orders (cuts 10 / 2 / 90 columns wide)
    ((2500, 130, 1024, 1026, 3000, 64, 2, 1152, 2048, 900), "EPI1"),       # tile products with the C-tile prefetch epilogue (A/B form)
    # left-looking order (round 6): every time step's block column updated once with all virtual columns to its left
    ((1500, 258, 700, 256, 254, 1280, 1282, 66, 1024, 512, 130), "NB1LEFTC3"),
    ((4200, 3100, 600, 2050), "LEFT"),
    ((4200, 3100, 600, 2050, 4200, 3098), "NB2LEFT"),
    ((2200, 2200, 2200), "NB1LEFT"),
    ((778, 90, 1026, 2602), "LEFT"),
    ((5000,), "LEFT"),
])
def test_potrf_ragged_vs_numpy(gpu, orders, flags):
    """pta_potrf_ragged: matrices of different orders as ONE end-aligned schedule (red_noise.py:286-298 loops pulsars; a real array has
    as many TOA counts as pulsars) against LAPACK - NaN above every diagonal, NaN-filled workspace, leading-dimension and inter-matrix
    slack; orders that enter a time step at a block boundary, inside a block, one column before / after a panel boundary, and
    matrices smaller than one block."""
    lib = gpu["lib"]
    fl = {0: 0, "C1": lib.POTRF_CHAINS(1), "NOLA": lib.POTRF_NO_LOOKAHEAD, "NB1": lib.POTRF_NB(1), "NB1C3": lib.POTRF_NB(1) | lib.POTRF_CHAINS(3),
          "NB2": lib.POTRF_NB(2), "NB8": lib.POTRF_NB(8), "EPI1": lib.POTRF_EPI1, "LEFT": lib.POTRF_LEFT, "NB1LEFT": lib.POTRF_NB(1) | lib.POTRF_LEFT,
          "NB2LEFT": lib.POTRF_NB(2) | lib.POTRF_LEFT, "NB1LEFTC3": lib.POTRF_NB(1) | lib.POTRF_LEFT | lib.POTRF_CHAINS(3)}[flags]
    rng = np.random.default_rng(sum(orders))
    mats = [_spd(rng, n) for n in orders]
    Ls, info = _ragged_factor(gpu, mats, fl)
    assert not info.any(), info
    for b, (L, m) in enumerate(zip(Ls, mats)):
        ref = np.linalg.cholesky(m)
        assert np.all(np.isfinite(L)), (b, orders[b])
        assert np.max(np.abs(L - ref)) < 1e-10 * np.max(np.abs(ref)), (b, orders[b], flags)


@pytest.mark.parametrize("seed", range(8))
def test_potrf_ragged_random_order_sets(gpu, seed):
    """random batches (1 ... 9 matrices, even orders 2 ... 3000, any mix of equal and different orders) under a random choice of
    panel width / chains / look-ahead: every factor against LAPACK."""
    lib = gpu["lib"]
    rng = np.random.default_rng(1000 + seed)
    B = int(rng.integers(1, 10))
    orders = [2 * int(x) for x in rng.integers(1, 1501, B)]
    if seed % 3 == 0 and B > 2:
        orders[1] = orders[0]                      # equal orders share a time step from their first panel on
    flags = [0, lib.POTRF_NB(1), lib.POTRF_NB(2) | lib.POTRF_CHAINS(3), lib.POTRF_NO_LOOKAHEAD, lib.POTRF_NB(8), lib.POTRF_NB(3) | lib.POTRF_CHAINS(4),
             lib.POTRF_NB(1) | lib.POTRF_CHAINS(1), lib.POTRF_EPI1 | lib.POTRF_NB(2)][seed % 8]
    mats = [_spd(rng, n) for n in orders]
    Ls, info = _ragged_factor(gpu, mats, flags)
    assert not info.any(), (orders, info)
    for b, (L, m) in enumerate(zip(Ls, mats)):
        ref = np.linalg.cholesky(m)
        assert np.all(np.isfinite(L)), (b, orders)
        assert np.max(np.abs(L - ref)) < 1e-10 * np.max(np.abs(ref)), (b, orders, flags)


def test_potrf_ragged_is_deterministic_and_independent_of_the_batch(gpu):
    """two chains, look-ahead on side streams, matrices entering at different steps - and still: the same input gives the same bits
    run after run, and a matrix's factor does not depend on which other matrices share its batch (no atomics, no order-dependent sums)."""
    lib = gpu["lib"]
    rng = np.random.default_rng(77)
    orders = (2300, 600, 1400, 2300, 130)
    mats = [_spd(rng, n) for n in orders]
    a, info = _ragged_factor(gpu, mats, 0)
    b, _ = _ragged_factor(gpu, mats, 0)
    assert not info.any()
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    alone, _ = _ragged_factor(gpu, [mats[2]], 0)
    assert np.array_equal(alone[0], a[2])
    other, _ = _ragged_factor(gpu, [mats[4], mats[2], mats[0]], lib.POTRF_CHAINS(1))
    assert np.array_equal(other[1], a[2]) and np.array_equal(other[0], a[4])


def test_potrf_ragged_reports_the_first_bad_pivot_of_the_right_matrix(gpu):
    """info[b] in the caller's order and in the matrix's own numbering (LAPACK convention), whatever chain / time step the pivot falls in."""
    rng = np.random.default_rng(5)
    orders = (1400, 300, 2300, 700)
    mats = [_spd(rng, n) for n in orders]
    mats[2][1500:, 1500:] -= 2.0 * np.diag(np.diag(mats[2][1500:, 1500:]))   # Schur complement negative from row 1500 on
    Ls, info = _ragged_factor(gpu, mats, gpu["lib"].POTRF_NB(1), nan_upper=False)
    assert info[0] == 0 and info[1] == 0 and info[3] == 0
    # the first non-positive pivot by LAPACK-style elimination on the CPU
    try:
        np.linalg.cholesky(mats[2][:1500, :1500])
    except np.linalg.LinAlgError:  # pragma: no cover
        pytest.fail("the leading block was meant to be positive definite")
    lo, hi = 1500, 2300
    while lo < hi:  # smallest k whose leading minor of order k fails
        mid = (lo + hi) // 2
        try:
            np.linalg.cholesky(mats[2][:mid, :mid]); lo = mid + 1
        except np.linalg.LinAlgError:
            hi = mid
    assert info[2] == lo, (info, lo)
    for b in (0, 1, 3):
        ref = np.linalg.cholesky(mats[b])
        assert np.max(np.abs(Ls[b] - ref)) < 1e-10 * np.max(np.abs(ref)), b


def test_potrf_ragged_plan_rejects_odd_orders(gpu):
    lib, dv = gpu["lib"], gpu["dv"]
    n, off, ld = np.array([101], dtype=np.int32), np.array([0], dtype=np.int64), np.array([112], dtype=np.int64)
    plan = np.zeros(int(lib.lib.pta_potrf_ragged_plan_words(1)), dtype=np.int64)
    need = ctypes.c_int64(0)
    with pytest.raises(lib.PtaError):
        lib.call("pta_potrf_ragged_plan", dv.hptr(n), dv.hptr(off), dv.hptr(ld), 1, 0, dv.hptr(plan), ctypes.byref(need))


def test_potrf_not_positive_definite_raises(gpu):
    from pta_replicator_amd import red_noise as rn
    dv = gpu["dv"]
    A = np.eye(70); A[66, 66] = -1.0
    with pytest.raises(np.linalg.LinAlgError):
        rn.cholesky_device(dv.f64(A))


def test_potrf_headline_orf(gpu):
    """68-pulsar Hellings-Downs ORF (config 3 geometry): device Cholesky vs LAPACK."""
    from pta_replicator_amd import red_noise as rn, spharmORFbasis as anis
    rng = np.random.default_rng(68)
    locs = np.stack([rng.uniform(0, 24, 68) * np.pi / 12, np.pi / 2 - np.arcsin(rng.uniform(-1, 1, 68))], axis=1)
    orf = anis.orf_from_locations(locs)
    ref = np.linalg.cholesky(po.hd_orf_closed_form(locs))
    M = rn.cholesky_device(orf).cpu().numpy()
    assert np.max(np.abs(M - ref)) < 1e-13


def test_gwb_idft_mfma_vs_fft(gpu):
    """pruned inverse DFT as a GEMM == numpy's Hermitian-packed ifft, cropped (red_noise.py:275-285)."""
    dv, lib = gpu["dv"], gpu["lib"]
    rng = np.random.default_rng(3)
    for Nf, npts in ((3000, 600), (3001, 600), (601, 200)):
        M = 5
        w = rng.standard_normal((M, Nf, 2))
        C = rng.uniform(0.5, 2.0, Nf) * 1e-14
        dt = 1234.5
        T = dv.empty((2 * (Nf - 2), 608))
        sq_d, w_d = dv.f64(C ** 0.5), dv.f64(w)
        lib.call("pta_gwb_twiddle", dv.ptr(sq_d), Nf, npts, 10, 1.0 / dt, dv.ptr(T), 608, gpu["s"])
        Res_f = (w[..., 0] + 1j * w[..., 1]) * C ** 0.5
        Res_f[:, 0] = 0; Res_f[:, -1] = 0
        ref = po.gwb_time_series(Res_f, dt)[:, 10:npts + 10]
        for algo in (0, 1):
            G0 = dv.zeros((M, npts))
            lib.call("pta_gwb_idft", dv.ptr(w_d), 2 * Nf, M, Nf, dv.ptr(T), 608, npts, dv.ptr(G0), npts, algo, gpu["s"])
            assert np.max(np.abs(G0.cpu().numpy() - ref)) < 1e-12 * np.max(np.abs(ref)), (Nf, algo)


@pytest.mark.parametrize("variant", [0, 1, 2])
@pytest.mark.parametrize("Nf,npts", [(3000, 600), (3001, 600), (601, 200), (500, 37), (900, 601)])
def test_gwb_idft_rng_equals_replay_of_its_draws(gpu, variant, Nf, npts):
    """symmetric on-chip-RNG kernel == plain DFT-GEMM over the dumped draws (even/odd windows, both variants)."""
    dv, lib = gpu["dv"], gpu["lib"]
    seed, r0, R, P = 77, 1000, 3, 5
    C = np.linspace(2.0, 0.5, Nf) * 1e-14
    ldt = (npts + 15) // 16 * 16
    T = dv.empty((2 * (Nf - 2), ldt))
    sq_d = dv.f64(C ** 0.5)
    lib.call("pta_gwb_twiddle", dv.ptr(sq_d), Nf, npts, 10, 1.0 / 777.0, dv.ptr(T), ldt, gpu["s"])
    nrot = ctypes.c_int64(0)
    nsym = lib.lib.pta_gwb_twiddle_sym_size(Nf, npts, variant, ctypes.byref(nrot))
    Tsym, rot = dv.empty((nsym,)), dv.empty((nrot.value,))
    lib.call("pta_gwb_twiddle_sym", dv.ptr(sq_d), Nf, npts, 10, 1.0 / 777.0, dv.ptr(Tsym), dv.ptr(rot), variant, gpu["s"])
    G_rng = dv.zeros((R * P, npts))
    lib.call("pta_gwb_idft_rng", seed, r0, R, P, Nf, dv.ptr(Tsym), dv.ptr(rot), npts, dv.ptr(G_rng), npts, variant, 0, gpu["s"])
    w = dv.empty((R * P, 2 * Nf))
    for r in range(R):
        for a in range(P):
            lib.call("pta_rng_fill_normal", seed, r0 + r, 1, philox_ref.stream_id(1, a), Nf, 1,
                     ctypes.c_void_p(w.data_ptr() + 16 * Nf * (r * P + a)), None, 2 * Nf, 0, gpu["s"])
    G_rep = dv.zeros((R * P, npts))
    lib.call("pta_gwb_idft", dv.ptr(w), 2 * Nf, R * P, Nf, dv.ptr(T), ldt, npts, dv.ptr(G_rep), npts, 1, gpu["s"])
    a, b = G_rng.cpu().numpy(), G_rep.cpu().numpy()
    assert np.max(np.abs(a - b)) < 1e-12 * np.max(np.abs(b))
    # and the host twin of the draws
    z0, z1 = philox_ref.normal_pairs(seed, r0 + 2, philox_ref.stream_id(1, 4), Nf)
    wr = w.cpu().numpy()[2 * P + 4]
    assert np.max(np.abs(wr[0::2] - z0)) < 1e-13 and np.max(np.abs(wr[1::2] - z1)) < 1e-13


@pytest.mark.parametrize("variant", [0, 1, 11, 13, 17, 18])
@pytest.mark.parametrize("Nf,npts", [(3000, 600), (3001, 600), (601, 200), (500, 37), (2400, 601), (3400, 600)])
def test_gwb_chirp_z_fft_vs_numpy_and_vs_dft_gemm(gpu, Nf, npts, variant):
    """chirp-z path: (a) replay form vs numpy's Hermitian-packed ifft; (b) on-chip-RNG form vs the DFT-GEMM over the
    dumped draws."""
    dv, lib = gpu["dv"], gpu["lib"]
    assert lib.lib.pta_gwb_czt_fits(Nf, npts, 10) == 1 and lib.lib.pta_gwb_czt_fits(5000, 1000, 10) == 0
    rng = np.random.default_rng(Nf)
    seed, r0, R, P = 99, 12345, 2, 3
    M = R * P
    C = rng.uniform(0.5, 2.0, Nf) * 1e-14
    dt = 977.0
    sq_d = dv.f64(C ** 0.5)
    tabs = [dv.empty((8192,)), dv.empty((8192,)), dv.empty((8192,)), dv.empty((2 * npts,))]
    lib.call("pta_gwb_czt_setup", dv.ptr(sq_d), Nf, npts, 10, 1.0 / dt, *[dv.ptr(x) for x in tabs], gpu["s"])
    w = rng.standard_normal((M, Nf, 2))
    w_d = dv.f64(w)
    G = dv.zeros((M, npts))
    lib.call("pta_gwb_czt", 0, 0, dv.ptr(w_d), 2 * Nf, R, P, Nf, npts, 10, *[dv.ptr(x) for x in tabs], dv.ptr(G), npts, variant, 0, gpu["s"])
    Res_f = (w[..., 0] + 1j * w[..., 1]) * C ** 0.5
    Res_f[:, 0] = 0; Res_f[:, -1] = 0
    ref = po.gwb_time_series(Res_f, dt)[:, 10:npts + 10]
    assert np.max(np.abs(G.cpu().numpy() - ref)) < 1e-12 * np.max(np.abs(ref))
    # throughput form against the plain DFT-GEMM on its own draws
    G_rng = dv.zeros((M, npts))
    lib.call("pta_gwb_czt", seed, r0, None, 0, R, P, Nf, npts, 10, *[dv.ptr(x) for x in tabs], dv.ptr(G_rng), npts, variant, 0, gpu["s"])
    wd = dv.empty((M, 2 * Nf))
    for r in range(R):
        for a in range(P):
            lib.call("pta_rng_fill_normal", seed, r0 + r, 1, philox_ref.stream_id(1, a), Nf, 1,
                     ctypes.c_void_p(wd.data_ptr() + 16 * Nf * (r * P + a)), None, 2 * Nf, 0, gpu["s"])
    ldt = (npts + 15) // 16 * 16
    T = dv.empty((2 * (Nf - 2), ldt))
    lib.call("pta_gwb_twiddle", dv.ptr(sq_d), Nf, npts, 10, 1.0 / dt, dv.ptr(T), ldt, gpu["s"])
    G_rep = dv.zeros((M, npts))
    lib.call("pta_gwb_idft", dv.ptr(wd), 2 * Nf, M, Nf, dv.ptr(T), ldt, npts, dv.ptr(G_rep), npts, 1, gpu["s"])
    a_, b_ = G_rng.cpu().numpy(), G_rep.cpu().numpy()
    assert np.max(np.abs(a_ - b_)) < 1e-12 * np.max(np.abs(b_))


def test_td_mode_against_oracle(gpu):
    """dense path: covariance assembly -> blocked Cholesky -> L z, vs numpy on the same recipe and draws."""
    from pta_replicator_amd import red_noise as rn
    dv, lib = gpu["dv"], gpu["lib"]
    rng = np.random.default_rng(11)
    N, nm, R = 333, 30, 6
    t = np.sort(rng.uniform(53000, 58000, N)) * 86400.0
    epoch_of, ne, first, _ = po.quantize(t / 86400.0, dt=20.0)
    sig2 = rng.uniform(1e-14, 4e-14, N)
    ec = rng.uniform(1e-7, 3e-7, ne)
    Cref = po.td_covariance(t, -13.5, 3.3, nm, sig2, epoch_of, ec)
    Tspan = t.max() - t.min()
    f = np.arange(1, nm + 1) / Tspan
    Ft = dv.empty((2 * nm, N))
    # keep every device operand alive in a named tensor: a temporary freed inside the argument list hands its
    # block straight back to torch's caching allocator, and the next upload of the same call would overwrite it
    t_d, f_d = dv.f64(t), dv.f64(f)
    lib.call("pta_rn_basis", dv.ptr(t_d), N, 0.0, dv.ptr(f_d), None, nm, 0, dv.ptr(Ft), N, gpu["s"])
    phi = po.red_noise_prior(np.repeat(f, 2), -13.5, 3.3, Tspan)
    Cd = dv.zeros((N, N))
    phi_d, sig_d, ep_d, ec2_d = dv.f64(phi), dv.f64(sig2), dv.i32(epoch_of), dv.f64((ec ** 2)[epoch_of])
    lib.call("pta_td_cov_assemble", dv.ptr(Ft), N, N, 2 * nm, dv.ptr(phi_d), dv.ptr(sig_d), dv.ptr(ep_d), dv.ptr(ec2_d), dv.ptr(Cd), N,
             gpu["s"])
    lo = np.tril_indices(N)
    assert np.max(np.abs(Cd.cpu().numpy()[lo] - Cref[lo])) < 1e-10 * np.max(np.abs(Cref))
    # this matrix is deliberately harsh (0.1-0.2 us white noise under a 1e-13.5 red process: condition number ~1e7); the forward
    # error of ANY backward-stable Cholesky is ~cond * eps, so the factor is held to that bound, its backward error to 1e-14, and
    # the forward-substitution schedule (LAPACK-like, no inverted diagonal blocks) to 1e-10.  Realistic conditioning at the
    # headline size N = 5000 is pinned at 1e-10 in tests/test_gpu_td.py::test_td_headline_size_vs_numpy.
    cond = np.linalg.cond(Cref)
    Csym = Cd.clone()
    L = rn.cholesky_device(Cd)
    Lref = np.linalg.cholesky(Cref)
    Lh = L.cpu().numpy()
    assert np.max(np.abs(Lh - Lref)) < max(1e-10, 4 * cond * 2.2e-16) * np.max(np.abs(Lref)), cond
    assert np.max(np.abs(Lh @ Lh.T - Cref)) < 1e-14 * np.max(np.abs(Cref))
    Ls = rn.cholesky_device(Csym, lib.POTRF_SUBSTITUTION).cpu().numpy()
    assert np.max(np.abs(Ls - Lref)) < max(1e-10, 0.5 * cond * 2.2e-16) * np.max(np.abs(Lref)), cond
    z = rng.standard_normal((R, N))
    out = dv.zeros((R, N))
    z_d = dv.f64(z)
    lib.call("pta_td_trmm", dv.ptr(L), N, N, dv.ptr(z_d), N, R, dv.ptr(out), N, 0, 1, gpu["s"])
    ref = po.td_draw(Cref, z.T).T
    assert np.max(np.abs(out.cpu().numpy() - ref)) < max(1e-10, 4 * cond * 2.2e-16) * np.sqrt(np.mean(ref ** 2))
    assert np.max(np.abs(out.cpu().numpy() - z @ Lh.T)) < 1e-13 * np.sqrt(np.mean(ref ** 2))     # the product itself, same factor


@pytest.mark.parametrize("variant", [0, 1, 2])
@pytest.mark.parametrize("P,npts,R,ldg", [(68, 600, 5, 600), (3, 50, 1, 50), (16, 64, 9, 64), (17, 65, 4, 72), (80, 601, 3, 608), (81, 100, 2, 100),
                                          (200, 120, 2, 120), (1, 7, 6, 7)])
def test_gwb_mix_against_numpy(gpu, P, npts, R, ldg, variant):
    """G[r] = Mchol @ G0[r] (red_noise.py:268 behind the transform): the LDS-resident specialisation (P <= 80) and the generic
    batched GEMM, ragged sizes and a padded leading dimension."""
    dv, lib = gpu["dv"], gpu["lib"]
    rng = np.random.default_rng(P * 1000 + npts)
    M = np.tril(rng.standard_normal((P, P)))
    G0 = rng.standard_normal((R, P, ldg))
    M_d, G0_d = dv.f64(M), dv.f64(G0)
    G_d = dv.f64(np.full((R, P, ldg), 7.0))
    lib.call("pta_gwb_mix", dv.ptr(M_d), P, dv.ptr(G0_d), R, npts, ldg, dv.ptr(G_d), variant, gpu["s"])
    G = G_d.cpu().numpy()
    ref = np.einsum("ab,rbj->raj", M, G0[:, :, :npts])
    assert np.max(np.abs(G[:, :, :npts] - ref)) < 1e-13 * max(1.0, np.max(np.abs(ref)))
    assert np.all(G[:, :, npts:] == 7.0)            # padding columns untouched


def test_clock_probe_measures_a_plausible_engine_clock(gpu):
    """pta_clock_probe: (s_memrealtime, s_memtime) pairs from one wave on a side stream; the slope is the engine clock."""
    dv, lib, torch = gpu["dv"], gpu["lib"], gpu["torch"]
    ns = 64
    buf = torch.zeros((ns, 2), dtype=torch.int64, device="cuda")
    side = torch.cuda.Stream()
    lib.call("pta_clock_probe", buf.data_ptr(), ns, 20000, 500, side.cuda_stream)       # 20 ms, one sample per 0.5 ms
    side.synchronize()
    s = buf.cpu().numpy()
    s = s[s[:, 0] > 0]
    assert 30 <= len(s) <= ns
    assert np.all(np.diff(s[:, 0]) > 0) and np.all(np.diff(s[:, 1]) > 0)
    ghz = np.diff(s[:, 1]) / (np.diff(s[:, 0]) * 10.0)                                # cycles per 10 ns tick of the 100 MHz reference
    assert 0.05 < np.median(ghz) < 3.5, np.median(ghz)
    with pytest.raises(lib.PtaError):
        lib.call("pta_clock_probe", None, ns, 1000, 100, None)
