"""TD ("time-domain") mode on the GPU: dense covariance -> blocked Cholesky -> L . z with on-chip deviates (+ GWB), the path
BASELINE.json's north_star names.  Oracle: oracle/pta_oracle.py td_* (NumPy / LAPACK) on the deviates the kernels drew.

Tolerances.  1e-10 relative RMS wherever the comparison is between two evaluations of the SAME linear map (device product vs
NumPy product with the same factor) and for the per-pulsar factors (condition number <~ 1e8).  The GWB grid covariance has a
condition number of ~3e14: two correct float64 Cholesky factorisations of it differ by ~cond * eps, LAPACK's included (measured
below against an 80-bit factorisation), so the device factor is pinned by its BACKWARD error (< 1e-12 and within 10x of
LAPACK's on the same matrix) and by a forward error no worse than 20x LAPACK's.
"""
import ctypes

import numpy as np
import pytest

from helpers import relrms
from oracle import pta_oracle as po
from oracle import philox_ref

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a MI355X"
    from pta_replicator_amd import _lib, device as dv
    return {"torch": torch, "lib": _lib, "dv": dv, "s": dv.stream_ptr()}


def _plan(gpu, Ls, lds, rows_per_real, kind, keep, old_strips=False):
    """pta_td_plan over a list of host factors (row-major, leading dimension lds[b]); the upper triangles hold NaN."""
    lib, dv = gpu["lib"], gpu["dv"]
    from pta_replicator_amd.engine_td import _strips
    ns = [L.shape[0] for L in Ls]
    pos = np.concatenate([[0], np.cumsum([n * l for n, l in zip(ns, lds)])]).astype(np.int64)
    buf = np.full(int(pos[-1]), np.nan)
    for b, L in enumerate(Ls):
        v = buf[pos[b]:pos[b] + ns[b] * lds[b]].reshape(ns[b], lds[b])
        il = np.tril_indices(ns[b])
        v[il] = L[il]
    off = np.concatenate([[0], np.cumsum(ns)]).astype(np.int32)
    blk, n0, rows = _strips(ns)
    if old_strips:   # the layout of rounds 2-5: strips at multiples of PTA_TD_STRIP, item_rows = NULL (the partial strip last)
        items = sorted([(min(int(n), k + 256), b, k) for b, n in enumerate(ns) for k in range(0, int(n), 256)], key=lambda x: -x[0])
        blk, n0, rows = np.array([x[1] for x in items], dtype=np.int32), np.array([x[2] for x in items], dtype=np.int32), None
    dev = [dv.f64(buf), dv.i64(pos[:-1]), dv.i32(lds), dv.i32(ns), dv.i32(off[:-1] if rows_per_real == 1 else [0]), dv.i32(blk), dv.i32(n0)]
    if rows is not None:
        dev.append(dv.i32(rows))
    keep.extend(dev)
    tp = lib.TdPlan()
    tp.Lbase = dev[0].data_ptr()
    tp.blk_pos, tp.blk_ld, tp.blk_n, tp.blk_off, tp.item_blk, tp.item_n0 = [x.data_ptr() for x in dev[1:7]]
    tp.item_rows = dev[7].data_ptr() if rows is not None else None
    tp.n_blocks, tp.n_items, tp.rows_per_real, tp.stream_kind, tp.rng_fast = len(Ls), len(blk), rows_per_real, kind, 0
    return tp, off


def _normals(gpu, seed, r, kind, psr, n):
    lib, dv = gpu["lib"], gpu["dv"]
    npair = (n + 1) // 2
    buf = dv.empty((2 * npair,))
    lib.call("pta_rng_fill_normal", seed, r, 1, philox_ref.stream_id(kind, psr), npair, 1, dv.ptr(buf), None, 2 * npair, 0, gpu["s"])
    return buf.cpu().numpy()[:n]


@pytest.mark.parametrize("sizes,R", [((1, 17, 256, 257), 5), ((300, 333, 513), 70), ((1024,), 64), ((2,), 1), ((4200, 4100, 4300, 4250), 3)])
def test_td_trmm_rng_per_pulsar_blocks(gpu, sizes, R):
    """out[m, off_b + i] = sum_{j <= i} L_b[i, j] z(m, b, j) with z generated in registers: ragged factor orders (strip
    remainders, a 1 x 1 factor), odd orders with padded leading dimensions, realisation counts off the 64-row groups, NaN above
    the diagonals (never read).  The last case has 68 strips: from 64 on, whole strips are dealt to the XCDs (the launch order
    of the 68 x 5000 array); below, every XCD takes a share of every strip's row groups."""
    lib, dv = gpu["lib"], gpu["dv"]
    rng = np.random.default_rng(sum(sizes) + R)
    Ls = [np.tril(rng.standard_normal((n, n))) / np.sqrt(n) + 3 * np.eye(n) for n in sizes]
    lds = [(n + 1) // 2 * 2 + (2 if b % 2 else 0) for b, n in enumerate(sizes)]
    keep = []
    tp, off = _plan(gpu, Ls, lds, 1, 5, keep)
    ntot = int(off[-1])
    out = dv.f64(np.full((R, ntot + 3), 7.0))
    seed, r0 = 99, 1234567890123
    lib.call("pta_td_trmm_rng", ctypes.byref(tp), seed, r0, R, dv.ptr(out), ntot + 3, gpu["s"])
    got = out.cpu().numpy()
    assert np.all(got[:, ntot:] == 7.0)
    for b, L in enumerate(Ls):
        n = sizes[b]
        for m in range(R):
            z = _normals(gpu, seed, r0 + m, 5, b, n)
            ref = L @ z
            assert np.max(np.abs(got[m, off[b]:off[b] + n] - ref)) < 1e-12 * max(1.0, np.max(np.abs(ref))), (b, m)


@pytest.mark.parametrize("zmem", [False, True])
def test_td_trmm_partial_strip_first_is_bit_identical_to_partial_strip_last(gpu, zmem):
    """round 6: pta_td_plan.item_rows lets a factor's partial strip come FIRST (rows [0, f), K extent f) instead of last (K extent = the
    factor's order: 7 of 16 column tiles of every slab dead at 5000 TOAs).  Same accumulators, same K order per output element: the
    output must be BIT-identical to the rounds 2-5 layout (strips at multiples of 256, item_rows = NULL) - orders on both sides of every
    boundary (f = 16 k, 16 k + 1, 255, 256, 257, below one strip, 5000 and 600 themselves), register and memory deviates."""
    lib, dv, torch = gpu["lib"], gpu["dv"], gpu["torch"]
    sizes, R = (5000, 600, 257, 255, 256, 272, 273, 17, 1, 513, 1000), 70
    rng = np.random.default_rng(77)
    Ls = [np.tril(rng.standard_normal((n, n))) / np.sqrt(n) + 3 * np.eye(n) for n in sizes]
    lds = [(n + 1) // 2 * 2 + (2 if b % 2 else 0) for b, n in enumerate(sizes)]
    outs = []
    for old in (True, False):
        keep = []
        tp, off = _plan(gpu, Ls, lds, 1, 5, keep, old_strips=old)
        ntot = int(off[-1])
        seed, r0 = 99, 1234567890123
        if zmem:
            zoff = np.concatenate([[0], np.cumsum((np.array(sizes) + 1) // 2 * 2)]).astype(np.int32)
            z = dv.zeros((R, int(zoff[-1]) + 16))
            keep += [dv.i32(list(sizes)), dv.i32(zoff[:-1])]
            lib.call("pta_rng_fill_normal_blocks", seed, r0, R, 5, len(sizes), dv.ptr(keep[-2]), dv.ptr(keep[-1]), int(max(sizes)), dv.ptr(z), z.stride(0), 0, gpu["s"])
            tp.z, tp.ld_z, tp.blk_zoff = z.data_ptr(), z.stride(0), keep[-1].data_ptr()
        out = dv.f64(np.full((R, ntot + 3), 7.0))
        lib.call("pta_td_trmm_rng", ctypes.byref(tp), seed, r0, R, dv.ptr(out), ntot + 3, gpu["s"])
        outs.append(out.cpu().numpy())
    assert np.array_equal(outs[0], outs[1])
    assert np.all(outs[1][:, -3:] == 7.0) and np.all(np.isfinite(outs[1]))
    z0 = _normals(gpu, 99, 1234567890123, 5, 0, 5000)
    ref = Ls[0] @ z0
    assert np.max(np.abs(outs[1][0, :5000] - ref)) < 1e-12 * np.max(np.abs(ref))


def test_td_trmm_rng_shared_grid_factor(gpu):
    """rows_per_real = P: one factor shared by (realisation, pulsar) rows, stream (TDGW, pulsar) - the GWB grid form."""
    lib, dv = gpu["lib"], gpu["dv"]
    rng = np.random.default_rng(5)
    n, P, R = 600, 7, 11
    L = np.tril(rng.standard_normal((n, n))) / 10 + np.eye(n)
    keep = []
    tp, _ = _plan(gpu, [L], [n], P, 6, keep)
    out = dv.zeros((R * P, n))
    seed, r0 = 3, 40
    lib.call("pta_td_trmm_rng", ctypes.byref(tp), seed, r0, R * P, dv.ptr(out), n, gpu["s"])
    got = out.cpu().numpy().reshape(R, P, n)
    for r in (0, 5, 10):
        for a in (0, 3, 6):
            ref = L @ _normals(gpu, seed, r0 + r, 6, a, n)
            assert np.max(np.abs(got[r, a] - ref)) < 1e-12 * np.max(np.abs(ref))


def _small_array(with_gwb, with_cgw=False, seed=21):
    from pta_replicator_amd.engine import ReplicaEngine
    from pta_replicator_amd.simulate import ArrayTOAs, SimulatedPulsar, make_ideal
    rng = np.random.default_rng(seed)
    Ns = [389, 256, 301]
    psrs = []
    for a, N in enumerate(Ns):
        mjd = np.sort(rng.uniform(53000, 57000, N))
        mjd[5:8] = mjd[5] + np.array([0.0, 0.01, 0.03])           # a three-TOA epoch
        flags = [{"f": "A" if x < 0.5 else "B"} for x in rng.uniform(size=N)]
        p = SimulatedPulsar(toas=ArrayTOAs(mjd, rng.uniform(0.3, 0.8, N), flags=flags), name=f"J{a:02d}",
                            loc={"RAJ": float(rng.uniform(0, 24)), "DECJ": float(rng.uniform(-60, 60))})
        make_ideal(p)
        psrs.append(p)
    eng = ReplicaEngine(psrs, seed=4242)
    noise = dict(flags=[["A", "B"]] * 3, efac=[np.array([1.1, 0.9])] * 3, log10_equad=[np.array([-6.6, -6.3])] * 3,
                 log10_ecorr=[np.array([-6.4, -6.7])] * 3, rn_log10_A=[-13.6, None, -14.1], rn_gamma=[3.1, None, 4.4])
    eng.set_white_noise(efac=noise["efac"], log10_equad=noise["log10_equad"], flags=noise["flags"])
    eng.set_jitter(log10_ecorr=noise["log10_ecorr"], flags=noise["flags"], coarsegrain=0.1)
    eng.set_red_noise(noise["rn_log10_A"], noise["rn_gamma"], components=12)
    if with_gwb:
        eng.set_gwb(-14.3, 13. / 3.)
    if with_cgw:
        eng.add_cgw(gwtheta=1.1, gwphi=2.5, mc=1e9, dist=15.0, fgw=1e-8, phase0=0.5, psi=1.5, inc=0.7, pdist=1.0, pphase=None,
                    psrTerm=True, evolve=True, phase_approx=False, tref=53000 * 86400)
    return eng, psrs, noise


def _oracle_covariances(eng, psrs, noise, components=12, only=None):
    covs = []
    for a, psr in enumerate(psrs):
        if only is not None and a != only:
            covs.append(None)
            continue
        n = int(eng.counts[a])
        tf = np.array([f["f"] for f in psr.toas.table["flags"].data])
        sig = eng.sigma_s[a]
        efv = po.flag_vector(tf, noise["flags"][a], noise["efac"][a], n)
        eqv = po.flag_vector(tf, noise["flags"][a], 10 ** np.asarray(noise["log10_equad"][a]), n)
        sigma2 = (efv * sig) ** 2 + (efv * eqv) ** 2
        epoch_of, ne, first, _ = po.quantize(eng.mjd[a], dt=0.1)
        ecv = po.jitter_ecorr_vector(ne, first, noise["log10_ecorr"][a], toa_flags=tf, flags=noise["flags"][a])
        t = eng.tdb_s[a]
        if noise["rn_log10_A"][a] is None:
            C = np.diag(sigma2) + (epoch_of[:, None] == epoch_of[None, :]) * (ecv[epoch_of] ** 2)[:, None]
        else:
            C = po.td_covariance(t, noise["rn_log10_A"][a], noise["rn_gamma"][a], components, sigma2, epoch_of, ecv)
        covs.append(C)
    return covs


def test_td_engine_per_pulsar_part_vs_oracle():
    """prepare_td + generate_td without a GWB: covariance assembly, batched/ragged factorisation and the in-register-draw
    product of three ragged pulsars (one without red noise) against NumPy on the dumped deviates, 1e-10 relative RMS; the CGW
    term rides along; replay_td (explicit-operand kernels) gives the same numbers."""
    eng, psrs, noise = _small_array(False, with_cgw=True)
    eng.prepare_td()
    R = 5
    out = eng.generate_td(R, r0=7).cpu().numpy()
    covs = _oracle_covariances(eng, psrs, noise)
    det = eng.d_det.cpu().numpy()
    draws = [eng.dump_draws_td(7 + r) for r in range(R)]
    for a in range(3):
        sl = slice(eng.off[a], eng.off[a + 1])
        L = eng.td_factor(a).cpu().numpy()
        Lref = np.linalg.cholesky(covs[a])
        assert np.max(np.abs(L - Lref)) < 1e-10 * np.max(np.abs(Lref)), a
        for r in range(R):
            ref = po.td_draw(covs[a], draws[r]["td"][a]) + det[sl]
            assert relrms(out[r, sl], ref) < 1e-10, (a, r)
    rep = eng.replay_td(draws).cpu().numpy()
    assert np.max(np.abs(rep - out)) < 1e-12 * np.sqrt(np.mean(out ** 2))
    assert np.array_equal(eng.generate_td(2, r0=9).cpu().numpy(), out[2:4])      # offset / batch independent


def test_td_engine_gwb_part():
    """the GWB of TD mode: grid covariance T^T T against the Toeplitz formula of SURVEY.md App. A.1 (1e-12), its device
    factor by backward error (1e-12, within 10x of LAPACK's) and by forward error against an 80-bit factorisation (no worse than 20x LAPACK's), and
    the injected term against the oracle evaluated with the device's own factor (1e-10) and with NumPy's (conditioning-limited)."""
    eng, psrs, noise = _small_array(True)
    eng.prepare_td()
    eng0, _, _ = _small_array(False)
    eng0.prepare_td()
    R = 3
    out = eng.generate_td(R).cpu().numpy()
    base = eng0.generate_td(R).cpu().numpy()                  # same seed and TD streams: the difference is the GWB term
    npts = eng.plan.gw_npts
    grid = po.gwb_grid([float(m.min()) for m in eng.mjd], [float(m.max()) for m in eng.mjd])
    Cf = po.gwb_spectrum(grid["f"], grid["dur"], grid["howml"], -14.3, 13. / 3.)
    S = po.td_gwb_grid_covariance(grid, Cf)
    Sg = eng.Sg_td[:, :npts].cpu().numpy()
    il = np.tril_indices(npts)
    assert np.max(np.abs(Sg[il] - S[il])) < 1e-12 * np.max(np.abs(S))
    Lg = eng.gw_grid_factor().cpu().numpy()
    Sdev = np.tril(Sg) + np.tril(Sg, -1).T                       # what the device factored (its own Gram matrix, lower half)
    Sj = Sdev + eng.gw_td_jitter * np.mean(np.diag(Sdev)) * np.eye(npts)
    back_dev = np.max(np.abs(Lg @ Lg.T - Sj)) / np.max(np.abs(S))
    Lnp = np.linalg.cholesky(Sj)
    back_np = np.max(np.abs(Lnp @ Lnp.T - Sj)) / np.max(np.abs(S))
    assert back_dev < 1e-12 and back_dev < 10 * back_np + 1e-14, (back_dev, back_np)     # n eps = 1.3e-13 at n = 600
    Lx = po.cholesky_longdouble(Sj)
    err_dev = np.max(np.abs(Lg - Lx.astype(np.float64)))
    err_lapack = np.max(np.abs(np.linalg.cholesky(Sj) - Lx.astype(np.float64)))
    assert err_dev < 20 * err_lapack + 1e-16 * np.max(np.abs(Lg)), (err_dev, err_lapack)
    Mchol = np.linalg.cholesky(po.hd_orf_closed_form(po.psr_locs_equatorial([p.loc for p in psrs])))
    toa_s = [m * 86400 for m in eng.mjd]
    for r in range(R):
        d = eng.dump_draws_td(r)
        zero = [np.zeros((int(n), int(n))) + np.eye(int(n)) for n in eng.counts]
        gw_dev = po.td_realisation(toa_s, zero, [np.zeros(int(n)) for n in eng.counts], grid=grid, Lg=Lg, M=Mchol, z_gw=d["gwb"])
        gw_np = po.td_realisation(toa_s, zero, [np.zeros(int(n)) for n in eng.counts], grid=grid, Lg=np.linalg.cholesky(Sj), M=Mchol,
                                  z_gw=d["gwb"])
        for a in range(3):
            sl = slice(eng.off[a], eng.off[a + 1])
            got = out[r, sl] - base[r, sl]
            assert relrms(got, gw_dev[a]) < 1e-10, (r, a)
            assert relrms(got, gw_np[a]) < 1e-3, (r, a)          # two float64 factors of a cond ~ 3e14 matrix
    rep = eng.replay_td([eng.dump_draws_td(r) for r in range(R)]).cpu().numpy()
    assert np.max(np.abs(rep - out)) < 1e-12 * np.sqrt(np.mean(out ** 2))


def test_td_and_throughput_mode_have_the_same_ensemble_covariance():
    """TD mode, throughput mode and throughput mode with grid-drawn GWB draw from the same Gaussian: sample covariances of
    16384 realisations each (two pulsars, RN + EFAC/EQUAD + ECORR + GWB) agree element by element within sampling error with
    the analytic covariance C_a (+ ORF_ab A_a Sigma_g A_b^T for the GWB)."""
    import torch
    from pta_replicator_amd.engine import ReplicaEngine
    from pta_replicator_amd.simulate import ArrayTOAs, SimulatedPulsar, make_ideal
    rng = np.random.default_rng(77)
    Ns, R = [40, 33], 16384
    psrs = []
    for a, N in enumerate(Ns):
        mjd = np.sort(rng.uniform(53000, 56000, N))
        p = SimulatedPulsar(toas=ArrayTOAs(mjd, rng.uniform(0.3, 0.6, N)), name=f"J{a}", loc={"RAJ": 3.0 + 7 * a, "DECJ": 10.0 - 30 * a})
        make_ideal(p)
        psrs.append(p)
    eng = ReplicaEngine(psrs, seed=31)
    eng.set_white_noise(efac=1.1, log10_equad=-6.5)
    eng.set_jitter(log10_ecorr=-6.4, coarsegrain=0.1)
    eng.set_red_noise([-13.2, -13.5], [2.5, 3.5], components=5)
    eng.set_gwb(-13.3, 13. / 3.)
    eng.prepare_td()
    x_td = eng.generate_td(R)
    x_fd = eng.generate(R)
    eng.gwb_mode = "grid"                 # throughput mode with the GWB drawn on the grid (npts deviates per pulsar)
    x_gr = eng.generate(R)
    eng.gwb_mode = "fourier"
    assert torch.equal(eng.generate(3), x_fd[:3])          # the default path is untouched by the mode switch
    c_td = (x_td.T @ x_td / R).cpu().numpy()
    c_fd = (x_fd.T @ x_fd / R).cpu().numpy()
    c_gr = (x_gr.T @ x_gr / R).cpu().numpy()
    # analytic covariance
    n = eng.n_toa
    Cm = np.zeros((n, n))
    grid = po.gwb_grid([float(m.min()) for m in eng.mjd], [float(m.max()) for m in eng.mjd])
    Cf = po.gwb_spectrum(grid["f"], grid["dur"], grid["howml"], -13.3, 13. / 3.)
    S = po.td_gwb_grid_covariance(grid, Cf)
    orf = po.hd_orf_closed_form(po.psr_locs_equatorial([p.loc for p in psrs]))
    A = []
    for a in range(2):
        Aa = np.stack([po.lerp_sorted(grid["ut"], e, eng.mjd[a] * 86400) for e in np.eye(grid["npts"])], axis=1)   # [N_a, npts]
        A.append(Aa)
    for a in range(2):
        sa = slice(eng.off[a], eng.off[a + 1])
        epoch_of, ne, first, _ = po.quantize(eng.mjd[a], dt=0.1)
        sig2 = (1.1 * eng.sigma_s[a]) ** 2 + (1.1 * 10 ** -6.5) ** 2
        Cm[sa, sa] = po.td_covariance(eng.tdb_s[a], [-13.2, -13.5][a], [2.5, 3.5][a], 5, sig2, epoch_of, np.full(ne, 10 ** -6.4))
        for b in range(2):
            sb = slice(eng.off[b], eng.off[b + 1])
            Cm[sa, sb] += orf[a, b] * (A[a] @ S @ A[b].T)
    d = np.sqrt(np.diag(Cm))
    for emp in (c_td, c_fd, c_gr):
        assert np.max(np.abs(emp - Cm) / np.outer(d, d)) < 6.0 / np.sqrt(R)
    assert abs(np.trace(c_td) / np.trace(c_fd) - 1) < 0.03


@pytest.mark.parametrize("P,N,unsorted", [(5, 2600, False), (3, 1410, True), (2, 5000, False)])
def test_td_fused_assembly_factorisation_equals_two_step(P, N, unsorted):
    """round 6: pta_td_assemble_potrf - the covariance never written, every block column of the left-looking factorisation
    COMPUTED by its update (F phi F^T + diag + ECORR - L L^T) - against the two-step path (k_td_cov_walk, then pta_potrf_batched_ws in the
    same panel order) on the ng15 noise values: per-backend EFAC / EQUAD / ECORR, a pulsar without red noise, 2 / 3 / 5 pulsars (chains
    of unequal size: the assembly operands are offset per chain), 2 - 5 panels, TOAs in time order and shuffled (ECORR epochs scattered
    over the tiles).  Factors within 1e-12 of each other relative to max |L| (the K = 60 product is summed in another order) and within
    1e-10 of LAPACK on the oracle's covariance; L.z on the dumped deviates at 1e-10; the fused path is deterministic (bit-equal reruns)."""
    import torch
    from pta_replicator_amd.engine import ReplicaEngine
    from bench import configure_engine, headline_array
    psrs, noise = headline_array(P, N, seed=70 + P)
    if unsorted:
        from pta_replicator_amd.simulate import ArrayTOAs, SimulatedPulsar, make_ideal
        rng = np.random.default_rng(3)
        sh = []
        for p in psrs:
            perm = rng.permutation(N)
            q = SimulatedPulsar(toas=ArrayTOAs(np.asarray(p.toas.get_mjds().value)[perm], 0.5, flags=[p.toas.flags[i] for i in perm]), name=p.name, loc=dict(p.loc))
            make_ideal(q)
            sh.append(q)
        psrs = sh
    noise["rn_log10_A"][1], noise["rn_gamma"][1] = None, None        # one pulsar without red noise
    eng = configure_engine(ReplicaEngine(psrs, seed=5), noise)
    eng._gw = None
    eng.prepare()
    eng.td_fused = False
    eng.prepare_td()
    assert eng.td_cov_kernel_used == "walk"
    two = [eng.td_factor(a).cpu().numpy() for a in range(P)]
    eng.td_fused = True
    eng.d_Ltd.fill_(float("nan"))                                     # the fused path reads nothing of the buffer
    eng.prepare_td()
    assert eng.td_cov_kernel_used == "fused" and eng.td_potrf_mode_used == "uniform"
    fused = [eng.td_factor(a).cpu().numpy() for a in range(P)]
    first = eng.d_Ltd.clone()
    for a in range(P):
        assert np.all(np.isfinite(fused[a]))
        assert np.max(np.abs(fused[a] - two[a])) < 1e-12 * np.max(np.abs(two[a])), a
    covs = _oracle_covariances(eng, psrs, noise, components=30)
    for a in (0, 1):
        Lref = np.linalg.cholesky(covs[a])
        assert np.max(np.abs(fused[a] - Lref)) < 1e-10 * np.max(np.abs(Lref)), a
    out = eng.generate_td(2).cpu().numpy()
    for a in (0, 1):
        z = eng.dump_draws_td(1)["td"][a]
        assert relrms(out[1, eng.off[a]:eng.off[a + 1]], np.linalg.cholesky(covs[a]) @ z) < 1e-10
    eng.prepare_td()
    lower = torch.tril(torch.ones((eng.td_nst[0], eng.td_ld[0]), dtype=torch.bool, device="cuda"))
    v0, v1 = first.view(P, eng.td_nst[0], eng.td_ld[0]), eng.d_Ltd.view(P, eng.td_nst[0], eng.td_ld[0])
    assert bool(torch.equal(v0[:, lower], v1[:, lower]))


def test_td_headline_size_vs_numpy():
    """N = 5000 TOAs, two pulsars with the ng15 noise values of config 3: covariance, factor and L.z against NumPy / LAPACK at
    1e-10 (the size BASELINE.json's metric is quoted on; LAPACK's potrf of a 5000^2 takes about a second)."""
    from pta_replicator_amd.engine import ReplicaEngine
    from bench import configure_engine, headline_array
    psrs, noise = headline_array(2, 5000)
    eng = configure_engine(ReplicaEngine(psrs, seed=5), noise)
    eng._gw = None
    eng.prepare_td()
    R = 3
    out = eng.generate_td(R).cpu().numpy()
    covs = _oracle_covariances(eng, psrs, noise, components=30)
    for a in range(2):
        n, ld = int(eng.counts[a]), eng.td_ld[a]
        Lref = np.linalg.cholesky(covs[a])
        L = eng.td_factor(a).cpu().numpy()
        assert np.max(np.abs(L - Lref)) < 1e-10 * np.max(np.abs(Lref)), a
        sl = slice(eng.off[a], eng.off[a + 1])
        for r in range(R):
            z = eng.dump_draws_td(r)["td"][a]
            assert relrms(out[r, sl], Lref @ z) < 1e-10, (a, r)
    # the covariance itself (lower triangle) as the device assembled it: re-assemble into a scratch buffer
    import torch
    from pta_replicator_amd import _lib, device as dv
    a, n = 1, 5000
    Cd = dv.zeros((n, n))
    o = int(eng.off[a])
    phi = (eng.d_amp ** 2).contiguous()
    ec2 = (eng.d_ecorr_toa ** 2).contiguous()
    _lib.call("pta_td_cov_assemble", ctypes.c_void_p(eng.d_Ft.data_ptr() + 8 * o), eng.n_toa, n, eng.K, ctypes.c_void_p(phi.data_ptr() + 8 * a * eng.K),
              ctypes.c_void_p(eng._td_sigma2.data_ptr() + 8 * o), ctypes.c_void_p(eng.d_epoch_of.data_ptr() + 4 * o),
              ctypes.c_void_p(ec2.data_ptr() + 8 * o), dv.ptr(Cd), n, dv.stream_ptr())
    il = np.tril_indices(n)
    assert np.max(np.abs(Cd.cpu().numpy()[il] - covs[a][il])) < 1e-10 * np.max(np.abs(covs[a]))


def test_td_config5_size_one_pulsar_vs_lapack():
    """N = 10 000 TOAs (BASELINE.json config 5's per-pulsar size; 10000 mod 128 = 16, so the first panel is 1040 wide), three
    pulsars with ng15 noise values: the factor of one pulsar against LAPACK and L.z against NumPy at 1e-10."""
    from pta_replicator_amd.engine import ReplicaEngine
    from bench import configure_engine, headline_array
    psrs, noise = headline_array(3, 10000)
    eng = configure_engine(ReplicaEngine(psrs, seed=55), noise)
    eng._gw = None
    eng.prepare_td()
    out = eng.generate_td(2).cpu().numpy()
    a = 1
    cov = _oracle_covariances(eng, psrs, noise, components=30, only=a)[a]
    Lref = np.linalg.cholesky(cov)
    L = eng.td_factor(a).cpu().numpy()
    assert np.max(np.abs(L - Lref)) < 1e-10 * np.max(np.abs(Lref))
    sl = slice(eng.off[a], eng.off[a + 1])
    for r in range(2):
        assert relrms(out[r, sl], Lref @ eng.dump_draws_td(r)["td"][a]) < 1e-10, r


def test_td_engine_fast_rng_math_is_self_consistent():
    """engine.rng_fast = 1 reaches the TD kernels through their plan structs: generate_td still equals replay_td on its own
    dumped deviates, differs from the fp64-transform realisation at the 1e-6 level of the deviates only, and switching it off
    restores the default bit for bit."""
    eng, psrs, noise = _small_array(True)
    eng.prepare_td()
    ref = eng.generate_td(3, r0=5).cpu().numpy()
    eng.rng_fast = 1
    out = eng.generate_td(3, r0=5).cpu().numpy()
    rep = eng.replay_td([eng.dump_draws_td(5 + r) for r in range(3)]).cpu().numpy()
    assert np.max(np.abs(out - rep)) < 1e-12 * np.sqrt(np.mean(rep ** 2))
    assert 0 < np.max(np.abs(out - ref)) < 1e-3 * np.sqrt(np.mean(ref ** 2))
    eng.rng_fast = 0
    assert np.array_equal(eng.generate_td(3, r0=5).cpu().numpy(), ref)


def test_td_draws_from_memory_equal_draws_in_registers():
    """ReplicaEngine.td_draws = "memory": the product READS its deviates (pta_td_plan.z, written once per batch by pta_rng_fill_normal)
    instead of generating them in its loop - same numbers, same MFMA order: bit-identical realisations, on ragged pulsars with odd
    TOA counts, a realisation count that is not a multiple of the 64-row groups, GWB + deterministic epilogue included."""
    import torch
    from pta_replicator_amd.engine import ReplicaEngine
    from pta_replicator_amd.simulate import ArrayTOAs, SimulatedPulsar, make_ideal
    rng = np.random.default_rng(8)
    psrs = []
    for a, n in enumerate((777, 300, 1025)):
        p = SimulatedPulsar(toas=ArrayTOAs(np.sort(rng.uniform(53000, 56000, n)), rng.uniform(0.3, 1.5, n)), name=f"J{a:04d}",
                            loc={"RAJ": 2.0 + 3 * a, "DECJ": -20.0 + 25 * a})
        make_ideal(p)
        psrs.append(p)
    eng = ReplicaEngine(psrs, seed=3)
    eng.set_white_noise(efac=1.1, log10_equad=-6.3)
    eng.set_jitter(log10_ecorr=-6.5, coarsegrain=0.1)
    eng.set_red_noise([-13.6, None, -14.0], [3.1, None, 4.2])
    eng.set_gwb(-14.2, 13. / 3.)
    eng.add_cgw(gwtheta=1.0, gwphi=2.0, mc=1e9, dist=20.0, fgw=2e-8, phase0=0.3, psi=0.4, inc=0.5, tref=53000 * 86400)
    eng.prepare().prepare_td()
    eng.td_draws = "registers"
    ref = eng.generate_td(70, r0=5)
    eng.td_draws = "memory"
    got = eng.generate_td(70, r0=5)
    assert torch.equal(got, ref)
    assert torch.equal(eng.generate_td(9, r0=40), ref[35:44])
    eng.td_draws = "registers"
    assert torch.equal(eng.generate_td(70, r0=5), ref)
    # chunks pipelined on a side stream (td_overlap, opt-in, for batches larger than td_chunk): deviates + GWB grid series of
    # chunk c + 1 prepared beside the product of chunk c, two buffers in rotation - the same realisations bit for bit, also when the
    # last chunk is short and when the buffers are re-used by a second call right behind the first
    eng.td_draws, eng.td_chunk, eng.td_overlap = "memory", 16, True
    for _ in range(2):
        assert torch.equal(eng.generate_td(70, r0=5), ref)
    eng.td_overlap = False
    assert torch.equal(eng.generate_td(70, r0=5), ref)
    eng.td_fill_beside_gwb = False        # the deviate fill on the product's own stream instead of beside the GWB grid stage (default: beside)
    assert torch.equal(eng.generate_td(70, r0=5), ref)
    eng.td_fill_beside_gwb = True
    assert torch.equal(eng.generate_td(70, r0=5), ref)


def _ragged_engine(components, sizes=(777, 90, 1025, 2601), jitter=True, seed=None, toas_per_epoch=3):
    """a small ragged array: odd counts, a count below one 256-column group, multi-TOA ECORR epochs, a pulsar without red noise"""
    from pta_replicator_amd.engine import ReplicaEngine
    from pta_replicator_amd.simulate import ArrayTOAs, SimulatedPulsar, make_ideal
    rng = np.random.default_rng(components if seed is None else seed)
    psrs = []
    for a, n in enumerate(sizes):
        ep = np.sort(rng.uniform(53000, 56000, n // toas_per_epoch + 1))
        mjd = (ep[:, None] + rng.uniform(0, 0.01, (len(ep), toas_per_epoch))).ravel()[:n]
        p = SimulatedPulsar(toas=ArrayTOAs(mjd, rng.uniform(0.3, 1.5, n)), name=f"J{a:04d}", loc={"RAJ": 2.0 + 3 * a, "DECJ": -20.0 + 25 * a})
        make_ideal(p)
        psrs.append(p)
    eng = ReplicaEngine(psrs, seed=3)
    eng.set_white_noise(efac=1.1, log10_equad=-6.3)
    if jitter:
        eng.set_jitter(log10_ecorr=-6.5, coarsegrain=0.1)
    P = len(sizes)
    eng.set_red_noise([None if a == 1 else -13.6 - 0.2 * a for a in range(P)], [None if a == 1 else 3.1 + 0.4 * (a % 3) for a in range(P)], components=components)
    return eng.prepare()


@pytest.mark.parametrize("components,jitter,sizes", [(30, True, None), (32, True, None), (29, True, None), (29, False, None), (10, True, None), (2, True, None),
                                                     (13, False, None), (30, True, (2, 3, 17, 64, 65, 255, 256, 257, 511, 513))])
def test_td_covariance_walk_kernel_equals_the_tile_kernel_inside_a_nan_slab(components, jitter, sizes):
    """pta_td_cov_assemble_walk (a wave keeps the phi-scaled operand of its 64 columns in registers and walks down the rows) against
    pta_td_cov_assemble_all (64 x 128 tiles through LDS), K = 60 / 64 / 58 / 20 / 4 / 26: whole, cut and missing k-steps, with and without
    ECORR.  The design matrix is a VIEW into a slab that is NaN in front of and behind it - round 4's withdrawn kernel read rows k >= K behind
    the matrix and multiplied them by zero (0 x NaN = NaN: scripts/gpu_r5_nan_repro.py); any such read shows up here.  The lower
    triangles agree to rounding (phi enters on the other operand) and nothing outside them is written."""
    import torch
    eng = _ragged_engine(components, jitter=jitter) if sizes is None else _ragged_engine(components, sizes=sizes, jitter=jitter)   # orders around every tile / group / segment boundary
    K, N = eng.plan.rn_k, eng.n_toa
    slab = torch.full((4 * N + K * N + 4 * N,), float("nan"), dtype=torch.float64, device="cuda")
    slab[4 * N:4 * N + K * N] = eng.d_Ft.reshape(-1)
    eng.d_Ft = slab[4 * N:4 * N + K * N].view(K, N)
    eng.prepare_td()
    assert eng.td_cov_kernel_used == "walk"
    got = {}
    for kernel, variant in (("tile", 0), ("walk", 0), ("walk", 1)):                    # variant 0 = default = 1 (the argument is reserved)
        eng.d_Ltd.fill_(float("nan"))
        eng.td_cov_walk_variant = variant
        eng.td_assemble(kernel=kernel)
        got[(kernel, variant)] = eng.d_Ltd.clone()
    eng.td_cov_walk_variant = 0
    for a in range(eng.P):
        n, ld, pos = int(eng.counts[a]), eng.td_ld[a], int(eng.td_pos[a])
        v1 = got[("tile", 0)][pos:pos + n * ld].view(n, ld)[:, :n].cpu().numpy()
        lo, up = np.tril_indices(n), np.triu_indices(n, 1)
        for variant in (0, 1):
            v2 = got[("walk", variant)][pos:pos + n * ld].view(n, ld)[:, :n].cpu().numpy()
            assert np.all(np.isfinite(v2[lo])), (a, variant)
            assert np.max(np.abs(v1[lo] - v2[lo])) < 1e-13 * np.max(np.abs(v1[lo])), (a, variant)
            assert np.all(np.isnan(v2[up])), (a, variant)                              # the upper triangle is not touched
            pad = got[("walk", variant)][pos:pos + n * ld].view(n, ld)[:, n:].cpu().numpy()
            assert np.all(np.isnan(pad)), (a, variant)                                 # nor the row padding
        assert np.array_equal(got[("walk", 0)][pos:pos + n * ld].view(n, ld)[:, :n].cpu().numpy()[lo],
                              got[("walk", 1)][pos:pos + n * ld].view(n, ld)[:, :n].cpu().numpy()[lo]), a   # deterministic
    out = eng.generate_td(3)                                                          # and the factorisation is happy with it
    assert bool(torch.isfinite(out).all())


def _poison_allocator(torch, sizes_mb=(1, 3, 17, 64, 200, 700)):
    """leave NaN-filled blocks of assorted sizes in the caching allocator's pools: whatever is allocated next (factor buffers, potrf
    workspaces, operand temporaries) starts out as NaN, so a kernel that consumes bytes it did not produce cannot go unnoticed"""
    blocks = [torch.full((mb * (1 << 20) // 8,), float("nan"), dtype=torch.float64, device="cuda") for mb in sizes_mb for _ in range(2)]
    del blocks


def _lower_factors(eng, torch):
    return [torch.tril(eng.d_Ltd[int(eng.td_pos[a]):int(eng.td_pos[a]) + eng.td_nst[a] * eng.td_ld[a]].view(eng.td_nst[a], eng.td_ld[a])[:, :eng.td_nst[a]]).clone()
            for a in range(eng.P)]


def _prepare_td_serialised(eng, torch):
    """prepare_td() with a device-wide synchronisation around every stage (the reference of the stress test below)"""
    torch.cuda.synchronize()
    eng.d_Ltd.fill_(float("nan"))
    torch.cuda.synchronize()
    eng.td_assemble()
    torch.cuda.synchronize()
    eng.td_factorise()
    torch.cuda.synchronize()


@pytest.mark.parametrize("kernel", ["walk", "tile"])
def test_td_prepare_without_host_sync_is_bit_equal_to_a_serialised_run(kernel):
    """three engines back to back in one process - a uniform batch, a ragged array and BASELINE config 2's shape (three orders,
    two of them odd; here at a tenth of the TOA counts so that 20 repetitions fit the test budget, the full shape runs in the next test) -
    with the allocator's free blocks poisoned with NaN before every prepare_td(), no host synchronisation between td_assemble() and
    td_factorise() (the product path has none since round 5), 20 repetitions: every factor bit-equal to the one a fully serialised run
    (device-wide synchronisation around every stage) produced."""
    import torch
    engines = [_ragged_engine(30, sizes=(1000,) * 6, seed=11), _ragged_engine(30, seed=12), _ragged_engine(30, sizes=(776, 2303, 3503), seed=13)]
    refs = []
    for eng in engines:
        eng.td_cov_kernel = kernel
        eng.prepare_td()
        _prepare_td_serialised(eng, torch)
        refs.append(_lower_factors(eng, torch))
    for rep in range(20):
        for eng, ref in zip(engines, refs):
            eng.d_Ltd = None                                                            # the factor buffer's block goes back to the pool ...
            _poison_allocator(torch)                                                    # ... and is poisoned with the rest
            eng.prepare_td()                                                            # fresh factor buffer and workspaces out of the poisoned pools
            for a, (got, want) in enumerate(zip(_lower_factors(eng, torch), ref)):
                assert torch.equal(got, want), (rep, eng.P, a)


def test_td_prepare_without_host_sync_config2_shape():
    """the same at BASELINE config 2's full TOA counts (7758 / 23 023 / 35 037: 14.6 GB of factors, one ragged schedule), 3 repetitions"""
    import torch
    if torch.cuda.get_device_properties(0).total_memory < 100e9:
        pytest.skip("needs ~45 GB of device memory")
    eng = _ragged_engine(30, sizes=(7758, 23023, 35037), seed=14, toas_per_epoch=4)
    eng.prepare_td()
    _prepare_td_serialised(eng, torch)
    ref = _lower_factors(eng, torch)
    for rep in range(3):
        eng.d_Ltd = None
        _poison_allocator(torch, sizes_mb=(1, 64, 700, 4000, 15000))
        eng.prepare_td()
        for a, want in enumerate(ref):
            n, ld, pos = eng.td_nst[a], eng.td_ld[a], int(eng.td_pos[a])
            got = torch.tril(eng.d_Ltd[pos:pos + n * ld].view(n, ld)[:, :n])
            assert torch.equal(got, want), (rep, a)
            del got
    out = eng.generate_td(2)
    assert bool(torch.isfinite(out).all())
