"""BASELINE.json configs 2, 4 and 5 at their full sizes on the MI355X (run with -m gpu): the CPU reference cannot
produce these in test time, so they are checked through size-independent properties plus oracle spot checks."""
import numpy as np
import pytest

from helpers import load, relrms
from oracle import pta_oracle as po

pytestmark = pytest.mark.gpu


def _fused_equals_replay(eng, r):
    out = eng.generate(1, r0=r).cpu().numpy()[0]
    rep = eng.replay([eng.dump_draws(r)]).cpu().numpy()[0]
    assert np.all(np.isfinite(out))
    assert np.max(np.abs(out - rep)) < 1e-12 * np.sqrt(np.mean(rep ** 2))
    return out


def _config2_engine(seed=222):
    """BASELINE.json config 2: B1855+09 (the reference's real tim file: 7758 unsorted TOAs, 4 backends, multi-TOA 1 s epochs) +
    B1937+21 / J1909-3744 synthesised at their par-file NTOA (23023 / 35037; test_partim/par/*.par:17), per-backend EFAC / EQUAD / ECORR
    (coarsegrain 1 s as in notebook cell 9) + per-pulsar red noise."""
    from pta_replicator_amd.engine import ReplicaEngine
    from pta_replicator_amd.simulate import ArrayTOAs, SimulatedPulsar, make_ideal
    z = load("c2_b1855.npz")
    backends = [str(b) for b in z["backends"]]
    rng = np.random.default_rng(2)
    psrs = []
    mjd = z["mjd_hi"].astype(np.longdouble) + z["mjd_lo"].astype(np.longdouble)
    flags = [{"f": backends[i]} for i in z["flag_index"]]
    psrs.append(SimulatedPulsar(toas=ArrayTOAs(mjd, z["err_us"], flags), name="B1855+09", loc={"RAJ": 18.96, "DECJ": 9.72}))
    for name, n, lo, hi, loc in (("B1937+21", 23023, 53420.0, 59070.0, {"RAJ": 19.66, "DECJ": 21.58}),
                                 ("J1909-3744", 35037, 53292.0, 59070.0, {"RAJ": 19.16, "DECJ": -37.74})):
        ep = np.sort(rng.uniform(lo, hi, n // 8 + 1))
        m = (ep[:, None] + rng.uniform(0, 0.02, (len(ep), 8))).ravel()[:n]        # ~8 sub-band TOAs per epoch
        fl = [{"f": backends[int(k)]} for k in rng.integers(0, 4, len(ep)).repeat(8)[:n]]
        psrs.append(SimulatedPulsar(toas=ArrayTOAs(m, np.exp(rng.uniform(np.log(0.05), np.log(30.0), n)), fl), name=name, loc=loc))
    for p in psrs:
        make_ideal(p)
    P = len(psrs)
    eng = ReplicaEngine(psrs, seed=seed)
    eng.set_white_noise(efac=[z["efac"]] * P, log10_equad=[z["log10_equad"]] * P, flags=[list(z["efac_flags"])] * P)
    eng.set_jitter(log10_ecorr=[z["log10_ecorr"]] * P, flags=[list(z["ecorr_flags"])] * P, coarsegrain=1.0 / 86400.0)
    eng.set_red_noise([float(z["rn_log10_amp"]), -13.5, -14.2], [float(z["rn_gamma"]), 2.5, 4.0], components=30)
    eng.prepare()
    return eng, psrs, z


def test_config2_test_partim_set_per_backend_noise_256_realisations():
    """B1855+09 (real tim: 7758 unsorted TOAs, 4 backends) + B1937+21 / J1909-3744 synthesised at their par-file
    NTOA (23023 / 35037), per-backend EFAC/EQUAD/ECORR (coarsegrain 1 s as in notebook cell 9) + per-pulsar RN."""
    eng, psrs, z = _config2_engine()
    P = len(psrs)
    assert eng.n_toa == 7758 + 23023 + 35037
    out = eng.generate(256).cpu().numpy()
    assert out.shape == (256, eng.n_toa) and np.all(np.isfinite(out))
    one = _fused_equals_replay(eng, 17)
    assert np.array_equal(one, out[17])
    # oracle on the dumped draws of one realisation, EVERY pulsar: the real tim file and the two synthesised ones at their
    # par-file NTOA (per-backend EFAC / EQUAD / ECORR with 1 s epochs over ~8 sub-band TOAs + per-pulsar RN), 1e-10 relative RMS
    d = eng.dump_draws(3)
    rn_par = [(float(z["rn_log10_amp"]), float(z["rn_gamma"])), (-13.5, 2.5), (-14.2, 4.0)]
    tf = None
    for a in range(P):
        tfa = np.array([f["f"] for f in psrs[a].toas.flags])
        tf = tfa if a == 0 else tf
        n = int(eng.counts[a])
        ref = po.measurement_noise_dt(eng.sigma_s[a], po.flag_vector(tfa, z["efac_flags"], z["efac"], n),
                                      po.flag_vector(tfa, z["efac_flags"], 10 ** z["log10_equad"], n), *d["wn"][a])
        epoch_of, ne, first, _ = po.quantize(eng.mjd[a], tfa, dt=1.0 / 86400.0)
        ref = ref + po.jitter_dt(epoch_of, po.jitter_ecorr_vector(ne, first, z["log10_ecorr"], tfa, z["ecorr_flags"]), d["ecorr"][a])
        ref = ref + po.red_noise_dt(psrs[a].toas.table["tdbld"], rn_par[a][0], rn_par[a][1], d["rn"][a])
        assert relrms(out[3, eng.off[a]:eng.off[a + 1]], ref) < 1e-10, a
    # ensemble statistics over 256 realisations: white + ECORR variance of the 430_ASP backend TOAs of pulsar 0
    sel = np.where(tf == "430_ASP")[0]
    k = list(z["efac_flags"]).index("430_ASP")
    kj = list(z["ecorr_flags"]).index("430_ASP")
    rn_part = eng.replay([eng.dump_draws(r) for r in range(8)], per_signal=True)["rn"].cpu().numpy()[:, sel]
    white = out[:8, sel] - rn_part
    expect = np.mean((z["efac"][k] * eng.sigma_s[0][sel]) ** 2 + (z["efac"][k] * 10 ** z["log10_equad"][k]) ** 2) + (10 ** z["log10_ecorr"][kj]) ** 2
    assert abs(np.var(white) / expect - 1) < 0.15


def test_config2_td_mode():
    """BASELINE.json config 2 ("batched Cholesky bring-up") through the DENSE path: the real unsorted 4-backend B1855+09 (its same-epoch
    ECORR blocks are scattered through the matrix) and the 23023- / 35037-TOA pulsars - three different orders (one of them odd twice
    over), factored as ONE ragged schedule (pta_potrf_ragged).  Per pulsar, 64 sampled rows i: (L L^T)[i, :] against the oracle's covariance
    row (oracle/pta_oracle.py: td_covariance_rows) at 1e-12 ||C||_max; the whole 7758^2 factor against LAPACK; generate_td(256) finite;
    L z of the dumped deviates on those rows at 1e-10; memory-draw == register-draw; ragged == per-matrix schedule."""
    import torch
    eng, psrs, z = _config2_engine(seed=223)
    P = len(psrs)
    eng.td_potrf_mode = "ragged"
    eng.prepare_td()
    assert eng.td_potrf_mode_used == "ragged"
    rn_par = [(float(z["rn_log10_amp"]), float(z["rn_gamma"])), (-13.5, 2.5), (-14.2, 4.0)]
    rng = np.random.default_rng(64)
    R = 256
    eng.td_draws = "memory"
    out = eng.generate_td(R, r0=5)
    assert bool(torch.isfinite(out).all())
    eng.td_draws = "registers"
    assert torch.equal(eng.generate_td(4, r0=5), out[:4])           # the same deviates generated inside the product: bit-identical
    eng.td_draws = "memory"
    draws = eng.dump_draws_td(7)
    got7 = out[2].cpu().numpy()
    rows_of = {}
    for a in range(P):
        n = int(eng.counts[a])
        tfa = np.array([f["f"] for f in psrs[a].toas.flags])
        sigma2 = (po.flag_vector(tfa, z["efac_flags"], z["efac"], n) * eng.sigma_s[a]) ** 2 + \
                 (po.flag_vector(tfa, z["efac_flags"], z["efac"], n) * po.flag_vector(tfa, z["efac_flags"], 10 ** z["log10_equad"], n)) ** 2
        epoch_of, ne, first, _ = po.quantize(eng.mjd[a], tfa, dt=1.0 / 86400.0)
        ecv = po.jitter_ecorr_vector(ne, first, z["log10_ecorr"], tfa, z["ecorr_flags"])
        rows = np.unique(np.concatenate([[0, 1, n // 2, n - 2, n - 1], rng.integers(0, n, 64)]))
        rows_of[a] = rows
        Cref = po.td_covariance_rows(eng.tdb_s[a], rn_par[a][0], rn_par[a][1], 30, sigma2, epoch_of, ecv, rows)
        L = eng.td_factor(a)                                          # device copy, upper triangle zeroed
        ridx = torch.as_tensor(rows, device=L.device)
        Lr = L[ridx]
        LLt = (Lr @ L.T).cpu().numpy()
        scale = np.max(np.abs(np.diag(Cref[:, rows])))
        assert np.max(np.abs(LLt - Cref)) < 1e-12 * scale, (a, np.max(np.abs(LLt - Cref)) / scale)
        # L z on the sampled rows (NumPy on the copied-back rows of the device factor)
        ref = Lr.cpu().numpy() @ draws["td"][a]
        sl = slice(int(eng.off[a]), int(eng.off[a + 1]))
        assert np.max(np.abs(got7[sl][rows] - ref)) < 1e-10 * np.sqrt(np.mean(ref ** 2)), a
        if a == 0:   # the real tim file's pulsar: the whole factor against LAPACK
            Cfull = po.td_covariance(eng.tdb_s[a], rn_par[a][0], rn_par[a][1], 30, sigma2, epoch_of, ecv)
            Lref = np.linalg.cholesky(Cfull)
            Ld = L.cpu().numpy()
            assert np.max(np.abs(Ld - Lref)) < 1e-10 * np.max(np.abs(Lref))
        del L, Lr
    # the per-matrix schedule (batches of one) gives the same factors
    fr = {a: eng.td_factor(a)[torch.as_tensor(rows_of[a], device="cuda")].cpu().numpy() for a in range(P)}
    eng.td_assemble()
    eng.td_factorise(mode="uniform")
    for a in range(P):
        fu = eng.td_factor(a)[torch.as_tensor(rows_of[a], device="cuda")].cpu().numpy()
        assert np.max(np.abs(fu - fr[a])) < 1e-11 * np.max(np.abs(fu)), a


def test_config4_headline_array_with_cgw():
    """68 x 5000 + one continuous-wave source (the reference test's CW parameters): the deterministic term is added
    once per TOA and is identical in every realisation."""
    from pta_replicator_amd.engine import ReplicaEngine
    from bench import configure_engine, headline_array
    psrs, noise = headline_array(68, 5000)
    cw = dict(gwtheta=np.pi / 2, gwphi=2.5, mc=1e9, dist=5.0, fgw=1e-8, phase0=0.5, psi=1.5, inc=np.pi / 4, pdist=1.0,
              pphase=None, psrTerm=True, evolve=True, phase_approx=False, tref=53000 * 86400)

    def build(with_cw):
        e = ReplicaEngine(psrs, seed=4)
        configure_engine(e, noise)
        if with_cw:
            e.add_cgw(**cw)
        return e.prepare()

    a, b = build(True), build(False)
    oa, ob = a.generate(3).cpu().numpy(), b.generate(3).cpu().numpy()
    det = a.d_det.cpu().numpy()
    assert np.max(np.abs((oa - ob) - det[None, :])) < 1e-12 * np.sqrt(np.mean(det ** 2))
    for p in (0, 33, 67):   # oracle waveform on three pulsars
        ra, dec = psrs[p].loc["RAJ"] * np.pi / 12, psrs[p].loc["DECJ"] * np.pi / 180
        ref = po.cgw_dt(a.mjd[p], np.pi / 2 - dec, ra, **cw)
        assert relrms(det[a.off[p]:a.off[p + 1]], ref) < 2e-10      # evolving phase: see TOL_CGW in tests/test_gpu_parity.py
    _fused_equals_replay(a, 2)
    # the summed residual (GWB + RN + EFAC/EQUAD + ECORR + CGW) of one realisation against the CPU oracle, EVERY pulsar
    from helpers import oracle_realisation
    ref = oracle_realisation(po, psrs, noise, a.dump_draws(1))
    worst = 0.0
    for p in range(68):
        ra, dec = psrs[p].loc["RAJ"] * np.pi / 12, psrs[p].loc["DECJ"] * np.pi / 180
        full = ref[p] + po.cgw_dt(a.mjd[p], np.pi / 2 - dec, ra, **cw)
        worst = max(worst, relrms(oa[1, a.off[p]:a.off[p + 1]], full))
    assert worst < 1e-10, worst
    # one GPU's share of the config (2048 of the 16384 realisations, rank 3's range): batch == one-by-one, finite, and the
    # ensemble variance of one pulsar's residuals is stationary across the shard
    import torch
    from pta_replicator_amd.distributed import shard_range
    lo, hi = shard_range(16384, rank=3, world=8)
    assert (lo, hi) == (6144, 8192)
    big = a.generate(hi - lo, r0=lo)
    assert big.shape == (2048, a.n_toa) and bool(torch.isfinite(big).all())
    for r in (0, 1000, 2047):
        assert torch.equal(big[r], a.generate(1, r0=lo + r)[0])
    sel = big[:, a.off[5]:a.off[6]] - a.d_det[a.off[5]:a.off[6]]
    v1, v2 = float(sel[:1024].var()), float(sel[1024:].var())
    assert abs(v1 / v2 - 1) < 0.2
    del big, sel
    torch.cuda.empty_cache()


def test_config5_ska_scale_anisotropic():
    """200 pulsars x 10000 TOAs, anisotropic GWB through the l <= 4 ORF basis: the 20100-pair basis evaluation that takes
    the reference ~15 min of Python is one kernel launch; spot-checked pair by pair against the oracle."""
    import time
    import torch
    from pta_replicator_amd import spharmORFbasis as anis
    from pta_replicator_amd.engine import ReplicaEngine
    from pta_replicator_amd.simulate import ArrayTOAs, SimulatedPulsar, make_ideal
    P, N, lmax = 200, 10000, 4
    rng = np.random.default_rng(200)
    raj, decj = rng.uniform(0, 24, P), np.degrees(np.arcsin(rng.uniform(-1, 1, P)))
    psrs = []
    for a in range(P):
        psr = SimulatedPulsar(toas=ArrayTOAs(np.sort(rng.uniform(53000, 60305, N)), 0.5), name=f"J{a:04d}", loc={"RAJ": raj[a], "DECJ": decj[a]})
        make_ideal(psr)
        psrs.append(psr)
    locs = po.psr_locs_equatorial([p.loc for p in psrs])
    anis.correlated_basis_device(locs[:8], lmax)                       # first launch of the kernel (code object load) outside the timing
    torch.cuda.synchronize(); t0 = time.perf_counter()
    basis = anis.correlated_basis_device(locs, lmax)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    # measured (bench_extras.orf_numbers): pta_orf_basis 0.38 ms + pta_orf_combine 0.005 ms on the device, ~1 ms of host pair separations
    # (native since round 6; 61 ms as a Python loop) - 12-15 min in the reference.  The DEVICE time is asserted with HIP events on the
    # launch stream (a wall-clock bound of 0.7 s flakes on a loaded host); the wall clock only has to stay far from the reference's
    from pta_replicator_amd import _lib as lib_, device as dv_
    zc_d, locs_d = dv_.f64(anis.pair_zeta_cos(locs)), dv_.f64(np.ascontiguousarray(locs))
    tmp = dv_.zeros((25, P, P))
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    lib_.call("pta_orf_basis", dv_.ptr(locs_d), dv_.ptr(zc_d), P, lmax, dv_.ptr(tmp), dv_.stream_ptr())
    ev[1].record(); torch.cuda.synchronize()
    assert ev[0].elapsed_time(ev[1]) < 5.0, ev[0].elapsed_time(ev[1])          # ms: 13x the measured 0.38
    assert torch.equal(tmp, basis)
    assert dt < 2.0, dt
    basis = basis.cpu().numpy()
    assert basis.shape == (25, P, P) and np.all(np.isfinite(basis)) and np.allclose(basis, basis.transpose(0, 2, 1), rtol=0, atol=0)
    for (a, b) in ((0, 0), (3, 77), (150, 199), (42, 43), (9, 120)):
        for ll in range(lmax + 1):
            zeta = po.calczeta(locs[a, 0], locs[b, 0], locs[a, 1], locs[b, 1])
            plus = [po.arbCompFrame_ORF(mm, ll, zeta) for mm in range(ll + 1)]
            gamma_ml = [(-1) ** mm * plus[mm] for mm in range(1, ll + 1)][::-1] + plus
            for mi in range(2 * ll + 1):
                ref = po.real_rotated_Gammas(mi - ll, ll, locs[a, 0], locs[b, 0], locs[a, 1], locs[b, 1], gamma_ml)
                assert abs(basis[ll * ll + mi, a, b] - ref) < 1e-11 * (abs(ref) + 0.03)
    # anisotropy coefficients: isotropic term + 10 % perturbations, redrawn until the ORF is positive definite
    crng = np.random.default_rng(200)
    for _ in range(50):
        clm = np.concatenate([[np.sqrt(4 * np.pi)], 0.1 * crng.standard_normal(24)])
        orf = 2 * np.tensordot(clm, basis, axes=1)
        if np.all(np.linalg.eigvalsh(orf) > 1e-6):
            break
    else:
        pytest.fail("no positive-definite anisotropic ORF found")
    # the same ORF from the UNMODIFIED reference (oracle/gen_c5_orf.py -> tests/golden/c5_orf_lmax4.npz: all 20 100 pairs through
    # the reference's own spharmORFbasis functions): geometry, anisotropy coefficients, every degree l, and the sum
    z = load("c5_orf_lmax4.npz")
    assert np.array_equal(z["raj"], raj) and np.array_equal(z["decj"], decj) and np.allclose(z["clm"], clm, rtol=0, atol=0)
    # l <= 3: 1e-12 absolute on every pair (measured 6e-14 at l = 3).  l = 4: 2e-12 on the pairs up to 170 degrees apart (measured
    # 1e-12); the 19 pairs closer to antipodal are where the reference's own float64 sums lose their digits - at the worst one
    # (174.9 degrees) the reference is 1e-10 from a longdouble evaluation of its own formula and the device 2.3e-10
    # (tests/test_hostcheck.py::test_orf_l4_near_antipodal_pairs_are_ill_conditioned_in_the_reference) - so they are held to 2e-10,
    # 1e-10 of the ORF's scale (its diagonal is 2)
    zeta_deg = np.degrees(np.arccos(np.clip(np.sin(locs[:, 1])[:, None] * np.sin(locs[:, 1])[None, :] * np.cos(locs[:, 0][:, None] - locs[:, 0][None, :])
                                            + np.cos(locs[:, 1])[:, None] * np.cos(locs[:, 1])[None, :], -1, 1)))
    for ll in range(lmax + 1):
        dev_l = 2 * np.tensordot(clm[ll * ll:(ll + 1) ** 2], basis[ll * ll:(ll + 1) ** 2], axes=1)
        err = np.abs(dev_l - z["orf_l"][ll])
        if ll < 4:
            assert err.max() < 1e-12, (ll, err.max())
        else:
            assert err[zeta_deg <= 170.0].max() < 2e-12 and err.max() < 2e-10, (err[zeta_deg <= 170.0].max(), err.max())
    assert np.max(np.abs(orf - z["orf"])) < 2e-10
    eng = ReplicaEngine(psrs, seed=5)
    eng.set_white_noise(efac=1.0, log10_equad=-6.5)
    eng.set_red_noise(-14.0, 3.0)
    eng.set_gwb(-14.6733, 13. / 3., clm=clm, lmax=lmax)
    eng.prepare()
    assert np.max(np.abs(eng.ORF.cpu().numpy() - orf)) < 1e-13 and np.max(np.abs(eng.ORF.cpu().numpy() - z["orf"])) < 2e-10
    M = eng.d_M.cpu().numpy()
    assert np.max(np.abs(M @ M.T - orf)) < 1e-12
    Mref = np.linalg.cholesky(z["orf"])                       # LAPACK on the reference's matrix (red_noise.py:235)
    assert np.max(np.abs(M - Mref)) < 1e-10 * np.max(np.abs(Mref))
    out = eng.generate(2).cpu().numpy()
    assert out.shape == (2, P * N) and np.all(np.isfinite(out))
    # ---- one whole realisation against the CPU oracle on the dumped deviates, EVERY pulsar: GWB through the reference's ORF and
    # LAPACK's factor of it (red_noise.py:224-287), RN (:106-135), EFAC / t2EQUAD (white_noise.py:47-125); 1e-10 relative RMS
    d = eng.dump_draws(1)
    mjd = [np.asarray(p.toas.get_mjds().value, dtype=np.float64) for p in psrs]
    grid = po.gwb_grid([float(m.min()) for m in mjd], [float(m.max()) for m in mjd])
    C = po.gwb_spectrum(grid["f"], grid["dur"], grid["howml"], -14.6733, 13. / 3.)
    res_gw, _ = po.gwb_dt(grid, Mref, d["gwb"], C, [m * 86400 for m in mjd])
    worst = 0.0
    for a in range(P):
        sig = np.asarray(psrs[a].toas.get_errors().to("s").value, dtype=np.float64)
        ref = po.measurement_noise_dt(sig, np.ones(N), np.ones(N) * 10 ** -6.5, *d["wn"][a])
        ref = ref + po.red_noise_dt(psrs[a].toas.table["tdbld"], -14.0, 3.0, d["rn"][a]) + res_gw[a]
        worst = max(worst, relrms(out[1, eng.off[a]:eng.off[a + 1]], ref))
    assert worst < 1e-10, worst
    _fused_equals_replay(eng, 1)


def test_bench_helpers_small_shapes():
    """the functions the bench line is assembled from (grid cells, ragged TD numbers) on small shapes: every key the line carries is there
    and finite - the driver's bench run depends on them."""
    import bench
    c = bench.grid_cell(3, 122, td=True)
    t, d = c["throughput"], c["td"]
    assert t["realisations_per_s"] > 0 and 0 < t["tile_fill"] <= 1 and t["dominant_kernel"] in t["kernels_ms"]
    assert d["finite"] and d["potrf_ms"] > 0 and d["realisations_per_s"] > 0 and d["schedule"] == "uniform"
    r = bench.td_ragged_numbers(R=32, counts=(600, 1301, 2050, 130))
    assert r["schedule"] == "ragged" and r["finite"] and r["potrf_TFLOPs"] > 0 and r["per_matrix_schedule"]["potrf_ms"] > 0
    assert r["n_toa_total"] == 600 + 1301 + 2050 + 130
    ck = bench.engine_clock_during(lambda: bench.grid_cell, 0.05)      # a no-op load: the probe alone
    assert ck is None or 0.05 < ck["GHz"] < 3.5
