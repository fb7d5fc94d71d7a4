"""Generate ``tests/golden/*.npz`` by running the UNMODIFIED reference under stubs.

TEST INFRASTRUCTURE ONLY.  Run in the build container (needs ``/root/reference``):

    python oracle/gen_golden.py

The fixtures hold the inputs (so that tests can rebuild the same pulsars without the
reference) and the reference's outputs: the per-signal delays every ``add_*`` call hands to
``update_added_signals`` and, for ``add_gwb``, the intermediates captured by wrapping
``np.linalg.cholesky`` / ``np.fft.ifft`` while the reference runs.  MJDs are longdouble in
the mock (as in PINT) and are stored as a (hi, lo) float64 pair.

Fixtures
--------
c1_small.npz     config 1: the reference's own test recipe (tests/test_against_libstempo.py:19-53)
                 on test_partim_small, once on the raw tim-file MJDs (Nf = 3001) and once with the
                 1 us nudge that puts Nf on the libstempo side of the knife-edge (Nf = 3000,
                 SURVEY.md §0.4); also carries the shipped libstempo golden vector.
c2_b1855.npz     config 2 bring-up: the real, unsorted, 4-backend B1855+09.tim with per-backend
                 EFAC/EQUAD/ECORR and RN from noise_dicts/ng15_dict.json (notebook cell 9 recipe).
orf_basis.npz    spharmORFbasis.correlated_basis up to lmax = 4 incl. coincident/antipodal pairs.
variants.npz     the branches the shipped vector does not pin: turnover, userSpec,
                 no_correlations, lmax=2, tnequad, per-flag WN/ECORR with multi-TOA epochs,
                 default RN convention, every add_cgw branch.
c3_mini.npz      a 6-pulsar miniature of config 3 (HD GWB + RN + EFAC/EQUAD + ECORR, notebook seeds).
cw_catalog.npz   add_catalog_of_cws through both numba kernels of the reference (40 and 1200 sources).
rn_modes.npz     add_red_noise(modes=...) in both phase conventions.
transients.npz   add_burst / add_noise_transient / add_gw_memory with Gaussian-sine waveforms.
population.npz   add_gwb_plus_outlier_cws on a 1500-binary synthetic population (userSpec GWB + 22 outlier CWs).
"""
import glob
import json
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, _HERE)
import run_reference as rr  # noqa: E402

OUT = os.path.join(os.path.dirname(_HERE), "tests", "golden")
REF = rr.REFERENCE_ROOT


def split_ld(x):
    x = np.asarray(x, dtype=np.longdouble)
    hi = x.astype(np.float64)
    lo = (x - hi.astype(np.longdouble)).astype(np.float64)
    return hi, lo


class Capture:
    """Record the arguments/results of np.linalg.cholesky and np.fft.ifft while active."""

    def __enter__(self):
        self.chol_in, self.chol_out, self.ifft_in, self.ifft_out = [], [], [], []
        self._chol, self._ifft = np.linalg.cholesky, np.fft.ifft

        def chol(a):
            r = self._chol(a)
            self.chol_in.append(np.array(a))
            self.chol_out.append(np.array(r))
            return r

        def ifft(a, *args, **kw):
            r = self._ifft(a, *args, **kw)
            self.ifft_in.append(np.array(a))
            self.ifft_out.append(np.array(r))
            return r

        np.linalg.cholesky, np.fft.ifft = chol, ifft
        return self

    def __exit__(self, *exc):
        np.linalg.cholesky, np.fft.ifft = self._chol, self._ifft


def sig(psr, key):
    """value array of an added signal (seconds, except '<name>_gwb' which the reference stores in days)."""
    return np.array(psr.added_signals_time[f"{psr.name}_{key}"].value, dtype=np.float64)


def pulsar_inputs(prefix, psrs, mjd0):
    d = {}
    for i, (p, m) in enumerate(zip(psrs, mjd0)):
        hi, lo = split_ld(m)
        d[f"{prefix}mjd_hi_{i}"] = hi
        d[f"{prefix}mjd_lo_{i}"] = lo
        d[f"{prefix}err_us_{i}"] = p.toas.errors_us
    d[prefix + "names"] = np.array([p.name for p in psrs])
    if all("RAJ" in p.loc for p in psrs):
        d[prefix + "raj_hours"] = np.array([p.loc["RAJ"] for p in psrs])
        d[prefix + "decj_deg"] = np.array([p.loc["DECJ"] for p in psrs])
    return d


# ------------------------------------------------------------------------------------------
def gen_c1(ref):
    pars = sorted(glob.glob(f"{REF}/test_partim_small/par/*.par"))
    tims = sorted(glob.glob(f"{REF}/test_partim_small/tim/*.tim"))
    out = {"libstempo_residuals": np.load(f"{REF}/tests/libstempo_test_residuals_efac_ecorr_rn_gwb_cgw.npz")["residuals"]}
    for tag, nudge_us in (("raw_", 0.0), ("nudged_", 1.0)):
        psrs, mjd0 = [], []
        for p, t in zip(pars, tims):
            name, loc = rr.read_par_loc(p)
            mjd, err, fl = rr.read_tim(t)
            if nudge_us and name == "JPSR02":
                mjd[0] += np.longdouble(nudge_us * 1e-6) / 86400
            psrs.append(rr.make_pulsar(ref, name, mjd, err, loc, fl))
            mjd0.append(mjd.copy())
        out.update(pulsar_inputs(tag, psrs, mjd0))
        with Capture() as cap:
            ref.red_noise.add_gwb(psrs, -14, 4.33, seed=123456)
        for ii, psr in enumerate(psrs):
            ref.white_noise.add_measurement_noise(psr, efac=1.00, log10_equad=None, seed=54321 + ii, tnequad=False)
            ref.white_noise.add_jitter(psr, log10_ecorr=np.log10(3e-7), seed=54321 + ii)
        for ii, psr in enumerate(psrs):
            ref.red_noise.add_red_noise(psr, -15, 4.2, components=30, Tspan=None, seed=12345 + ii,
                                        libstempo_convention=True)
        for psr in psrs:
            ref.deterministic.add_cgw(psr, gwtheta=np.pi / 2, gwphi=2.5, mc=1e9, dist=5.0, fgw=1e-8, phase0=0.5,
                                      psi=1.5, inc=np.pi / 4, pdist=1.0, pphase=None, psrTerm=True, evolve=True,
                                      phase_approx=False, tref=53000 * 86400)
        for k in ("gwb", "measurement_noise", "jitter", "red_noise", "cw"):
            out[tag + k] = np.array([sig(p, k) for p in psrs])
        out[tag + "residuals"] = np.array([p.toas.residuals_s() for p in psrs])
        out[tag + "ORF"] = cap.chol_in[0]
        out[tag + "M"] = cap.chol_out[0]
        out[tag + "Nf"] = np.array(cap.ifft_in[0].shape[1] // 2 + 1)
        if nudge_us:
            nf = int(out[tag + "Nf"])
            out[tag + "Res_f"] = cap.ifft_in[0][:, :nf]
            out[tag + "Res_t_used"] = np.real(cap.ifft_out[0])[:, 10:610]  # before the /dt of red_noise.py:279
        gold = out["libstempo_residuals"]
        res = out[tag + "residuals"]
        print(f"c1 {tag}: Nf={int(out[tag + 'Nf'])}  max|d|/rms vs libstempo =",
              [float(np.max(np.abs(res[i] - gold[i])) / np.sqrt(np.mean(res[i] ** 2))) for i in range(3)])
    np.savez_compressed(os.path.join(OUT, "c1_small.npz"), **out)


# ------------------------------------------------------------------------------------------
def noise_for(noise_params, p):
    """the notebook's cell-6 parsing of noise_dicts/ng15_dict.json (examples/add_noise.ipynb cell 6)."""
    d = {"equad": [], "efac": [], "ecorr": []}
    for ky, val in noise_params.items():
        if p in ky:
            if "equad" in ky:
                d["equad"].append([ky.replace(p + "_", "").replace("_log10_t2equad", ""), val])
            if "efac" in ky:
                d["efac"].append([ky.replace(p + "_", "").replace("_efac", ""), val])
            if "ecorr" in ky:
                d["ecorr"].append([ky.replace(p + "_", "").replace("_log10_ecorr", ""), val])
            if "gamma" in ky:
                d["rn_gamma"] = val
            if "log10_A" in ky:
                d["rn_log10_amp"] = val
    return d


def gen_c2(ref):
    name, _ = rr.read_par_loc(f"{REF}/test_partim/par/B1855+09.par")
    mjd, err, fl = rr.read_tim(f"{REF}/test_partim/tim/B1855+09.tim")
    noise = noise_for(json.load(open(f"{REF}/noise_dicts/ng15_dict.json")), name)
    backends = sorted({f["f"] for f in fl})
    out = {"name": np.array(name), "err_us": err, "backends": np.array(backends),
           "flag_index": np.array([backends.index(f["f"]) for f in fl])}
    out["mjd_hi"], out["mjd_lo"] = split_ld(mjd)
    efl = np.array([r[0] for r in noise["efac"]]); efv = np.array([float(r[1]) for r in noise["efac"]])
    eql = np.array([r[0] for r in noise["equad"]]); eqv = np.array([float(r[1]) for r in noise["equad"]])
    ecl = np.array([r[0] for r in noise["ecorr"]]); ecv = np.array([float(r[1]) for r in noise["ecorr"]])
    assert list(efl) == list(eql)
    out.update(efac_flags=efl, efac=efv, log10_equad=eqv, ecorr_flags=ecl, log10_ecorr=ecv,
               rn_log10_amp=np.array(noise["rn_log10_amp"]), rn_gamma=np.array(noise["rn_gamma"]))
    loc = {"RAJ": 18.96, "DECJ": 9.72}  # unused by WN/ECORR/RN (the par file is ELONG/ELAT: ephem branch unpinned)
    for cg_tag, cg in (("cg1s_", 1.0 / 86400.0), ("cg01_", 0.1)):
        psr = rr.make_pulsar(ref, name, mjd, err, loc, fl)
        ref.white_noise.add_measurement_noise(psr, efac=efv, log10_equad=eqv, flagid="f", flags=efl, seed=10660)
        ref.white_noise.add_jitter(psr, log10_ecorr=ecv, flagid="f", flags=ecl, coarsegrain=cg, seed=17763)
        ref.red_noise.add_red_noise(psr, log10_amplitude=noise["rn_log10_amp"], spectral_index=noise["rn_gamma"],
                                    components=30, seed=19870)
        for k in ("measurement_noise", "jitter", "red_noise"):
            out[cg_tag + k] = sig(psr, k)
        out[cg_tag + "residuals"] = psr.toas.residuals_s()
        # epoch structure straight from the reference's quantize_fast
        t, f, U = ref.white_noise.quantize_fast(np.array(mjd, dtype=np.float64), np.array([x["f"] for x in fl]), dt=cg)
        out[cg_tag + "n_epochs"] = np.array(U.shape[1])
        out[cg_tag + "epoch_of"] = np.argmax(U, axis=1)
        out[cg_tag + "aveflags"] = f
        print(f"c2 {cg_tag}: N={len(mjd)} epochs={U.shape[1]}")
    np.savez_compressed(os.path.join(OUT, "c2_b1855.npz"), **out)


# ------------------------------------------------------------------------------------------
def gen_orf(ref):
    rng = np.random.default_rng(424242)
    P = 7
    phi = rng.uniform(0, 2 * np.pi, P)
    theta = np.arccos(rng.uniform(-1, 1, P))
    # a coincident pair (zeta == 0 off the diagonal) and an exactly antipodal pair on the equator (zeta == pi)
    phi = np.concatenate([phi, [phi[2]], [0.0, np.pi]])
    theta = np.concatenate([theta, [theta[2]], [np.pi / 2, np.pi / 2]])
    locs = np.stack([phi, theta], axis=1)
    basis = np.array(ref.spharm.correlated_basis(locs, 4))
    zeta = np.array([[ref.spharm.calczeta(locs[a, 0], locs[b, 0], locs[a, 1], locs[b, 1]) for b in range(len(locs))]
                     for a in range(len(locs))])
    print("orf_basis:", basis.shape, "zeta==pi pairs:", int(np.sum(zeta == np.pi)))
    np.savez_compressed(os.path.join(OUT, "orf_basis.npz"), psr_locs=locs, basis=basis, zeta=zeta, lmax=np.array(4))


# ------------------------------------------------------------------------------------------
def synth_array(ref, P, N, seed, cadence_days=None, backends=("A_X", "B_Y"), burst=1):
    """synthetic pulsars: sorted uniform MJDs in [53000, 58478] (or bursts of `burst` TOAs within ~20 min)."""
    rng = np.random.default_rng(seed)
    psrs, mjd0 = [], []
    for a in range(P):
        if burst > 1:
            nep = N // burst
            ep = np.sort(rng.uniform(53000, 58478, nep))
            mjd = (ep[:, None] + rng.uniform(0, 0.014, (nep, burst))).ravel()
        else:
            mjd = np.sort(rng.uniform(53000, 58478, N))
        mjd = np.array(mjd, dtype=np.longdouble) + np.longdouble(rng.uniform(0, 1e-9, len(mjd)))
        perm = rng.permutation(len(mjd)) if a % 2 else np.arange(len(mjd))  # odd pulsars unsorted
        mjd = mjd[perm]
        err = rng.uniform(0.1, 2.0, len(mjd))
        if burst > 1:
            fl = [{"f": backends[(i // burst) % len(backends)]} for i in perm]
        else:
            fl = [{"f": backends[int(rng.integers(len(backends)))]} for _ in range(len(mjd))]
        loc = {"RAJ": float(rng.uniform(0, 24)), "DECJ": float(np.degrees(np.arcsin(rng.uniform(-1, 1))))}
        psrs.append(rr.make_pulsar(ref, f"J{a:04d}+00", mjd, err, loc, fl))
        mjd0.append(mjd.copy())
    return psrs, mjd0


def flag_arrays(prefix, psrs):
    d = {}
    for i, p in enumerate(psrs):
        d[f"{prefix}flag_{i}"] = np.array([f["f"] for f in p.toas.flags])
    return d


def gen_variants(ref):
    out = {}
    P, N = 4, 150

    def fresh():
        return synth_array(ref, P, N, seed=77, burst=3)

    psrs, mjd0 = fresh()
    out.update(pulsar_inputs("", psrs, mjd0))
    out.update(flag_arrays("", psrs))
    backends = np.array(["A_X", "B_Y"])
    out["backends"] = backends

    # --- add_gwb branches (each on a fresh copy so that TOAs are the ideal ones) ---
    userSpec = np.stack([np.logspace(-9.5, -7.2, 12), 1e-15 * (np.logspace(-9.5, -7.2, 12) / 3e-8) ** (-0.6)], axis=1)
    clm2 = np.array([np.sqrt(4 * np.pi), 0.3, -0.2, 0.25, 0.1, -0.15, 0.2, 0.05, -0.1])
    out["userSpec"] = userSpec
    out["clm2"] = clm2
    gwb_cases = {
        "gwb_turnover": dict(log10_amplitude=-14.2, spectral_index=13. / 3., seed=501, turnover=True, f0=3e-9, beta=1.2, power=2.0),
        "gwb_userspec": dict(log10_amplitude=-14.2, spectral_index=13. / 3., seed=502, userSpec=userSpec),
        "gwb_nocorr": dict(log10_amplitude=-14.5, spectral_index=3.9, seed=503, no_correlations=True),
        "gwb_lmax2": dict(log10_amplitude=-14.5, spectral_index=13. / 3., seed=504, clm=clm2, lmax=2),
        "gwb_grid": dict(log10_amplitude=-14.5, spectral_index=13. / 3., seed=505, npts=200, howml=4),
    }
    for tag, kw in gwb_cases.items():
        ps, _ = fresh()
        with Capture() as cap:
            ref.red_noise.add_gwb(ps, **kw)
        out[tag] = np.array([sig(p, "gwb") for p in ps])  # days
        if cap.chol_in:
            out[tag + "_ORF"] = cap.chol_in[0]
        out[tag + "_Nf"] = np.array(cap.ifft_in[0].shape[1] // 2 + 1)

    # --- white noise branches ---
    ps, _ = fresh()
    for ii, p in enumerate(ps):
        ref.white_noise.add_measurement_noise(p, efac=1.3, log10_equad=-6.3, seed=600 + ii, tnequad=True)
    out["wn_tnequad"] = np.array([sig(p, "measurement_noise") for p in ps])
    ps, _ = fresh()
    efac = np.array([1.1, 0.9]); l10eq = np.array([-6.5, -6.1]); l10ec = np.array([-6.2, -6.6])
    out.update(efac=efac, log10_equad=l10eq, log10_ecorr=l10ec)
    for ii, p in enumerate(ps):
        ref.white_noise.add_measurement_noise(p, efac=efac, log10_equad=l10eq, flagid="f", flags=backends, seed=610 + ii)
        ref.white_noise.add_jitter(p, log10_ecorr=l10ec, flagid="f", flags=backends, coarsegrain=0.1, seed=620 + ii)
    out["wn_flags"] = np.array([sig(p, "measurement_noise") for p in ps])
    out["jitter_flags"] = np.array([sig(p, "jitter") for p in ps])
    ps, _ = fresh()
    for ii, p in enumerate(ps):
        ref.white_noise.add_jitter(p, log10_ecorr=-6.4, coarsegrain=0.1, seed=630 + ii)
    out["jitter_scalar"] = np.array([sig(p, "jitter") for p in ps])

    # --- red noise, default (sin-first, absolute-t) convention, 15 components ---
    ps, _ = fresh()
    for ii, p in enumerate(ps):
        ref.red_noise.add_red_noise(p, -13.8, 3.3, components=15, seed=640 + ii)
    out["rn_default"] = np.array([sig(p, "red_noise") for p in ps])

    # --- add_cgw branches ---
    base = dict(gwtheta=1.1, gwphi=4.0, mc=3e9, dist=40.0, fgw=2.2e-8, phase0=1.3, psi=0.4, inc=1.0, tref=53000 * 86400)
    cgw_cases = {
        "cgw_evolve": dict(pdist=1.3, psrTerm=True, evolve=True),
        "cgw_phase_approx": dict(pdist=0.9, psrTerm=True, evolve=False, phase_approx=True),
        "cgw_mono": dict(pdist=1.1, psrTerm=True, evolve=False, phase_approx=False),
        "cgw_earth_only": dict(pdist=1.0, psrTerm=False, evolve=True),
        "cgw_pphase": dict(pphase=2.1, psrTerm=True, evolve=True),
        "cgw_mono_earth": dict(pdist=1.0, psrTerm=False, evolve=False, phase_approx=False),
    }
    for tag, kw in cgw_cases.items():
        ps, _ = fresh()
        for p in ps:
            ref.deterministic.add_cgw(p, **base, **kw)
        out[tag] = np.array([sig(p, "cw") for p in ps])
    np.savez_compressed(os.path.join(OUT, "variants.npz"), **out)
    print("variants:", len(out), "arrays")


# ------------------------------------------------------------------------------------------
def gen_c3_mini(ref):
    P, N = 6, 400
    psrs, mjd0 = synth_array(ref, P, N, seed=68, burst=1, backends=("L-wide_PUPPI",))
    out = pulsar_inputs("", psrs, mjd0)
    rng = np.random.default_rng(3)
    rn_A = rng.uniform(-14.5, -13.0, P); rn_g = rng.uniform(1.5, 5.0, P)
    efac = rng.uniform(0.9, 1.2, P); l10eq = rng.uniform(-7.0, -6.0, P); l10ec = rng.uniform(-7.0, -6.0, P)
    out.update(rn_log10_A=rn_A, rn_gamma=rn_g, efac=efac, log10_equad=l10eq, log10_ecorr=l10ec,
               gw_log10_A=np.array(-14.6733), gw_gamma=np.array(13. / 3.))
    seed_efac_equad, seed_jitter, seed_red, seed_gwb = 10660, 17763, 19870, 16672   # notebook cell 8
    # GWB first, as in the reference's own test (tests/test_against_libstempo.py:25): the frequency grid is then
    # built from the ideal TOAs, which is also how the batched engine defines a realisation
    with Capture() as cap:
        ref.red_noise.add_gwb(psrs, log10_amplitude=-14.6733, spectral_index=13. / 3., seed=seed_gwb)
    for ii, psr in enumerate(psrs):
        ref.white_noise.add_measurement_noise(psr, efac=float(efac[ii]), log10_equad=float(l10eq[ii]), seed=seed_efac_equad + ii)
        ref.white_noise.add_jitter(psr, log10_ecorr=float(l10ec[ii]), coarsegrain=0.1, seed=seed_jitter + ii)
        ref.red_noise.add_red_noise(psr, float(rn_A[ii]), float(rn_g[ii]), components=30, seed=seed_red + ii)
    for k in ("measurement_noise", "jitter", "red_noise", "gwb"):
        out[k] = np.array([sig(p, k) for p in psrs])
    out["residuals"] = np.array([p.toas.residuals_s() for p in psrs])
    out["ORF"] = cap.chol_in[0]
    out["Nf"] = np.array(cap.ifft_in[0].shape[1] // 2 + 1)
    np.savez_compressed(os.path.join(OUT, "c3_mini.npz"), **out)
    print("c3_mini: Nf =", int(out["Nf"]))


def gen_cw_catalog(ref):
    """add_catalog_of_cws (deterministic.py:188-561): 40 sources through loop_over_CWs and 1200 through the
    >1000-source branch loop_over_CWs_parallel (numba is stubbed, so both run as plain Python), incl. two binaries that
    have already merged (NaN terms that the reference zeroes)."""
    psrs, mjd0 = synth_array(ref, 3, 200, seed=99, burst=1, backends=("X",))
    out = pulsar_inputs("", psrs, mjd0)
    rng = np.random.default_rng(7)
    for tag, ncw in (("small_", 40), ("large_", 1200)):
        src = dict(gwtheta=np.arccos(rng.uniform(-1, 1, ncw)), gwphi=rng.uniform(0, 2 * np.pi, ncw),
                   mc=10 ** rng.uniform(8.0, 9.5, ncw), dist=10 ** rng.uniform(1.0, 3.0, ncw), fgw=10 ** rng.uniform(-8.8, -7.5, ncw),
                   phase0=rng.uniform(0, 2 * np.pi, ncw), psi=rng.uniform(0, np.pi, ncw), inc=np.arccos(rng.uniform(-1, 1, ncw)))
        src["mc"][3] = 3e10; src["fgw"][3] = 4e-7      # merges within the data span: NaN terms
        src["mc"][11] = 2e10; src["fgw"][11] = 6e-7
        for k, v in src.items():
            out[tag + k] = v
        for case, kw in (("evolve", dict(pdist=1.2, psrTerm=True, evolve=True)), ("mono", dict(pdist=0.8, psrTerm=True, evolve=False)),
                         ("approx", dict(pdist=1.0, psrTerm=True, evolve=False, phase_approx=True)),
                         ("earth", dict(pdist=1.0, psrTerm=False, evolve=True)), ("pphase", dict(pphase=1.7, psrTerm=True, evolve=True))):
            if tag == "large_" and case not in ("evolve", "mono"):
                continue
            ps, _ = synth_array(ref, 3, 200, seed=99, burst=1, backends=("X",))
            with np.errstate(invalid="ignore"):
                for p in ps:
                    ref.deterministic.add_catalog_of_cws(p, src["gwtheta"], src["gwphi"], src["mc"], src["dist"], src["fgw"], src["phase0"],
                                                         src["psi"], src["inc"], tref=53000 * 86400, **kw)
            out[tag + case] = np.array([sig(p, "cw_catalog") for p in ps])
    np.savez_compressed(os.path.join(OUT, "cw_catalog.npz"), **out)
    print("cw_catalog:", len(out), "arrays")


def population_inputs(seed=31):
    """a small binned SMBHB population in holodeck's layout: vals = [Mtot (g), q, z, f_obs (Hz)], weights, bin edges"""
    rng = np.random.default_rng(seed)
    n = 1500
    T_obs = 10 * 365.25 * 86400.0
    fobs = np.arange(1, 8) / T_obs                       # 6 bins
    msol = 1.988409870698051e33
    mtot = 10 ** rng.uniform(8.5, 10.3, n) * msol
    mrat = rng.uniform(0.1, 1.0, n)
    redz = rng.uniform(0.02, 2.0, n)
    fo = 10 ** rng.uniform(np.log10(fobs[0] * 0.8), np.log10(fobs[-1] * 1.1), n)   # a few fall outside the edges
    last = np.nonzero((fo >= fobs[-2]) & (fo < fobs[-1]))[0]
    fo[last[2:]] = fobs[1] * 1.3                         # leave only two binaries in the last bin (< outlier_per_bin)
    weights = 10 ** rng.uniform(-1.0, 3.0, n)
    weights[5] = weights[17] = 50.0                      # an exact tie is broken by position
    mtot[17], mrat[17], redz[17], fo[17] = mtot[5], mrat[5], redz[5], fo[5]
    return np.array([mtot, mrat, redz, fo]), weights, fobs, T_obs


def gen_population(ref):
    """add_gwb_plus_outlier_cws (deterministic.py:565-715) on 3 x 200 TOAs: loudest 4 binaries per bin as CWs, the rest as a
    user-spectrum GWB; holodeck is the formula stub in oracle/_stubs/holodeck (parity with holodeck itself unpinned)."""
    psrs, mjd0 = synth_array(ref, 3, 200, seed=123, burst=1, backends=("X",))
    out = pulsar_inputs("", psrs, mjd0)
    vals, weights, fobs, T_obs = population_inputs()
    out.update(vals=vals, weights=weights, fobs=fobs, T_obs=np.array(T_obs))
    with np.errstate(invalid="ignore", divide="ignore"):
        ret = ref.deterministic.add_gwb_plus_outlier_cws(psrs, vals, weights, fobs, T_obs, outlier_per_bin=4, seed=4711)
    names = ("f_centers", "free_spec", "outlier_fo", "outlier_hs", "outlier_mc", "outlier_dl", "gwthetas", "gwphis", "phases",
             "psis", "incs")
    for k, v in zip(names, ret):
        out["ret_" + k] = np.asarray(v)
    out["gwb_days"] = np.array([sig(p, "gwb") for p in psrs])
    out["cw_catalog"] = np.array([sig(p, "cw_catalog") for p in psrs])
    out["shift_days"] = np.array([np.sum(np.array(p.toas.shifts_day, dtype=np.longdouble), axis=0).astype(np.float64) for p in psrs])
    np.savez_compressed(os.path.join(OUT, "population.npz"), **out)
    print("population:", len(ret[2]), "outliers; free_spec", ret[1])


BURST = dict(t0=2.2e8, tau=4.0e7, f=3.0e-8, a_plus=2.0e-7, a_cross=1.3e-7)


def burst_plus(t, b=BURST):
    return b["a_plus"] * np.exp(-0.5 * ((t - b["t0"]) / b["tau"]) ** 2) * np.cos(2 * np.pi * b["f"] * (t - b["t0"]))


def burst_cross(t, b=BURST):
    return b["a_cross"] * np.exp(-0.5 * ((t - b["t0"]) / b["tau"]) ** 2) * np.sin(2 * np.pi * b["f"] * (t - b["t0"]))


def gen_transients(ref):
    """add_burst (with and without the quadratic fit), add_noise_transient, add_gw_memory (deterministic.py:718-884)."""
    psrs, mjd0 = synth_array(ref, 3, 150, seed=321, burst=1, backends=("X",))
    out = pulsar_inputs("", psrs, mjd0)
    tref = 53000 * 86400
    for case, rq in (("burst", False), ("burst_quad", True)):
        ps, _ = synth_array(ref, 3, 150, seed=321, burst=1, backends=("X",))
        for p in ps:
            ref.deterministic.add_burst(p, 1.1, 4.0, burst_plus, burst_cross, psi=0.4, tref=tref, remove_quad=rq)
        out[case] = np.array([sig(p, "burst") for p in ps])
    ps, _ = synth_array(ref, 3, 150, seed=321, burst=1, backends=("X",))
    for p in ps:
        ref.deterministic.add_noise_transient(p, burst_plus, tref=tref)
        ref.deterministic.add_gw_memory(p, 3e-14, 0.7, 5.1, 0.9, 55500.0)
    out["noise_transient"] = np.array([sig(p, "noise_transient") for p in ps])
    out["gw_memory"] = np.array([sig(p, "gw_memory") for p in ps])
    np.savez_compressed(os.path.join(OUT, "transients.npz"), **out)
    print("transients:", {k: float(np.sqrt(np.mean(out[k] ** 2))) for k in ("burst", "burst_quad", "noise_transient", "gw_memory")})


def gen_rn_modes(ref):
    """add_red_noise with an explicit `modes` frequency list (red_noise.py:64-66,119-121; `components` is then ignored), both
    phase conventions."""
    psrs, mjd0 = synth_array(ref, 2, 180, seed=555, burst=1, backends=("X",))
    out = pulsar_inputs("", psrs, mjd0)
    modes = np.array([1.0, 2.0, 3.0, 5.0, 8.0, 13.0]) / (5478 * 86400.0)
    out["modes"] = modes
    for tag, conv in (("default", False), ("libstempo", True)):
        ps, _ = synth_array(ref, 2, 180, seed=555, burst=1, backends=("X",))
        for a, p in enumerate(ps):
            ref.red_noise.add_red_noise(p, -13.3, 3.7, components=30, seed=4242 + a, modes=modes, libstempo_convention=conv)
        out["rn_" + tag] = np.array([sig(p, "red_noise") for p in ps])
    np.savez_compressed(os.path.join(OUT, "rn_modes.npz"), **out)
    print("rn_modes: rms", float(np.sqrt(np.mean(out["rn_default"] ** 2))))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    ref = rr.load_reference()
    gen_c1(ref)
    gen_c2(ref)
    gen_orf(ref)
    gen_variants(ref)
    gen_c3_mini(ref)
    gen_cw_catalog(ref)
    gen_population(ref)
    gen_transients(ref)
    gen_rn_modes(ref)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)) // 1024, "KiB")
