"""CPU baseline of bench.py (TEST / MEASUREMENT INFRASTRUCTURE - never imported by the product).

    python oracle/cpu_baseline.py WORKLOAD.npz [--subset S] [--repeats K]

Times one realisation of the bench workload (68 pulsars x 5000 TOAs: HD GWB + per-pulsar RN + per-backend EFAC / t2EQUAD +
per-backend ECORR) on the host cores, the way the reference spends its time - everything rebuilt per call:

* kind "reference": when /root/reference is mounted (the build container), the UNMODIFIED reference functions
  add_gwb / add_measurement_noise / add_jitter / add_red_noise (red_noise.py:106-298, white_noise.py:47-198) run under the
  dependency stubs of oracle/run_reference.py (numeric core only: the PINT sink is a no-op);
* kind "port": elsewhere (the GPU box), the NumPy restatement oracle/pta_oracle.py with the reference's dense-U ECORR
  (white_noise.py:37-39,182) spelled out.

Bounded sample (BASELINE.md §3 asks for >= 3 repeats after a warm-up): add_gwb is timed on the WHOLE array; the per-pulsar
calls on the first S pulsars and scaled by P / S.  BLAS threads come from the environment (bench.py runs this file twice,
OPENBLAS/OMP/MKL_NUM_THREADS = 1 and = all cores).  Prints one JSON object.
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def load(path):
    z = np.load(path, allow_pickle=False)
    noise = json.loads(str(z["noise_json"]))
    return z, noise


def run_reference(z, noise, S, repeats):
    from oracle import run_reference as rr
    ref = rr.load_reference()
    P = z["mjd"].shape[0]
    names = [str(x) for x in z["names"]]

    def build():
        psrs = []
        for a in range(P):
            be = noise["flags"][a]
            flags = [{"f": be[k]} for k in z["which"][a]]
            psrs.append(rr.make_pulsar(ref, names[a], z["mjd"][a], 0.5, {"RAJ": float(z["raj"][a]), "DECJ": float(z["decj"][a])}, flags=flags))
        return psrs

    def once():
        psrs = build()
        t = {}
        t0 = time.perf_counter()
        ref.red_noise.add_gwb(psrs, noise["gw_log10_A"], 13. / 3., seed=16672)
        t["gwb"] = time.perf_counter() - t0
        t["wn"] = t["ecorr"] = t["rn"] = 0.0
        for a in range(S):
            p = psrs[a]
            t0 = time.perf_counter()
            ref.white_noise.add_measurement_noise(p, efac=np.array(noise["efac"][a]), log10_equad=np.array(noise["log10_equad"][a]), flagid="f",
                                                  flags=noise["flags"][a], seed=10660 + a)
            t1 = time.perf_counter()
            ref.white_noise.add_jitter(p, log10_ecorr=np.array(noise["log10_ecorr"][a]), flagid="f", flags=noise["flags"][a], coarsegrain=0.1,
                                       seed=17763 + a)
            t2 = time.perf_counter()
            if noise["rn_log10_A"][a] is not None:
                ref.red_noise.add_red_noise(p, noise["rn_log10_A"][a], noise["rn_gamma"][a], components=30, seed=19870 + a)
            t3 = time.perf_counter()
            t["wn"] += t1 - t0
            t["ecorr"] += t2 - t1
            t["rn"] += t3 - t2
        return t

    once()                                    # warm-up
    return [once() for _ in range(repeats)]


def run_port(z, noise, S, repeats):
    from oracle import pta_oracle as po
    P = z["mjd"].shape[0]
    mjd = [z["mjd"][a] for a in range(P)]
    locs = po.psr_locs_equatorial([{"RAJ": float(z["raj"][a]), "DECJ": float(z["decj"][a])} for a in range(P)])

    def once():
        t = {}
        t0 = time.perf_counter()
        grid = po.gwb_grid([float(m.min()) for m in mjd], [float(m.max()) for m in mjd])
        ORF = po.gwb_orf(locs)                      # pair loop in Python, like spharmORFbasis.correlated_basis
        M = np.linalg.cholesky(ORF)
        w = po.gwb_draws(16672, P, grid["Nf"])
        C = po.gwb_spectrum(grid["f"], grid["dur"], grid["howml"], noise["gw_log10_A"], 13. / 3.)
        po.gwb_dt(grid, M, w, C, [m * 86400 for m in mjd])
        t["gwb"] = time.perf_counter() - t0
        t["wn"] = t["ecorr"] = t["rn"] = 0.0
        for a in range(S):
            n = len(mjd[a])
            tf = np.array([noise["flags"][a][k] for k in z["which"][a]])
            t0 = time.perf_counter()
            z1, z2 = po.legacy_normals(10660 + a, [n, n])
            efv = po.flag_vector(tf, noise["flags"][a], noise["efac"][a], n)
            eqv = po.flag_vector(tf, noise["flags"][a], 10 ** np.asarray(noise["log10_equad"][a]), n)
            po.measurement_noise_dt(np.full(n, 0.5e-6), efv, eqv, z1, z2)
            t1 = time.perf_counter()
            epoch_of, ne, first, _ = po.quantize(mjd[a], dt=0.1)
            (ze,) = po.legacy_normals(17763 + a, [ne])
            ecv = po.jitter_ecorr_vector(ne, first, noise["log10_ecorr"][a], toa_flags=tf, flags=noise["flags"][a])
            U = np.zeros((n, ne), "d")              # white_noise.py:37-39,182: dense N x E indicator matrix and matvec
            U[np.arange(n), epoch_of] = 1
            np.dot(U * ecv, ze)
            t2 = time.perf_counter()
            if noise["rn_log10_A"][a] is not None:
                (zr,) = po.legacy_normals(19870 + a, [60])
                po.red_noise_dt(mjd[a], noise["rn_log10_A"][a], noise["rn_gamma"][a], zr)
            t3 = time.perf_counter()
            t["wn"] += t1 - t0
            t["ecorr"] += t2 - t1
            t["rn"] += t3 - t2
        return t

    once()
    return [once() for _ in range(repeats)]


def main():
    path = sys.argv[1]
    S = int(sys.argv[sys.argv.index("--subset") + 1]) if "--subset" in sys.argv else 8
    repeats = int(sys.argv[sys.argv.index("--repeats") + 1]) if "--repeats" in sys.argv else 3
    z, noise = load(path)
    P, N = z["mjd"].shape
    S = min(S, P)
    kind = "reference" if os.path.isdir("/root/reference") and "--port" not in sys.argv else "port"
    runs = (run_reference if kind == "reference" else run_port)(z, noise, S, repeats)
    scale = P / S
    totals = [r["gwb"] + scale * (r["wn"] + r["ecorr"] + r["rn"]) for r in runs]
    no_ecorr = [r["gwb"] + scale * (r["wn"] + r["rn"]) for r in runs]
    med = float(np.median(totals))
    print(json.dumps({
        "kind": kind, "P": int(P), "N": int(N), "subset": int(S), "repeats": int(repeats),
        "threads_env": os.environ.get("OPENBLAS_NUM_THREADS") or os.environ.get("OMP_NUM_THREADS"),
        "seconds_per_realisation": med, "seconds_runs": [round(t, 4) for t in totals],
        "seconds_without_ecorr": float(np.median(no_ecorr)),
        "seconds_parts_last_run": {k: round(v * (1 if k == "gwb" else scale), 4) for k, v in runs[-1].items()},
    }))


if __name__ == "__main__":
    main()
