#!/usr/bin/env python3
"""Headline benchmark: realisations/sec, 68 pulsars x 5000 TOAs, GWB + RN + WN (EFAC/EQUAD + ECORR), fp64.

    python bench.py --gpus N --steps K --warmup W [--batch R]

Workload = BASELINE.json config 3 as SURVEY.md §8d specifies it (headline_array): the 68 pulsars of ng15_dict.json with their
per-backend EFAC / t2EQUAD / ECORR and red-noise values, HD GWB at the dictionary's gw_log10_A, 1024 realisations per step.
A "step" is one pass of the hot path over one batch of R realisations of the whole array, every Gaussian deviate drawn on
chip, inputs resident in HBM (ReplicaEngine.generate = one pta_engine_generate call: pta_engine_rn_coef -> pta_gwb_czt ->
pta_gwb_mix -> pta_engine_synth).  N > 1: launched by torch.distributed.run, one rank per GPU; realisations are independent,
so rank g generates realisations [g*R*(K+W) .. ) of the same seeded stream (weak scaling, no data-path collective).  The
north_star's gather of the residual arrays to rank 0 is timed separately as a pipelined generate+gather
(`gathered_to_rank0`), not folded into the step.  Rank 0 prints ONE JSON line.

Besides the contract fields the line carries
  roofline      the dominant kernel of the step against its bound: algorithmic units per launch / launch time measured
                here with HIP events on the launch stream; HBM peak 8 TB/s (MI355X_MICROARCH.md); fp64 matrix peak
                78.6 TFLOP/s = AMD's public MI355X figure, cross-checked by the in-library microbenchmark.  `traffic`
                (PMC HBM bytes per launch) comes from the committed rocprofv3 passes in profiles/r03_pmc.json and is
                quoted only while launch shape and kernel sources match that profile - otherwise null with the reason
  kernels_ms    per-kernel times of one step (HIP events)
  td_mode       the dense path of the north_star on the same array: covariance assembly, batched fp64 Cholesky
                (TFLOP/s, MFMA-busy %), whole-array realisations/s of generate_td
  cpu_baseline  oracle/cpu_baseline.py on the host cores (subprocess; BLAS threads 1 and all): the unmodified reference
                under stubs where /root/reference is mounted (kind "reference"), else the NumPy port (kind "port")
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
FP64_MFMA_PEAK_TFLOPS = 78.6   # AMD public MI355X fp64 matrix (= vector) figure (not in the local guide)


def ng15_noise():
    """per-pulsar, per-backend noise values of the reference's noise_dicts/ng15_dict.json (fixture written by
    oracle/gen_ng15_fixture.py; the dictionary itself does not travel to the GPU box)."""
    with open(os.path.join(ROOT, "tests", "golden", "ng15_noise.json")) as fh:
        return json.load(fh)


def headline_array(P=68, N=5000, seed=68):
    """BASELINE.json config 3 as SURVEY.md §8d specifies it: the 68 pulsars of ng15_dict.json (names, per-backend
    EFAC / t2EQUAD / ECORR, red noise of the 67 pulsars that have it, gw_log10_A), on synthetic inputs where the reference
    ships none - isotropic sky (RAJ ~ U(0, 24) h, sin DEC ~ U(-1, 1), default_rng(68)), N sorted TOAs ~ U(53000, 58478) MJD,
    0.5 us errors, every TOA tagged with one of its pulsar's backends (flag "f").  P != 68 cycles through the dictionary."""
    from pta_replicator_amd.simulate import ArrayTOAs, SimulatedPulsar, make_ideal
    nd = ng15_noise()
    names = list(nd["pulsars"])
    rng = np.random.default_rng(seed)
    raj = rng.uniform(0, 24, P)
    decj = np.degrees(np.arcsin(rng.uniform(-1, 1, P)))
    psrs = []
    noise = dict(flags=[], efac=[], log10_equad=[], log10_ecorr=[], rn_log10_A=[], rn_gamma=[], gw_log10_A=float(nd["gw_log10_A"]))
    for a in range(P):
        name = names[a % len(names)]
        rec = nd["pulsars"][name]
        mjd = np.sort(rng.uniform(53000, 58478, N))
        be = rec["backends"]
        which = rng.integers(0, len(be), N)
        psr = SimulatedPulsar(toas=ArrayTOAs(mjd, 0.5, flags=[{"f": be[k]} for k in which]),
                              name=name if a < len(names) else f"{name}_{a // len(names)}", loc={"RAJ": float(raj[a]), "DECJ": float(decj[a])})
        make_ideal(psr)
        psrs.append(psr)
        noise["flags"].append(list(be))
        noise["efac"].append(np.array([1.0 if v is None else v for v in rec["efac"]]))   # one backend has no EFAC entry: the default
        noise["log10_equad"].append(np.array(rec["log10_t2equad"]))
        noise["log10_ecorr"].append(np.array(rec["log10_ecorr"]))
        noise["rn_log10_A"].append(rec["red_noise_log10_A"])                              # None for J0614-3329
        noise["rn_gamma"].append(rec["red_noise_gamma"])
    return psrs, noise


def configure_engine(eng, noise):
    """GWB (HD, gamma = 13/3) + per-pulsar power-law RN (30 components) + per-backend EFAC / t2EQUAD / ECORR (0.1 d epochs)."""
    eng.set_white_noise(efac=noise["efac"], log10_equad=noise["log10_equad"], flags=noise["flags"])
    eng.set_jitter(log10_ecorr=noise["log10_ecorr"], flags=noise["flags"], coarsegrain=0.1)
    eng.set_red_noise(noise["rn_log10_A"], noise["rn_gamma"], components=30)
    eng.set_gwb(noise["gw_log10_A"], 13. / 3.)
    return eng


def build_engine(P, N, seed):
    from pta_replicator_amd.engine import ReplicaEngine
    psrs, noise = headline_array(P, N)
    eng = configure_engine(ReplicaEngine(psrs, seed=seed), noise)
    eng.prepare()
    return eng, psrs, noise


def src_sha(*files):
    """sha256 over the CODE of the kernel sources a profile was taken with (comments and white space removed: a reworded comment does
    not invalidate a measurement): a PMC figure in profiles/*.json is only quoted while the sources that produced it are unchanged."""
    import hashlib
    import re
    h = hashlib.sha256()
    for f in files:
        with open(os.path.join(ROOT, "pta_replicator_amd", "csrc", f), "r") as fh:
            code = re.sub(r"/\*.*?\*/", " ", fh.read(), flags=re.S)
            code = re.sub(r"//[^\n]*", " ", code)
            h.update(" ".join(code.split()).encode())
    return h.hexdigest()[:16]


PMC_FILE = os.path.join(ROOT, "profiles", "r05_pmc.json")
SYNTH_SRC = ("pta_engine_kernels.hip", "pta_rng.h", "pta_rng_tables.h", "pta_mfma.h")
TD_SRC = ("pta_td_kernels.hip", "pta_gemm.hip", "pta_orf_kernels.hip", "pta_rng.h", "pta_rng_tables.h", "pta_mfma.h")
CZT_SRC = ("pta_czt_kernels.hip", "pta_fft.h", "pta_rng.h", "pta_rng_tables.h")


def pmc_entry(key, srcs, **shape):
    """counters of kernel `key` from the committed rocprofv3 --pmc passes (scripts/gpu_profile_r5.sh -> profiles/r05_pmc.json),
    or (None, reason) when the file is missing, was taken at another launch shape, or the kernel sources changed since."""
    try:
        with open(PMC_FILE) as fh:
            e = json.load(fh)[key]
    except (OSError, KeyError, ValueError):
        return None, "no committed PMC profile for this kernel"
    if any(e.get(k) != v for k, v in shape.items()):
        return None, f"PMC profile was taken at another launch shape ({ {k: e.get(k) for k in shape} })"
    if e.get("src_sha") != src_sha(*srcs):
        return None, "kernel sources changed since the PMC profile was taken (stale)"
    return e, None


def dump_workload(psrs, noise, path):
    """the bench workload as plain arrays for oracle/cpu_baseline.py (a separate process: its BLAS thread count is set by
    the environment, and the reference's dependency stubs never enter this process)."""
    names = [p.name for p in psrs]
    mjd = np.stack([np.asarray(p.toas.get_mjds().value, dtype=np.float64) for p in psrs])
    which = np.stack([np.array([noise["flags"][a].index(f["f"]) for f in p.toas.table["flags"].data], dtype=np.int32) for a, p in enumerate(psrs)])
    nj = {k: ([None if x is None else (x.tolist() if hasattr(x, "tolist") else x) for x in v] if isinstance(v, list) else v) for k, v in noise.items()}
    np.savez(path, names=np.array(names), mjd=mjd, which=which, raj=np.array([p.loc["RAJ"] for p in psrs]),
             decj=np.array([p.loc["DECJ"] for p in psrs]), noise_json=np.array(json.dumps(nj)))


def cpu_baseline(psrs, noise, subset=8, repeats=3):
    """oracle/cpu_baseline.py twice - BLAS threads = 1 and = all host cores - on a bounded sample of the same workload
    (whole-array add_gwb + the per-pulsar calls of `subset` pulsars scaled to the array; one warm-up + `repeats` timed
    repeats each).  kind = "reference" (the unmodified reference under stubs) where /root/reference is mounted, else "port"."""
    import subprocess
    import tempfile
    ncpu = os.cpu_count() or 1
    res = {}
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "workload.npz")
        dump_workload(psrs, noise, path)
        for label, nt in (("single_thread", 1), ("all_cores", ncpu)):
            env = dict(os.environ, OPENBLAS_NUM_THREADS=str(nt), OMP_NUM_THREADS=str(nt), MKL_NUM_THREADS=str(nt))
            out = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "cpu_baseline.py"), path, "--subset", str(subset), "--repeats",
                                  str(repeats)], env=env, capture_output=True, text=True, timeout=600)
            if out.returncode != 0:
                raise RuntimeError(f"oracle/cpu_baseline.py failed: {out.stderr[-400:]}")
            res[label] = json.loads(out.stdout.strip().splitlines()[-1])
            res[label]["threads"] = nt
    best = min(res.values(), key=lambda r: r["seconds_per_realisation"])
    P, N = len(psrs), len(psrs[0].toas.get_mjds().value)
    return {"value": 1.0 / best["seconds_per_realisation"], "unit": "realisations/s", "cores": best["threads"], "kind": best["kind"],
            "sample": f"{best['repeats']} timed repeats after one warm-up of: add_gwb over all {P} pulsars x {N} TOAs + add_measurement_noise / add_jitter "
                      f"(reference-style dense-U ECORR) / add_red_noise on {best['subset']} pulsars, scaled x{P}/{best['subset']}; median; "
                      f"{'unmodified reference functions under the stubs of oracle/run_reference.py (numeric core, PINT sink excluded)' if best['kind'] == 'reference' else 'NumPy port oracle/pta_oracle.py (/root/reference is not mounted on this host)'}",
            "single_thread": {k: res["single_thread"][k] for k in ("seconds_per_realisation", "seconds_runs", "seconds_without_ecorr", "seconds_parts_last_run", "threads")},
            "all_cores": {k: res["all_cores"][k] for k in ("seconds_per_realisation", "seconds_runs", "seconds_without_ecorr", "seconds_parts_last_run", "threads")},
            "value_without_ecorr": 1.0 / best["seconds_without_ecorr"], "host_cpus": ncpu}


def api_mode_timing(psrs, noise, repeats=2):
    """ONE realisation of the bench array through the drop-in add_* API (replay mode: NumPy legacy draws on the host in the
    reference's order, host-owned pulsar objects, PCIe both ways), in ms: `loop` = the reference's usage, one call per pulsar and
    signal (tests/test_against_libstempo.py:25-53, notebook cell 9); `list` = the same calls given the pulsar list (one launch per
    signal, the pulsars' legacy streams drawn by the native restatement of NumPy's generator on host threads - pta_legacy_randn).
    `host_rng_ms` = what np.random alone costs for these draws on this host, single thread - the floor of the loop form."""
    import torch
    from pta_replicator_amd.simulate import make_ideal
    from pta_replicator_amd.white_noise import add_measurement_noise, add_jitter
    from pta_replicator_amd.red_noise import add_red_noise, add_gwb
    P = len(psrs)
    s_wn, s_ec, s_rn = [10660 + i for i in range(P)], [17763 + i for i in range(P)], [19870 + i for i in range(P)]

    def run(style):
        for p in psrs:
            make_ideal(p)
        t = {}
        torch.cuda.synchronize(); t0 = time.perf_counter()
        add_gwb(psrs, noise["gw_log10_A"], 13. / 3., seed=16672)
        torch.cuda.synchronize(); t["gwb"] = time.perf_counter() - t0; t0 = time.perf_counter()
        if style == "loop":
            for ii, p in enumerate(psrs):
                add_measurement_noise(p, efac=noise["efac"][ii], log10_equad=noise["log10_equad"][ii], flags=noise["flags"][ii], seed=s_wn[ii])
        else:
            add_measurement_noise(psrs, efac=noise["efac"], log10_equad=noise["log10_equad"], flags=noise["flags"], seed=s_wn)
        torch.cuda.synchronize(); t["wn"] = time.perf_counter() - t0; t0 = time.perf_counter()
        if style == "loop":
            for ii, p in enumerate(psrs):
                add_jitter(p, log10_ecorr=noise["log10_ecorr"][ii], flags=noise["flags"][ii], coarsegrain=0.1, seed=s_ec[ii])
        else:
            add_jitter(psrs, log10_ecorr=noise["log10_ecorr"], flags=noise["flags"], coarsegrain=0.1, seed=s_ec)
        torch.cuda.synchronize(); t["ecorr"] = time.perf_counter() - t0; t0 = time.perf_counter()
        if style == "loop":
            for ii, p in enumerate(psrs):
                if noise["rn_log10_A"][ii] is not None:
                    add_red_noise(p, noise["rn_log10_A"][ii], noise["rn_gamma"][ii], components=30, seed=s_rn[ii])
        else:
            add_red_noise(psrs, noise["rn_log10_A"], noise["rn_gamma"], components=30, seed=s_rn)
        torch.cuda.synchronize(); t["rn"] = time.perf_counter() - t0
        t["total"] = sum(t.values())
        return t, np.concatenate([p.residuals.resids_value for p in psrs])

    out = {}
    res = {}
    for style in ("loop", "list"):
        run(style)                                       # warm-up (flag index caches, pinned buffers)
        best = None
        for _ in range(repeats):
            t, r = run(style)
            if best is None or t["total"] < best["total"]:
                best = t
        out[style] = {k: round(v * 1e3, 3) for k, v in best.items()}
        res[style] = r
    out["list_equals_loop"] = bool(np.array_equal(res["loop"], res["list"]))
    t0 = time.perf_counter()
    np.random.seed(1)
    Nf = 3000
    for p in psrs:
        n = p.toas.ntoas
        np.random.randn(Nf); np.random.randn(Nf); np.random.randn(n); np.random.randn(n); np.random.randn(n); np.random.randn(60)
    out["host_rng_ms"] = round((time.perf_counter() - t0) * 1e3, 3)
    for p in psrs:
        make_ideal(p)
    return out


def td_mode_numbers(eng, R):
    """BASELINE.json's secondary metric on the SAME array: the dense time-domain path (no counterpart in the reference) -
    covariance assembly, batched blocked fp64 Cholesky (MFMA trailing update), then whole-array realisations/s of generate_td
    (L.z with in-register deviates + GWB grid factor + interpolation)."""
    import ctypes
    import torch
    from pta_replicator_amd import _lib, device as dv

    def wall(fn, reps=1):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps

    # what a first prepare_td() costs beyond its kernels is the driver allocating the factor buffer (13.6 GB at 68 x 5000: hipMalloc of
    # fresh memory, ~0.2-0.4 s, paid once per process - the caching allocator hands the block back on later calls): timed on its own,
    # then prepare_td() twice - the first call still creates the internal streams / events and the GWB grid factor
    counts = [int(c) for c in eng.counts]
    nbytes = 8 * sum(n * (n + (n & 1)) for n in counts)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    blk = torch.empty((nbytes,), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize(); t_alloc = time.perf_counter() - t0
    del blk
    t_first = wall(eng.prepare_td)
    t_warm = wall(eng.prepare_td)
    s = dv.stream_ptr()
    phi = (eng.d_amp ** 2).contiguous()
    ec2 = (eng.d_ecorr_toa ** 2).contiguous()

    assemble = eng.td_assemble     # the engine's default assembly kernel (column-walking where 1 <= K <= 64, else 64 x 128 tiles)

    uniform = len(set(counts)) == 1
    res = {"n_psr": eng.P, "n_toa": counts[0] if uniform else counts, "prepare_td_ms": t_warm * 1e3, "prepare_td_first_call_ms": t_first * 1e3,
           "factor_buffer_alloc_ms": t_alloc * 1e3, "factor_buffer_GB": nbytes / 1e9}
    flop_chol = sum(n ** 3 for n in counts) / 3.0
    # the assembly alone, both kernels, against the ALGORITHMIC bytes (8 per element of the lower triangles, diagonal included)
    cov_bytes = 8.0 * sum(n * (n + 1) / 2 for n in counts)
    res["cov_assemble_kernel"] = getattr(eng, "td_cov_kernel_used", None)       # what prepare_td() took
    for kname, kern_, var in (("walk", "walk", 0), ("tile", "tile", 0)):
        try:
            eng.td_cov_walk_variant = var
            eng.td_assemble(kernel=kern_)
            tk = min(wall(lambda: eng.td_assemble(kernel=kern_)) for _ in range(4))
            res[f"cov_assemble_{kname}_ms"] = tk * 1e3
            res[f"cov_assemble_{kname}_TBps"] = cov_bytes / tk / 1e12
        except Exception as e:  # pragma: no cover
            res[f"cov_assemble_{kname}_error"] = str(e)[:200]
    eng.td_cov_walk_variant = 0
    if uniform:
        n, ld, P = eng.td_nst[0], eng.td_ld[0], eng.P   # stored order: an odd TOA count carries one identity row / column (engine_td.prepare_td)
        info = dv.zeros((P,), dtype=torch.int32)
        # the schedule prepare_td() uses (workspace scheme, next panel's diagonal phase run ahead) and the workspace-free two-chain one
        need = int(_lib.lib.pta_potrf_workspace_doubles(n, P, _lib.POTRF_DIAG_AHEAD))
        work = dv.empty((need,))
        ts, ts_free, bad = [], [], 0
        for _ in range(2):
            wall(assemble)
            ts_free.append(wall(lambda: _lib.call("pta_potrf_batched_ex", dv.ptr(eng.d_Ltd), n, ld, n * ld, P, dv.ptr(info), 0, s)))
            bad += int(info.abs().sum().item())
        ts_epi = []
        for _ in range(4):
            ta = wall(assemble)
            ts.append(wall(lambda: _lib.call("pta_potrf_batched_ws", dv.ptr(eng.d_Ltd), n, ld, n * ld, P, dv.ptr(info), _lib.POTRF_DIAG_AHEAD,
                                             dv.ptr(work), need, s)))
            bad += int(info.abs().sum().item())
            wall(assemble)   # A/B: the tile products' C-tile prefetch epilogue (PTA_POTRF_EPI1)
            ts_epi.append(wall(lambda: _lib.call("pta_potrf_batched_ws", dv.ptr(eng.d_Ltd), n, ld, n * ld, P, dv.ptr(info), _lib.POTRF_DIAG_AHEAD | _lib.POTRF_EPI1,
                                                 dv.ptr(work), need, s)))
            bad += int(info.abs().sum().item())
        res["potrf_epi1_ms"] = min(ts_epi) * 1e3
        info.add_(bad)
        del work
        tp = min(ts)
        res.update({"potrf_workspace_GB": 8.0 * need / 1e9, "potrf_without_workspace_ms": min(ts_free) * 1e3})
        # the same batch through the END-ALIGNED ragged schedule (pta_potrf_ragged; a uniform batch is its special case front = const)
        try:
            tr = []
            for _ in range(3):
                wall(assemble)
                tr.append(wall(lambda: eng.td_factorise(mode="ragged")))
            res["potrf_ragged_schedule_ms"] = min(tr) * 1e3
            res["potrf_ragged_schedule_TFLOPs"] = flop_chol / min(tr) / 1e12
            wall(assemble)
            eng.td_factorise(mode="uniform")
        except Exception as e:  # pragma: no cover
            res["potrf_ragged_schedule_error"] = str(e)[:200]

        def factor_loop():
            assemble()
            _lib.call("pta_potrf_batched_ws", dv.ptr(eng.d_Ltd), n, ld, n * ld, P, dv.ptr(info), _lib.POTRF_DIAG_AHEAD, dv.ptr(work2), need, s)
        work2 = dv.empty((need,))
        ck = engine_clock_during(factor_loop, 0.6)
        del work2
        if ck:
            res["potrf_engine_clock_GHz"] = ck["GHz"]
        eng.prepare_td()
        res.update({"cov_assemble_ms": ta * 1e3, "cov_assemble_GBps_lower_triangle": cov_bytes / ta / 1e9,
                    "potrf_ms": tp * 1e3, "potrf_TFLOPs": flop_chol / tp / 1e12, "potrf_frac_of_fp64_mfma_peak": flop_chol / tp / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                    "positive_definite": int(info.abs().sum().item()) == 0})
    out = dv.empty((R, eng.n_toa))
    # the same realisations two ways: deviates generated inside the product's loop ("registers", no buffer) and written once per batch
    # and read by the product ("memory", the default)
    eng.td_draws = "registers"
    eng.generate_td(R, out=out)
    t_reg = wall(lambda: eng.generate_td(R, out=out), 2)
    eng.td_draws = "memory"
    eng.td_overlap = True                       # A/B (opt-in): chunks of 256, chunk c + 1 prepared on a side stream beside the product of chunk c
    eng.generate_td(R, out=out)
    t_overlap = wall(lambda: eng.generate_td(R, out=out), 2)
    eng.td_overlap = False                      # default: one chunk, deviates and GWB grid series in front of the product
    eng.generate_td(R, out=out)
    t = wall(lambda: eng.generate_td(R, out=out), 3)
    flop = float(sum(n * n for n in counts))       # useful flops per realisation of L.z (triangular): sum N_a^2
    ck = engine_clock_during(lambda: eng.generate_td(R, out=out), 0.5)
    if ck:
        res["trmm_engine_clock_GHz"] = ck["GHz"]
        res["trmm_frac_at_measured_clock"] = flop * R / t / 1e12 / (FP64_MFMA_PEAK_TFLOPS * ck["GHz"] / 2.4)
        if res.get("potrf_engine_clock_GHz") and res.get("potrf_TFLOPs"):
            res["potrf_frac_at_measured_clock"] = res["potrf_TFLOPs"] / (FP64_MFMA_PEAK_TFLOPS * res["potrf_engine_clock_GHz"] / 2.4)
    res.update({"generate_td_realisations": R, "generate_td_ms": t * 1e3, "realisations_per_s": R / t,
                "trmm_useful_TFLOPs": flop * R / t / 1e12, "trmm_frac_of_fp64_mfma_peak": flop * R / t / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                "td_draws": "memory (deviates written once per batch, read by the product)",
                "generate_td_ms_with_chunk_overlap_opt_in": t_overlap * 1e3,
                "draws_in_registers": {"generate_td_ms": t_reg * 1e3, "realisations_per_s": R / t_reg, "trmm_useful_TFLOPs": flop * R / t_reg / 1e12},
                "gw_grid_factor_jitter": eng.gw_td_jitter if eng.plan.gw_npts else None})
    # MFMA-busy % from the committed PMC pass: the tile product over its dispatches of >= 1 ms (the trailing updates; the mean over all
    # of its launches, small ones included, is carried as ..._all_dispatches), the L.z product for the default (memory) form
    for key, name in (("k_dgemm_glds128", "potrf_trailing_update_mfma_busy_pct"), ("k_td_trmm_rng<false, true>", "trmm_mfma_busy_pct"),
                      ("k_td_cov_walk", "cov_assemble_mfma_busy_pct")):
        e, why = pmc_entry(key, TD_SRC, n_psr=eng.P)
        big = (e or {}).get("dispatches_over_1ms")
        res[name] = (big or e)["mfma_busy_pct"] if e else None
        if e:
            if big:
                res[name + "_all_dispatches"] = e["mfma_busy_pct"]
                res[name.replace("mfma_busy_pct", "gui_active_cycles_per_xcd_per_ns")] = big.get("gui_active_cycles_per_xcd_per_ns")   # NOT a clock (launch gaps)
            res[name + "_source"] = e.get("source")
            if key == "k_td_cov_walk" and e.get("hbm_write_GBps"):   # counter bytes (WRITE_SIZE) over the rocprofv3 launch time
                res["cov_assemble_GBps_from_WRITE_SIZE"] = e["hbm_write_GBps"]
                res["cov_assemble_write_bytes_pmc"] = e["write_kib_per_dispatch"] * 1024.0
        else:
            res[name + "_note"] = why
    return res


def ragged_counts(P=42, lo=500, hi=35000):
    """an ng15-like spread of TOA counts: P quantiles of the log-uniform distribution on [lo, hi] (P = 42: sum N_a = 340 915, the
    headline array's total; 48 GB of factors, 46.9 TFLOP of factorisation), shuffled so that the array order is not the size order."""
    n = np.round(lo * (hi / lo) ** ((np.arange(P) + 0.5) / P)).astype(int)
    return [int(x) for x in np.random.default_rng(P).permutation(n)]


def ragged_array(counts, seed=42):
    """pulsars with the given TOA counts over the headline span, ng15 noise values cycled (as headline_array)."""
    from pta_replicator_amd.simulate import ArrayTOAs, SimulatedPulsar, make_ideal
    nd = ng15_noise()
    names = list(nd["pulsars"])
    rng = np.random.default_rng(seed)
    P = len(counts)
    raj = rng.uniform(0, 24, P)
    decj = np.degrees(np.arcsin(rng.uniform(-1, 1, P)))
    psrs = []
    noise = dict(flags=[], efac=[], log10_equad=[], log10_ecorr=[], rn_log10_A=[], rn_gamma=[], gw_log10_A=float(nd["gw_log10_A"]))
    for a, N in enumerate(counts):
        name = names[a % len(names)]
        rec = nd["pulsars"][name]
        be = rec["backends"]
        mjd = np.sort(rng.uniform(53000, 58478, N))
        which = rng.integers(0, len(be), N)
        psr = SimulatedPulsar(toas=ArrayTOAs(mjd, 0.5, flags=[{"f": be[k]} for k in which]), name=f"{name}_{a}", loc={"RAJ": float(raj[a]), "DECJ": float(decj[a])})
        make_ideal(psr)
        psrs.append(psr)
        noise["flags"].append(list(be))
        noise["efac"].append(np.array([1.0 if v is None else v for v in rec["efac"]]))
        noise["log10_equad"].append(np.array(rec["log10_t2equad"]))
        noise["log10_ecorr"].append(np.array(rec["log10_ecorr"]))
        noise["rn_log10_A"].append(rec["red_noise_log10_A"])
        noise["rn_gamma"].append(rec["red_noise_gamma"])
    return psrs, noise


def _wall(fn, reps=1):
    import torch
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def engine_clock_during(fn, seconds=0.5):
    """median engine clock [GHz] while `fn` loops on the current stream: pta_clock_probe (one wave on a side stream sampling
    s_memrealtime / s_memtime every 100 us; the slope between samples is the clock).  Returns {"GHz", "p05", "p95", "idle_GHz"}."""
    import torch
    from pta_replicator_amd import _lib
    ns = int((seconds + 0.25) * 1e4) + 16
    buf = torch.zeros((ns, 2), dtype=torch.int64, device="cuda")
    side = torch.cuda.Stream()
    fn()
    torch.cuda.synchronize()
    _lib.call("pta_clock_probe", buf.data_ptr(), ns, int((seconds + 0.2) * 1e6), 100, side.cuda_stream)
    time.sleep(0.06)
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < seconds:
        fn()
        n += 1
        if n % 8 == 0:
            torch.cuda.current_stream().synchronize()
    torch.cuda.synchronize()
    smp = buf.cpu().numpy()
    smp = smp[smp[:, 0] > 0]
    if len(smp) < 8:
        return None
    rt, sc = smp[:, 0].astype(np.float64), smp[:, 1].astype(np.float64)
    ghz = np.diff(sc) / (np.diff(rt) * 10.0)
    tt = (rt[1:] - rt[0]) / 1e8
    busy = (tt > 0.06 + 0.25 * seconds) & (tt < 0.06 + 0.9 * seconds)
    idle = tt < 0.04
    if busy.sum() < 8:
        return None
    return {"GHz": float(np.median(ghz[busy])), "p05": float(np.percentile(ghz[busy], 5)), "p95": float(np.percentile(ghz[busy], 95)),
            "idle_GHz": float(np.median(ghz[idle])) if idle.any() else None,
            "method": "pta_clock_probe: s_memtime / s_memrealtime slope, 100 us samples on a side stream beside the loop"}


def td_ragged_numbers(P=42, R=256, compare_per_matrix=True, counts=None):
    """TD mode on an ng15-like RAGGED array (VERDICT r3 #1b): TOA counts log-uniform 500 ... 35 000, sum = 340 k - the normal shape of real
    data (the reference loops over pulsars: red_noise.py:286-298).  `potrf` = ALL pulsars as one end-aligned schedule (pta_potrf_ragged);
    `per_matrix` = the batch-by-equal-order scheme of rounds 1-3 on the same array (P batches of one)."""
    import torch
    from pta_replicator_amd.engine import ReplicaEngine
    from pta_replicator_amd import device as dv
    counts = ragged_counts(P) if counts is None else [int(c) for c in counts]
    P = len(counts)
    psrs, noise = ragged_array(counts)
    eng = configure_engine(ReplicaEngine(psrs, seed=7), noise)
    if os.environ.get("PTA_TD_POTRF_FLAGS"):              # A/B aid: chains / panel width / no look-ahead of the ragged schedule
        eng.td_potrf_flags = int(os.environ["PTA_TD_POTRF_FLAGS"], 0)
    eng.prepare()
    eng.prepare_td()
    flop = sum(float(n) ** 3 for n in counts) / 3.0
    res = {"n_psr": P, "n_toa_min": min(counts), "n_toa_max": max(counts), "n_toa_total": int(sum(counts)), "factor_buffer_GB": eng.d_Ltd.numel() * 8 / 1e9,
           "potrf_TFLOP": flop / 1e12, "schedule": eng.td_potrf_mode_used}
    ta = min(_wall(eng.td_assemble) for _ in range(2))
    cov_bytes = 8.0 * sum(n * (n + 1) / 2 for n in counts)       # algorithmic: the lower triangles, written once
    res.update({"cov_assemble_ms": ta * 1e3, "cov_assemble_kernel": getattr(eng, "td_cov_kernel_used", None), "cov_assemble_TBps": cov_bytes / ta / 1e12})
    try:   # the tile kernel on the same array (its grid is sized by the LARGEST pulsar: most workgroups of a ragged launch leave at once)
        tt = min(_wall(lambda: eng.td_assemble(kernel="tile")) for _ in range(2))
        res.update({"cov_assemble_tile_ms": tt * 1e3, "cov_assemble_tile_TBps": cov_bytes / tt / 1e12})
    except Exception as e:  # pragma: no cover
        res["cov_assemble_tile_error"] = str(e)[:200]

    def timed_factor(mode):
        ts = []
        for _ in range(2):
            eng.td_assemble()
            ts.append(_wall(lambda: eng.td_factorise(mode=mode)))
        return min(ts)
    t_r = timed_factor("ragged")
    res.update({"potrf_ms": t_r * 1e3, "potrf_TFLOPs": flop / t_r / 1e12, "potrf_frac_of_fp64_mfma_peak": flop / t_r / 1e12 / FP64_MFMA_PEAK_TFLOPS})
    if compare_per_matrix:
        t_u = timed_factor("uniform")
        res["per_matrix_schedule"] = {"potrf_ms": t_u * 1e3, "potrf_TFLOPs": flop / t_u / 1e12, "note": "batches of one (rounds 1-3: runs of equal TOA count share a launch sequence)"}
        eng.td_assemble()
        eng.td_factorise(mode="ragged")
    out = dv.empty((R, eng.n_toa))
    eng.generate_td(R, out=out)
    t = _wall(lambda: eng.generate_td(R, out=out), 2)
    fl = float(sum(float(n) ** 2 for n in counts))
    res.update({"generate_td_realisations": R, "generate_td_ms": t * 1e3, "realisations_per_s": R / t, "trmm_useful_TFLOPs": fl * R / t / 1e12,
                "trmm_frac_of_fp64_mfma_peak": fl * R / t / 1e12 / FP64_MFMA_PEAK_TFLOPS, "finite": bool(torch.isfinite(out).all())})
    return res


def grid_cell(P, N, td=True, seed=20260921, td_gb_limit=200.0):
    """one (N_psr, N_toa) cell of the north_star's grid: the headline workload's recipe (ng15 noise values cycled, HD GWB + RN + per-backend
    EFAC / EQUAD / ECORR) at P pulsars x N TOAs - throughput mode (realisations/s, per-kernel ms, fractions of the 8 TB/s and 78.6 TFLOP/s
    roofs) and TD mode (assembly, batched Cholesky, L.z) where the factors fit."""
    import ctypes
    import torch
    from pta_replicator_amd import _lib, device as dv
    t0 = time.perf_counter()
    eng, psrs, noise = build_engine(P, N, seed)
    torch.cuda.synchronize()
    cell = {"n_psr": P, "n_toa": N, "prepare_s": time.perf_counter() - t0}
    ntot = eng.n_toa
    R = int(max(16, min(1024, (6 << 30) // (8 * ntot)) // 16 * 16))
    out = dv.empty((R, ntot))
    eng.generate(R, out=out)
    one = _wall(lambda: eng.generate(R, out=out))
    K = int(max(2, min(50, 0.4 / max(one, 1e-4))))
    step = _wall(lambda: eng.generate(R, out=out), K)
    npts, Nf = eng.plan.gw_npts, eng.grid["Nf"]
    s = dv.stream_ptr()
    ws = eng.workspace(R)
    kern = {}

    def timed(name, fn, reps=3):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        fn(); torch.cuda.synchronize()
        ev[0].record()
        for _ in range(reps):
            fn()
        ev[1].record(); torch.cuda.synchronize()
        kern[name] = ev[0].elapsed_time(ev[1]) / reps
    gwb_kernel = "pta_gwb_czt" if eng.use_czt else "pta_gwb_idft_rng"
    if eng.use_czt:
        timed("pta_gwb_czt", lambda: _lib.call("pta_gwb_czt", eng.seed, 0, None, 0, R, P, Nf, npts, 10, *[dv.ptr(x) for x in eng.d_czt], dv.ptr(ws["G0"]), npts, 0, 0, s))
    else:
        timed("pta_gwb_idft_rng", lambda: _lib.call("pta_gwb_idft_rng", eng.seed, 0, R, P, Nf, dv.ptr(eng.d_Tsym), dv.ptr(eng.d_rot), npts, dv.ptr(ws["G0"]), npts, eng.idft_variant, 0, s))
    timed("pta_gwb_mix", lambda: _lib.call("pta_gwb_mix", dv.ptr(eng.d_M), P, dv.ptr(ws["G0"]), R, npts, npts, dv.ptr(ws["G"]), 0, s))
    timed("pta_engine_synth", lambda: _lib.call("pta_engine_synth", ctypes.byref(eng.plan), eng.seed, 0, R, dv.ptr(out), out.stride(0), s))
    n_fft = 2 * Nf - 2
    n_epochs = int(sum(len(v) for v in eng.ecorrvec))
    flops_alg = 4.0 * P * P * Nf + 5.0 * n_fft * np.log2(n_fft) * P + 2.0 * eng.K * ntot + 10.0 * ntot
    dom = max(kern, key=kern.get)
    cell["throughput"] = {"realisations_per_step": R, "ms_per_step": step * 1e3, "realisations_per_s": R / step, "toa_per_s": R * ntot / step,
                          "kernels_ms": {k: round(v, 4) for k, v in kern.items()}, "dominant_kernel": dom,
                          "synth_alg_bytes_frac_of_hbm": 8.0 * ntot * R / (kern["pta_engine_synth"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                          "step_alg_bytes_frac_of_hbm": 8.0 * ntot * R / step / 1e9 / HBM_PEAK_GBS,
                          "step_frac_of_fp64_peak": flops_alg * R / step / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                          "tiles": int(eng.plan.n_tiles), "tile_fill": ntot / (eng.plan.n_tiles * 256.0), "Nf": int(Nf), "npts": int(npts)}
    del out
    if td:
        gb = 8.0 * P * float(N + (N & 1)) * ((N + (N & 1) + 15) // 16 * 16) / 1e9
        if gb > td_gb_limit:
            cell["td"] = {"skipped": f"{gb:.0f} GB of factors do not fit beside the workspace"}
        else:
            try:
                if os.environ.get("PTA_TD_POTRF_MODE"):      # A/B aid: "ragged" runs uniform batches through the end-aligned schedule too
                    eng.td_potrf_mode = os.environ["PTA_TD_POTRF_MODE"]
                if os.environ.get("PTA_TD_POTRF_FLAGS"):
                    eng.td_potrf_flags = int(os.environ["PTA_TD_POTRF_FLAGS"], 0)
                eng.prepare_td()
                flop = P * float(N) ** 3 / 3.0
                ta = min(_wall(eng.td_assemble) for _ in range(2))
                tf = []
                for _ in range(2):
                    eng.td_assemble()
                    tf.append(_wall(eng.td_factorise))
                tf = min(tf)
                Rt = int(max(32, min(1024, (3 << 30) // (8 * ntot)) // 32 * 32))
                o2 = dv.empty((Rt, ntot))
                eng.generate_td(Rt, out=o2)
                tg = _wall(lambda: eng.generate_td(Rt, out=o2), 2)
                cell["td"] = {"factor_GB": gb, "schedule": eng.td_potrf_mode_used, "cov_assemble_ms": ta * 1e3, "cov_assemble_kernel": getattr(eng, "td_cov_kernel_used", None), "cov_assemble_TBps_written": 8.0 * P * N * (N + 1) / 2 / ta / 1e12,
                              "potrf_ms": tf * 1e3, "potrf_TFLOPs": flop / tf / 1e12, "potrf_frac": flop / tf / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                              "generate_td_realisations": Rt, "generate_td_ms": tg * 1e3, "realisations_per_s": Rt / tg,
                              "trmm_useful_TFLOPs": P * float(N) ** 2 * Rt / tg / 1e12, "trmm_frac": P * float(N) ** 2 * Rt / tg / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                              "finite": bool(torch.isfinite(o2).all())}
                del o2
            except Exception as e:  # pragma: no cover
                cell["td"] = {"error": str(e)[:300]}
    eng.d_Ltd = None
    del eng
    torch.cuda.empty_cache()
    return cell


def orf_numbers(P=200, lmax=4, reps=5):
    """the anisotropic ORF basis of BASELINE config 5's geometry (200 pulsars, l <= 4: all 20 100 pairs x 25 modes - 12-15 minutes of
    Python in the reference, spharmORFbasis.py:385-434): HIP events around pta_orf_basis and pta_orf_combine on the
    stream they are launched on, the host's pair-separation loop (the reference's own scalar arithmetic, kept on the host) timed beside them"""
    import torch
    from pta_replicator_amd import _lib, device as dv, spharmORFbasis as anis
    rng = np.random.default_rng(200)
    raj, decj = rng.uniform(0, 24, P), np.degrees(np.arcsin(rng.uniform(-1, 1, P)))
    locs = np.ascontiguousarray(np.stack([raj * np.pi / 12.0, np.pi / 2.0 - np.radians(decj)], axis=1))
    t0 = time.perf_counter()
    zc = anis.pair_zeta_cos(locs)
    t_host = time.perf_counter() - t0
    locs_d, zc_d = dv.f64(locs), dv.f64(zc)
    nb = (lmax + 1) ** 2
    basis, orf = dv.zeros((nb, P, P)), dv.empty((P, P))
    clm = dv.f64(np.concatenate([[np.sqrt(4 * np.pi)], 0.1 * np.random.default_rng(200).standard_normal(nb - 1)]))
    s = dv.stream_ptr()

    def ev_time(fn):
        fn()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        torch.cuda.synchronize()
        ev[0].record()
        for _ in range(reps):
            fn()
        ev[1].record()
        torch.cuda.synchronize()
        return ev[0].elapsed_time(ev[1]) / reps

    tb = ev_time(lambda: _lib.call("pta_orf_basis", dv.ptr(locs_d), dv.ptr(zc_d), P, lmax, dv.ptr(basis), s))
    tc = ev_time(lambda: _lib.call("pta_orf_combine", dv.ptr(basis), dv.ptr(clm), nb, P, dv.ptr(orf), s))
    return {"n_psr": P, "lmax": lmax, "pairs": P * (P + 1) // 2, "modes": nb, "orf_basis_ms": tb, "orf_combine_ms": tc,
            "host_pair_separations_ms": t_host * 1e3, "finite": bool(torch.isfinite(orf).all()),
            "reference": "spharmORFbasis.correlated_basis: 12-15 min of Python for the same 20 100 pairs (BASELINE.md §2)"}


def compact_line(full):
    """the ONE JSON line of the contract, kept under ~6 KB: the driver's record keeps the flat keys of `roofline` / `cpu_baseline` and the
    tail of the line, so every figure DESIGN.md quotes is a flat scalar here and the TD-mode ones come last; the complete record (nested
    blocks, run lists, grid cells, API timing) goes to gpurun_out/bench_full.json and to stderr."""
    def g(d, *path, default=None):
        for k in path:
            if not isinstance(d, dict) or k not in d:
                return default
            d = d[k]
        return d

    def r(x, nd=4):
        return round(float(x), nd) if isinstance(x, (int, float)) and not isinstance(x, bool) else x

    roof = full.get("roofline") or {}
    td = full.get("td_mode") or {}
    step = full.get("step") or {}
    flat = {k: roof.get(k) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "engine_clock_GHz")}
    flat["traffic_source"] = "committed rocprofv3 PMC pass, keyed by kernel sources + launch shape" if roof.get("traffic") else None
    flat["valu_insts_per_output_element"] = r(g(roof, "valu_issue", "insts_valu_per_output_element"), 1)
    flat["valu_issue_frac_of_launch_at_measured_clock"] = r(g(roof, "valu_issue", "frac_of_launch_at_measured_clock"))
    flat["step_frac_of_fp64_peak"] = r(step.get("frac_of_fp64_peak"))
    flat["step_normals_T_per_s"] = r(step.get("normals_T_per_s"))
    flat["gwb_stage_kernel"] = g(roof, "also", "kernel")
    flat["gwb_stage_ms"] = r(g(roof, "also", "stage_ms"))
    flat["gwb_stage_frac_of_fp64_peak"] = r(g(roof, "also", "frac"))
    orf = full.get("orf_config5") or {}
    flat["orf_basis_ms_P200_lmax4"] = r(orf.get("orf_basis_ms"))
    flat["orf_combine_ms_P200_lmax4"] = r(orf.get("orf_combine_ms"))
    flat["orf_host_pair_loop_ms_P200"] = r(orf.get("host_pair_separations_ms"), 1)
    # TD mode (the north_star's dense path) on the SAME 68 x 5000 array, then the ragged arrays: flat scalars, fractions of 78.6 TFLOP/s / 8 TB/s
    flat["td_cov_kernel"] = td.get("cov_assemble_kernel")
    flat["td_cov_ms"] = r(td.get("cov_assemble_ms"))
    flat["td_cov_TBps"] = r((td.get("cov_assemble_GBps_lower_triangle") or 0) / 1e3) if td.get("cov_assemble_GBps_lower_triangle") else None
    flat["td_cov_frac_hbm"] = r((td.get("cov_assemble_GBps_lower_triangle") or 0) / HBM_PEAK_GBS) if td.get("cov_assemble_GBps_lower_triangle") else None
    flat["td_cov_walk_ms"] = r(td.get("cov_assemble_walk_ms"))
    flat["td_cov_tile_ms"] = r(td.get("cov_assemble_tile_ms"))
    flat["td_potrf_ms"] = r(td.get("potrf_ms"))
    flat["td_potrf_TFLOPs"] = r(td.get("potrf_TFLOPs"), 2)
    flat["td_potrf_frac"] = r(td.get("potrf_frac_of_fp64_mfma_peak"))
    flat["td_potrf_mfma_busy_pct_committed_pmc"] = r(td.get("potrf_trailing_update_mfma_busy_pct"), 1)
    flat["td_trmm_ms_per_1024"] = r(td.get("generate_td_ms"))
    flat["td_trmm_TFLOPs"] = r(td.get("trmm_useful_TFLOPs"), 2)
    flat["td_trmm_frac"] = r(td.get("trmm_frac_of_fp64_mfma_peak"))
    flat["td_realisations_per_s"] = r(td.get("realisations_per_s"), 1)
    flat["td_prepare_first_call_ms"] = r(td.get("prepare_td_first_call_ms"), 1)
    flat["td_prepare_warm_ms"] = r(td.get("prepare_td_ms"), 1)
    flat["td_ragged_potrf_TFLOPs"] = r(g(td, "ragged", "potrf_TFLOPs"), 2)
    flat["td_ragged_potrf_frac"] = r(g(td, "ragged", "potrf_frac_of_fp64_mfma_peak"))
    flat["td_ragged_trmm_frac"] = r(g(td, "ragged", "trmm_frac_of_fp64_mfma_peak"))
    flat["td_ragged_cov_TBps"] = r(g(td, "ragged", "cov_assemble_TBps"))
    flat["td_config2_potrf_frac"] = r(g(td, "config2_shape", "potrf_frac_of_fp64_mfma_peak"))
    flat["td_config2_realisations_per_s"] = r(g(td, "config2_shape", "realisations_per_s"), 1)
    # what THIS box's matrix pipe and HBM deliver in the same run (boxes of the pool differ by several per cent: 71-78 TFLOP/s on the
    # register-tile microbenchmark, 2.17-2.30 GHz under the VALU kernels), and the TD fractions against it
    mb = full.get("microbench") or {}
    flat["box_fp64_mfma_microbench_TFLOPs"] = r(mb.get("fp64_mfma_tile_tflops"), 2)
    flat["box_hbm_write_microbench_TBps"] = r(mb.get("hbm_write_TBps"), 3)
    if mb.get("fp64_mfma_tile_tflops"):
        flat["td_potrf_frac_of_box_microbench"] = r((td.get("potrf_TFLOPs") or 0) / mb["fp64_mfma_tile_tflops"]) if td.get("potrf_TFLOPs") else None
        flat["td_trmm_frac_of_box_microbench"] = r((td.get("trmm_useful_TFLOPs") or 0) / mb["fp64_mfma_tile_tflops"]) if td.get("trmm_useful_TFLOPs") else None
    cb = full.get("cpu_baseline") or {}
    cpu = {k: cb.get(k) for k in ("value", "unit", "cores", "kind", "value_without_ecorr", "host_cpus", "error") if k in cb}
    if cb.get("sample"):
        cpu["sample"] = cb["sample"][:160]
    if g(cb, "reference_container", "value"):
        cpu["unmodified_reference_in_build_container"] = g(cb, "reference_container", "value")
    cfg = dict(full.get("config") or {})
    if isinstance(cfg.get("workload"), str):
        cfg["workload"] = cfg["workload"][:230]
    line = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                                     "dtype", "data")}
    line["config"] = cfg
    for k in ("value_fast_rng_math", "value_gwb_grid_draws", "value_single_deviate_wn", "api_mode_ms", "rccl_ranks_seen", "backend", "ms_per_step_per_rank"):
        if k in full:
            line[k] = r(full[k], 1) if isinstance(full[k], float) else full[k]
    if isinstance(full.get("gathered_to_rank0"), dict):
        line["gathered_to_rank0"] = {k: full["gathered_to_rank0"].get(k) for k in ("realisations", "ms", "realisations_per_s", "error") if k in full["gathered_to_rank0"]}
    if isinstance(full.get("config4_shape"), dict):
        line["config4_realisations_per_s"] = r(full["config4_shape"].get("realisations_per_s"), 1)
    line["kernels_ms"] = full.get("kernels_ms")
    if isinstance(full.get("gpu_over_cpu"), dict):
        line["gpu_over_cpu"] = {k: r(v, 0) for k, v in full["gpu_over_cpu"].items()}
    line["grid"] = [{"P": c.get("n_psr"), "N": c.get("n_toa"), "real_per_s": r(g(c, "throughput", "realisations_per_s"), 0),
                     "td_potrf_frac": r(g(c, "td", "potrf_frac"), 3), "td_trmm_frac": r(g(c, "td", "trmm_frac"), 3)} for c in (full.get("grid") or []) if isinstance(c, dict)]
    line["full_record"] = "gpurun_out/bench_full.json (+ stderr): td_mode, grid, step, engine_clocks, api_mode, cpu_baseline run lists, microbench"
    line["cpu_baseline"] = cpu
    line["roofline"] = {k: (r(v) if isinstance(v, float) else v) for k, v in flat.items()}
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=1024, help="realisations per step per GPU (BASELINE.json config 3: 1024)")
    ap.add_argument("--psr", type=int, default=68)
    ap.add_argument("--toa", type=int, default=5000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-td", action="store_true", help="skip the TD-mode (dense covariance / Cholesky / L.z) measurement")
    ap.add_argument("--no-gather", action="store_true", help="N > 1: skip the timed generate+gather-to-rank-0 pipeline")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary measurements (config-4 shape, drop-in API timing): profiling runs")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # dry-run aids for a 1-GPU development box (never set by the driver): PTA_BENCH_SINGLE_DEVICE=1 puts every rank on cuda:0,
    # PTA_BENCH_BACKEND=gloo runs the control plane (barrier / all_reduce) without RCCL - together they exercise the N > 1
    # control flow of this file (env handling, per-rank realisation ranges, max-over-ranks timing, one JSON line) on one GPU
    if os.environ.get("PTA_BENCH_SINGLE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("PTA_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"

    import ctypes
    from pta_replicator_amd import _lib, device as dv
    eng, psrs, noise = build_engine(args.psr, args.toa, seed=20260921)
    R, K, W = args.batch, args.steps, args.warmup
    out = dv.empty((R, eng.n_toa))
    base = rank * R * (K + W)   # disjoint realisation ranges per rank: same stream, any GPU count

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_steps():
        for i in range(W):
            eng.generate(R, r0=base + i * R, out=out)
        barrier()
        t0 = time.perf_counter()
        for i in range(K):
            eng.generate(R, r0=base + (W + i) * R, out=out)
        barrier()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    elapsed = timed_steps()
    # ---- secondary numbers, same K steps: (i) opt-in fp32 Gaussian transform; (ii) GWB drawn on the npts-sample grid through the
    # factor of its covariance (npts instead of 2 Nf normals per pulsar; same distribution, not replayable
    # through the reference's frequency-domain algebra).  `value` stays the default: fp64 transform, reference-order draws.
    eng.rng_fast = 1
    elapsed_fast = timed_steps()
    eng.rng_fast = 0
    eng.gwb_mode = "grid"
    elapsed_grid = timed_steps()
    eng.gwb_mode = "fourier"
    # (iii) EFAC / EQUAD drawn with ONE deviate per TOA of the combined amplitude sqrt((efac sigma)^2 + (efac equad)^2) instead of the
    # reference's two (white_noise.py:105-109): same distribution, half the Box-Muller pairs of the fused kernel, not replayable
    eng.wn_mode = "single"
    elapsed_single = timed_steps()
    eng.wn_mode = "reference"

    # ---- N > 1: what the ranks actually did - an all_reduce of ones over RCCL (ranks seen), every rank's own
    # ms_per_step (all_gather), and BASELINE.json config 4's shape: 2048 realisations per GPU of the same array + one CGW ----
    multi = {}
    if world > 1:
        ones = torch.ones(1, dtype=torch.float64, device="cuda")
        dist.all_reduce(ones)
        mine = torch.tensor([0.0], dtype=torch.float64, device="cuda")
        barrier()
        t0 = time.perf_counter()
        for i in range(K):
            eng.generate(R, r0=base + (W + i) * R, out=out)
        torch.cuda.synchronize()
        mine[0] = (time.perf_counter() - t0) / K * 1e3
        allms = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allms, mine)
        multi = {"rccl_ranks_seen": int(round(float(ones.item()))), "backend": dist.get_backend(),
                 "ms_per_step_per_rank": [round(float(x.item()), 4) for x in allms]}

    def config4_shape():
        """BASELINE.json config 4 on this rank: the same 68 x 5000 array + one continuous-wave source (the reference test's CW
        parameters), 2048 realisations per GPU (16384 over 8), timed like the headline step; returns whole-job realisations/s."""
        from pta_replicator_amd.distributed import shard_range
        e4 = configure_engine(type(eng)(psrs, seed=20260921), noise)
        e4.add_cgw(gwtheta=np.pi / 2, gwphi=2.5, mc=1e9, dist=5.0, fgw=1e-8, phase0=0.5, psi=1.5, inc=np.pi / 4, pdist=1.0, pphase=None,
                   psrTerm=True, evolve=True, phase_approx=False, tref=53000 * 86400)
        e4.prepare()
        per = 2048
        lo, hi = shard_range(per * world, rank=rank, world=world)
        buf = dv.empty((per, e4.n_toa))
        e4.generate(per, r0=lo, out=buf)
        barrier()
        t0 = time.perf_counter()
        for _ in range(3):
            e4.generate(per, r0=lo, out=buf)
        barrier()
        el = (time.perf_counter() - t0) / 3
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        del buf
        return {"realisations_per_gpu": per, "realisations_total": per * world, "ms": el * 1e3, "realisations_per_s": per * world / el,
                "workload": "BASELINE.json config 4 shape: config 3's array + one CGW (deterministic.py:13-185 parameters of the reference test), "
                            "realisation ranges by shard_range(); generation only (the gather to rank 0 is `gathered_to_rank0`)"}

    cfg4 = None
    if not args.no_extras:
        try:
            cfg4 = config4_shape()
        except Exception as e:  # pragma: no cover
            cfg4 = {"error": str(e)[:300]}

    # ---- per-kernel times of one step: HIP events on the stream the kernels are launched on ----
    kern = {}
    s = dv.stream_ptr()
    ws = eng.workspace(R)
    eng.generate(R, r0=base, out=out)          # fills the workspace pointers of the plan
    npts, Nf, P = eng.plan.gw_npts, eng.grid["Nf"], eng.P

    def timed(name, fn, reps=3):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        fn()
        torch.cuda.synchronize()
        ev[0].record()
        for _ in range(reps):
            fn()
        ev[1].record()
        torch.cuda.synchronize()
        kern[name] = ev[0].elapsed_time(ev[1]) / reps

    timed("pta_engine_rn_coef", lambda: _lib.call("pta_engine_rn_coef", eng.seed, 0, R, P, eng.K, dv.ptr(eng.d_amp), dv.ptr(ws["coef"]), 0, s))
    timed("pta_gwb_idft_rng", lambda: _lib.call("pta_gwb_idft_rng", eng.seed, 0, R, P, Nf, dv.ptr(eng.d_Tsym), dv.ptr(eng.d_rot), npts, dv.ptr(ws["G0"]), npts,
                                                eng.idft_variant, 0, s))
    if eng.use_czt:
        timed("pta_gwb_czt", lambda: _lib.call("pta_gwb_czt", eng.seed, 0, None, 0, R, P, Nf, npts, 10, *[dv.ptr(x) for x in eng.d_czt], dv.ptr(ws["G0"]), npts, 0, 0, s))
    timed("pta_gwb_mix", lambda: _lib.call("pta_gwb_mix", dv.ptr(eng.d_M), P, dv.ptr(ws["G0"]), R, npts, npts, dv.ptr(ws["G"]), 0, s))
    timed("pta_engine_synth", lambda: _lib.call("pta_engine_synth", ctypes.byref(eng.plan), eng.seed, 0, R, dv.ptr(out), out.stride(0), s))

    # ---- engine clock while the hot kernels run: in-kernel probe on a side stream, N = 1 only ----
    clocks = {}
    if world == 1 and not args.no_extras:
        try:
            clocks["pta_engine_synth"] = engine_clock_during(lambda: _lib.call("pta_engine_synth", ctypes.byref(eng.plan), eng.seed, 0, R, dv.ptr(out), out.stride(0), s), 0.4)
            if eng.use_czt:
                clocks["pta_gwb_czt"] = engine_clock_during(lambda: _lib.call("pta_gwb_czt", eng.seed, 0, None, 0, R, P, Nf, npts, 10, *[dv.ptr(x) for x in eng.d_czt],
                                                                              dv.ptr(ws["G0"]), npts, 0, 0, s), 0.4)
            clocks["step"] = engine_clock_during(lambda: eng.generate(R, r0=base, out=out), 0.4)
        except Exception as e:  # pragma: no cover
            clocks["error"] = str(e)[:200]

    def gather_phase(line):
        """N > 1: the north_star's gather of the residual arrays to rank 0, pipelined with generation.  Runs LAST and under a
        60 s watchdog: the RCCL path cannot be exercised on the 1-GPU development box (two ranks on one device over gloo move
        CUDA tensors at ~50 MB/s: scripts/gpu_gather_debug.py checks the logic there at a reduced width), so if it were to hang, rank 0 still
        prints the line it has (without the gather figure) and every rank exits."""
        import threading

        def bail():
            if rank == 0:
                line["gathered_to_rank0"] = {"error": "timed out after 60 s (watchdog)"}
                print(json.dumps(compact_line(line)), flush=True)
            os._exit(0)
        dog = threading.Timer(60.0, bail)
        dog.daemon = True
        dog.start()
        try:
            from pta_replicator_amd.distributed import generate_gathered
            full = generate_gathered(eng, world * R, r0=0, chunk=256)      # warm-up (allocations, RCCL channels)
            del full
            barrier()
            tg = time.perf_counter()
            full = generate_gathered(eng, world * R, r0=0, chunk=256)
            barrier()
            tg = time.perf_counter() - tg
            del full
            res = {"realisations": world * R, "ms": tg * 1e3, "realisations_per_s": world * R / tg,
                   "gather_ms": tg * 1e3, "generate_only_ms_same_realisations": elapsed / K * 1e3,
                   "note": "every rank generates its shard in chunks of 256 while the previous chunk travels; rank 0 receives straight into the final tensor"}
        except Exception as e:  # pragma: no cover
            res = {"error": str(e)[:300]}
        dog.cancel()
        if line is not None:
            line["gathered_to_rank0"] = res

    if rank != 0:
        if world > 1:
            if not args.no_gather:
                gather_phase(None)
            dist.destroy_process_group()
        return

    # ---- roofline of the dominant kernel ----
    # algorithmic work per realisation: bytes = 8 * sum N_a (the residual array written once);
    # flops of the GWB frequency->time stage as the reference writes it = 4 P^2 Nf (M @ w) + 5 n log2 n P (ifft, n = 2Nf-2)
    n_fft = 2 * Nf - 2
    alg_bytes = 8.0 * eng.n_toa * R
    alg_flops_gwb = (4.0 * P * P * Nf + 5.0 * n_fft * np.log2(n_fft) * P) * R
    gwb_kernel = "pta_gwb_czt" if eng.use_czt else "pta_gwb_idft_rng"
    # executed flops: chirp-z = two 4096-point complex FFTs (5 N log2 N each) + chirp products per row; DFT-GEMM = 2 M K N
    exe_flops = {"pta_gwb_czt": (2 * 5.0 * 4096 * 12 + 6.0 * (4096 + 2 * (Nf - 2) + npts)) * R * P,
                 "pta_gwb_idft_rng": 2.0 * (R * P) * (2.0 * (Nf - 2)) * ((npts + 1) // 2)}

    def hbm_roof(k):
        ach = alg_bytes / (kern[k] * 1e-3) / 1e9
        d = {"kernel": k, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
             "traffic": None, "avg_launch_ms": kern[k]}
        e, why = pmc_entry("k_engine_synth_mfma<false, false>", SYNTH_SRC, R=R, n_toa=eng.n_toa)
        if e:   # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (KiB) of this very launch shape and these very sources
            d["traffic"] = (e["fetch_kib"] + e["write_kib"]) * 1024.0
            d["traffic_source"] = e.get("source")
            d["traffic_note"] = "FETCH_SIZE uncorrected (gfx950 under-counts wide streaming reads by up to 2x, MI355X_MICROARCH.md)"
            if "insts_valu" in e:   # 4 issue cycles per wave64 VALU instruction, 256 CUs x 4 SIMDs, 2.4 GHz peak engine clock
                issue_ms = e["insts_valu"] * 4.0 / (256 * 4) / 2.4e9 * 1e3
                d["valu_issue"] = {"insts_valu": e["insts_valu"], "insts_valu_per_output_element": e["insts_valu"] * 64.0 / (R * eng.n_toa),
                                   "issue_ms_at_2.4GHz": issue_ms, "frac_of_launch": issue_ms / kern[k], "valu_busy_pmc": e.get("valu_busy")}
        else:
            d["traffic_note"] = why
        ck = clocks.get(k)
        if ck:   # the clock the chip holds under this kernel (DVFS by instruction mix: 2.40 GHz idle / under MFMA, ~2.3 under fp64 VALU + Philox)
            d["engine_clock_GHz"] = ck["GHz"]
            if "valu_issue" in d:
                im = d["valu_issue"]["insts_valu"] * 4.0 / (256 * 4) / (ck["GHz"] * 1e9) * 1e3
                d["valu_issue"]["issue_ms_at_measured_clock"] = im
                d["valu_issue"]["frac_of_launch_at_measured_clock"] = im / kern[k]
            d["frac_at_measured_clock"] = d["frac"]   # an HBM roof does not move with the engine clock; the issue-bound view is valu_issue
        return d

    def flop_roof(k):
        # the GWB frequency -> time STAGE as the reference writes it is M @ w (4 P^2 Nf flop) + the inverse FFT (5 n log2 n P); on the
        # device the M @ w term runs in pta_gwb_mix behind the transform (linearity), so the stage's algorithmic flops are divided by
        # the time of BOTH kernels (dividing by the transform kernel alone flattered it); the transform kernel
        # alone is quoted against the inverse-FFT flops only, and `executed_tflops` are the flops it actually issues
        stage_ms = kern[k] + kern["pta_gwb_mix"]
        ach = alg_flops_gwb / (stage_ms * 1e-3) / 1e12
        bound = "mfma" if k == "pta_gwb_idft_rng" else "valu-fp64"
        ifft_flops = 5.0 * n_fft * np.log2(n_fft) * P * R
        d = {"kernel": k, "stage_kernels": [k, "pta_gwb_mix"], "bound": bound, "achieved": ach, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
             "frac": ach / FP64_MFMA_PEAK_TFLOPS, "traffic": None, "stage_ms": stage_ms, "avg_launch_ms": kern[k],
             "transform_kernel_alone": {"algorithmic_ifft_tflops": ifft_flops / (kern[k] * 1e-3) / 1e12,
                                        "frac": ifft_flops / (kern[k] * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                                        "executed_tflops": exe_flops[k] / (kern[k] * 1e-3) / 1e12,
                                        "executed_frac": exe_flops[k] / (kern[k] * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS},
             "executed_tflops": exe_flops[k] / (kern[k] * 1e-3) / 1e12}
        if k == "pta_gwb_czt":
            e, why = pmc_entry("k_gwb_czt<true, false, 15>", CZT_SRC, rows=R * P)
            if e:
                d["traffic"] = (e["fetch_kib"] + e["write_kib"]) * 1024.0
                d["traffic_source"] = e.get("source")
                if e.get("insts_valu"):
                    issue_ms = e["insts_valu"] * 4.0 / (256 * 4) / 2.4e9 * 1e3
                    d["valu_issue"] = {"insts_valu": e["insts_valu"], "insts_valu_per_row": e["insts_valu"] * 64.0 / (R * P),
                                       "issue_ms_at_2.4GHz": issue_ms, "frac_of_launch": issue_ms / kern[k], "valu_busy_pmc": e.get("valu_busy")}
            else:
                d["traffic_note"] = why
        ck = clocks.get(k)
        if ck:
            d["engine_clock_GHz"] = ck["GHz"]
            d["frac_at_measured_clock"] = ach / (FP64_MFMA_PEAK_TFLOPS * ck["GHz"] / 2.4)
        return d

    if kern["pta_engine_synth"] >= kern[gwb_kernel]:
        roof, other = hbm_roof("pta_engine_synth"), flop_roof(gwb_kernel)
    else:
        roof, other = flop_roof(gwb_kernel), hbm_roof("pta_engine_synth")
        if roof["bound"] != "mfma":   # the contract's enum: the chirp-z kernel runs on the fp64 vector pipe, same 78.6 TFLOP/s peak
            roof["bound_detail"], roof["bound"] = roof["bound"], "mfma"
    roof["also"] = other
    if "pta_gwb_idft_rng" in kern and gwb_kernel != "pta_gwb_idft_rng":
        roof["alternative_gwb_transform"] = flop_roof("pta_gwb_idft_rng")

    micro = {}
    if world == 1:   # the scaling runs only need the headline number; microbench / TD mode / CPU baseline are N = 1 extras
        try:
            res = ctypes.c_double(0.0)
            # fp64_mfma_tile: 4 A x 4 B fragments -> 16 accumulators (the GEMM kernels' pattern): reaches the 78.6 TFLOP/s spec;
            # fp64_mfma_8acc: round 1's loop (8 accumulators, one operand pair): 49 - limited by the dependent-accumulate
            # latency, NOT a ceiling of the matrix pipe
            # fp64_mfma_plus_fma: both loops on alternating waves of the same SIMDs, SUM of the two rates - they share the DP ALUs
            for kind, name in ((5, "fp64_mfma_tile_tflops"), (0, "fp64_mfma_8acc_tflops"), (1, "fp64_fma_tflops"), (6, "fp64_mfma_plus_fma_tflops"),
                               (2, "hbm_write_TBps"), (4, "normals_T_per_s")):
                _lib.call("pta_microbench", kind, 1 << 30, 2000 if kind in (0, 1) else (20 if kind == 2 else 200), 0, ctypes.byref(res))
                micro[name] = round(res.value, 3)
        except Exception as e:  # pragma: no cover
            micro["error"] = str(e)

    td = None
    if not args.no_td and world == 1:
        try:
            td = td_mode_numbers(eng, 1024)   # the whole array: 68 x 5000^2 fp64 = 13.6 GB of factors
        except Exception as e:  # pragma: no cover
            td = {"error": str(e)[:300]}
        if not args.no_extras:
            eng.d_Ltd = None                  # release the 13.6 GB before the 48 GB of the ragged array
            eng._td_prepared = False
            torch.cuda.empty_cache()
            try:   # TD mode on an ng15-like RAGGED array: 42 pulsars, TOA counts log-uniform 500 ... 35 000, sum = 340 915 (VERDICT r3 #1b)
                td["ragged"] = td_ragged_numbers()
                if td.get("potrf_TFLOPs"):
                    td["ragged"]["potrf_TFLOPs_over_uniform_68x5000"] = td["ragged"]["potrf_TFLOPs"] / td["potrf_TFLOPs"]
            except Exception as e:  # pragma: no cover
                td["ragged"] = {"error": str(e)[:300]}
            torch.cuda.empty_cache()
            try:   # BASELINE.json config 2's SHAPE in TD mode: the TOA counts of the reference's test_partim set (par/*.par:17), 256 realisations
                td["config2_shape"] = dict(td_ragged_numbers(R=256, compare_per_matrix=False, counts=(7758, 23023, 35037)),
                                           workload="TOA counts of test_partim (B1855+09 7758, B1937+21 23023, J1909-3744 35037; two of them odd), synthetic TOAs, "
                                                    "ng15 noise values cycled; parity on the real tim file: tests/test_gpu_configs.py::test_config2_td_mode")
            except Exception as e:  # pragma: no cover
                td["config2_shape"] = {"error": str(e)[:300]}
            torch.cuda.empty_cache()

    # ---- four cells of the (N_psr, N_toa) grid the north_star asks for (the full grid: scripts/gpu_grid_sweep.py -> profiles/r04_grid.json) ----
    grid = None
    if world == 1 and not args.no_extras:
        grid = []
        for gp, gn, gtd in ((3, 122, True), (16, 1000, True), (3, 10000, True), (200, 10000, False)):
            try:
                grid.append(grid_cell(gp, gn, td=gtd))
            except Exception as e:  # pragma: no cover
                grid.append({"n_psr": gp, "n_toa": gn, "error": str(e)[:200]})

    # ---- step level (VERDICT r2 #1b): the whole step against the algorithmic flop and deviate counts of SURVEY.md §8d ----
    n_epochs = int(sum(len(v) for v in eng.ecorrvec)) if getattr(eng, "ecorrvec", None) else 0
    flops_alg = 4.0 * P * P * Nf + 5.0 * n_fft * np.log2(n_fft) * P + 2.0 * eng.K * eng.n_toa + 10.0 * eng.n_toa
    deviates = 2.0 * P * Nf + P * eng.K + 2.0 * eng.n_toa + n_epochs
    step_s = elapsed / K
    step_block = {"flops_alg_per_realisation": flops_alg, "deviates_per_realisation": deviates,
                  "achieved_TFLOPs": flops_alg * R / step_s / 1e12, "frac_of_fp64_peak": flops_alg * R / step_s / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                  "normals_T_per_s": deviates * R / step_s / 1e12,
                  "frac_of_rng_microbench": (deviates * R / step_s / 1e12 / micro["normals_T_per_s"]) if micro.get("normals_T_per_s") else None,
                  "engine_clock_GHz": (clocks.get("step") or {}).get("GHz"),
                  "frac_of_fp64_peak_at_measured_clock": (flops_alg * R / step_s / 1e12 / (FP64_MFMA_PEAK_TFLOPS * clocks["step"]["GHz"] / 2.4)) if clocks.get("step") else None,
                  "sum_kernels_ms": sum(kern[k] for k in ("pta_engine_rn_coef", gwb_kernel, "pta_gwb_mix", "pta_engine_synth")),
                  "note": "flops = 4 P^2 Nf + 5 n log2 n P + 2 K sum N_a + 10 sum N_a; deviates = 2 P Nf + P K + 2 sum N_a + sum E_a (SURVEY.md §8d)"}

    line = {
        "metric": "realizations/sec, 68 psr x 5000 TOAs GWB+RN+WN", "value": world * R * K / elapsed, "unit": "realizations/s",
        "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"BASELINE.json config 3: the {args.psr} pulsars of ng15_dict.json x {args.toa} synthetic TOAs - HD GWB (gw_log10_A of the dict) + "
                               f"per-pulsar power-law RN (dict values, 30 components) + per-backend EFAC / t2EQUAD / ECORR (dict values), on-chip Philox "
                               f"draws, {R} realisations per step per GPU",
                   "realisations_per_step_per_gpu": R, "n_toa_total": eng.n_toa, "Nf": Nf, "npts": npts, "parallelism": f"replica-shard x{world}"},
        "value_fast_rng_math": world * R * K / elapsed_fast,
        "value_gwb_grid_draws": world * R * K / elapsed_grid,
        "value_single_deviate_wn": world * R * K / elapsed_single,
        "roofline": roof, "step": step_block, "kernels_ms": {k: round(v, 4) for k, v in kern.items()}, "microbench": micro,
    }
    if world > 1:
        line.update(multi)
    if td is not None:
        line["td_mode"] = td
        c5 = os.path.join(ROOT, "profiles", "r05_config5_td_mode.json")   # BASELINE config 5 in TD mode (160 GB of factors): a committed
        if os.path.exists(c5):                                            # measurement of scripts/gpu_config5_td.py, NOT re-run here
            try:
                td["config5_committed_measurement"] = dict(json.load(open(c5)), source="profiles/r05_config5_td_mode.json (scripts/gpu_config5_td.py)")
            except Exception as exc:
                td["config5_committed_measurement"] = {"error": str(exc)}
    if cfg4 is not None:
        line["config4_shape"] = cfg4
    if grid is not None:
        line["grid"] = grid
        line["grid_full"] = "profiles/r04_grid.json / r04_grid.txt (scripts/gpu_grid_sweep.py: P in {3, 16, 68, 200} x N in {122, 1000, 5000, 10000, 35000})"
    if clocks:
        line["engine_clocks"] = clocks
    if world == 1 and not args.no_extras:
        try:
            api = api_mode_timing(psrs, noise)
            line["api_mode_ms"] = api["loop"]["total"]
            line["api_mode"] = api
        except Exception as e:  # pragma: no cover
            line["api_mode"] = {"error": str(e)[:300]}
    if world > 1 and not args.no_gather:
        gather_phase(line)
    if world == 1 and not args.no_cpu_baseline:
        try:
            line["cpu_baseline"] = cpu_baseline(psrs, noise)
            line["gpu_over_cpu"] = {"with_reference_dense_U_ecorr": line["value"] / line["cpu_baseline"]["value"],
                                    "without_ecorr": line["value"] / line["cpu_baseline"]["value_without_ecorr"]}
            # the UNMODIFIED reference timed in the build container (the only place /root/reference exists): a committed record,
            # not measured in this run - on the GPU box the live figure above is the NumPy port (kind "port")
            ref_rec = os.path.join(ROOT, "profiles", "r03_cpu_baseline_reference.json")
            if os.path.exists(ref_rec):
                with open(ref_rec) as fh:
                    rc_ = json.load(fh)
                line["cpu_baseline"]["reference_container"] = rc_
                line["gpu_over_cpu"]["vs_reference_container_with_dense_U_ecorr"] = line["value"] / rc_["value"]
                line["gpu_over_cpu"]["vs_reference_container_without_ecorr"] = line["value"] / rc_["value_without_ecorr"]
        except Exception as e:  # pragma: no cover
            line["cpu_baseline"] = {"error": str(e)[:400]}
    if world == 1 and not args.no_extras:
        try:
            line["orf_config5"] = orf_numbers()
        except Exception as e:  # pragma: no cover
            line["orf_config5"] = {"error": str(e)[:300]}
    full_txt = json.dumps(line)
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "bench_full.json"), "w") as fh:
            fh.write(full_txt + "\n")
    except OSError:  # pragma: no cover
        pass
    print(full_txt, file=sys.stderr, flush=True)
    print(json.dumps(compact_line(line)), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
