#!/usr/bin/env python3
"""Headline benchmark: realisations/sec, 68 pulsars x 5000 TOAs, GWB + RN + WN (EFAC/EQUAD + ECORR), fp64.

    python bench.py --gpus N --steps K --warmup W [--batch R]

A "step" is one pass of the hot path over one batch of R realisations of the whole array, every Gaussian deviate
drawn on chip, inputs resident in HBM (ReplicaEngine.generate: pta_engine_rn_coef -> pta_gwb_idft_rng ->
pta_gwb_mix -> pta_engine_synth).  N > 1: launched by torch.distributed.run, one rank per GPU; realisations are
independent, so rank g generates realisations [g*R*K .. ) of the same seeded stream (weak scaling, no data-path
collective; the north_star's gather of the residual arrays to rank 0 is timed separately and reported as
`gather_ms`, not folded into the step).  Rank 0 prints ONE JSON line.

Besides the contract fields the line carries
  roofline      the dominant kernel of the step against its bound (algorithmic units per launch / measured
                launch time; HBM peak 8 TB/s from MI355X_MICROARCH.md, fp64 matrix peak 78.6 TFLOP/s = AMD's
                public MI355X figure, cross-checked by the in-library microbenchmark)
  kernels       per-kernel times of one step (HIP events on the launch stream)
  cpu_baseline  the CPU oracle (a NumPy port of the reference's algebra, oracle/pta_oracle.py) timed on this
                host for one realisation of the same workload, with the reference's dense-U ECORR
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
# HBM bytes per launch from the PMC passes of the same command (scripts/gpu_profile.sh -> profiles/), KiB as rocprofv3 reports them;
# FETCH_SIZE is uncorrected (MI355X_MICROARCH.md: it under-counts wide streaming reads by up to 2x on gfx950)
# insts_valu = SQ_INSTS_VALU (wave instructions) per launch, valu_busy = VALUBusy from the stall-counter pass of the same
# summary file: the kernel is VALU-issue bound
PMC_TRAFFIC = {"pta_engine_synth": {"R": 960, "n_toa": 340000, "fetch_kib": 361309.0, "write_kib": 2669630.0, "insts_valu": 1.44648e9, "valu_busy": 0.769,
                                    "source": "profiles/r01_rocprofv3_summary_run81.txt"}}


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


def cpu_baseline(psrs, noise, repeats=1):
    """One realisation of the same workload through the CPU oracle, the way the reference spends its time:
    everything (ORF, design matrices, dense ECORR U) rebuilt per call.  Returns dict for the JSON line."""
    from oracle import pta_oracle as po
    P = len(psrs)
    mjd = [np.asarray(p.toas.get_mjds().value, dtype=np.float64) for p in psrs]
    tdb = [p.toas.table["tdbld"] for p in psrs]
    sig = [np.asarray(p.toas.get_errors().to("s").value) for p in psrs]
    locs = po.psr_locs_equatorial([p.loc for p in psrs])
    parts = {}

    def run(dense_u):
        t0 = time.perf_counter()
        grid = po.gwb_grid([float(m.min()) for m in mjd], [float(m.max()) for m in mjd])
        ORF = po.gwb_orf(locs)                      # pair loop in Python, like spharmORFbasis.correlated_basis
        M = np.linalg.cholesky(ORF)
        w = po.gwb_draws(16672, P, grid["Nf"])
        C = po.gwb_spectrum(grid["f"], grid["dur"], grid["howml"], noise["gw_log10_A"], 13. / 3.)
        po.gwb_dt(grid, M, w, C, [m * 86400 for m in mjd])
        t1 = time.perf_counter()
        for a in range(P):
            if noise["rn_log10_A"][a] is not None:
                (zr,) = po.legacy_normals(19870 + a, [60])
                po.red_noise_dt(tdb[a], noise["rn_log10_A"][a], noise["rn_gamma"][a], zr)
        t2 = time.perf_counter()
        for a in range(P):
            n = len(mjd[a])
            z1, z2 = po.legacy_normals(10660 + a, [n, n])
            po.measurement_noise_dt(sig[a], np.ones(n) * noise["efac"][a], np.ones(n) * 10 ** noise["log10_equad"][a], z1, z2)
        t3 = time.perf_counter()
        for a in range(P):
            epoch_of, ne, first, _ = po.quantize(mjd[a], dt=0.1)
            (ze,) = po.legacy_normals(17763 + a, [ne])
            ecv = po.jitter_ecorr_vector(ne, first, noise["log10_ecorr"][a])
            if dense_u:   # white_noise.py:37-39,182: dense N x E indicator matrix and matvec
                U = np.zeros((len(mjd[a]), ne), "d")
                U[np.arange(len(mjd[a])), epoch_of] = 1
                np.dot(U * ecv, ze)
            else:
                po.jitter_dt(epoch_of, ecv, ze)
        t4 = time.perf_counter()
        return dict(gwb=t1 - t0, rn=t2 - t1, wn=t3 - t2, ecorr=t4 - t3, total=t4 - t0)

    runs = [run(True) for _ in range(max(2, repeats))]       # ~14 s of single-core work: the bounded sample
    dense = {k: float(np.mean([r[k] for r in runs])) for k in runs[0]}
    gather = run(False)
    return {"value": 1.0 / dense["total"], "unit": "realisations/s", "cores": 1, "kind": "port",
            "sample": f"{len(runs)} realisations of the same {P} psr x {len(mjd[0])} TOA workload (GWB+RN+EFAC/EQUAD+ECORR) through "
                      f"oracle/pta_oracle.py, reference-style dense-U ECORR, everything rebuilt per call like the reference; "
                      f"NumPy BLAS threads = default, Python loop single-threaded",
            "seconds": {k: round(v, 4) for k, v in dense.items()},
            "value_ecorr_as_gather": 1.0 / gather["total"],
            "host_cpus": os.cpu_count()}


def td_mode_numbers(N, B, R):
    """Secondary metric of BASELINE.json: the dense time-domain path (no counterpart in the reference) - covariance
    assembly GB/s, blocked fp64 Cholesky TFLOP/s (MFMA trailing update) and L.Z TFLOP/s for B pulsars of N TOAs."""
    import ctypes
    import torch
    from pta_replicator_amd import _lib, device as dv
    s = dv.stream_ptr()
    nm = 30
    rng = np.random.default_rng(5)
    t = np.sort(rng.uniform(53000, 58478, N)) * 86400.0
    Tspan = t.max() - t.min()
    f = np.arange(1, nm + 1) / Tspan
    freqs = np.repeat(f, 2)
    phi = (10 ** -14.0) ** 2 * (freqs * 365.25 * 86400) ** (-3.0) / (12 * np.pi ** 2 * Tspan) * (365.25 * 86400) ** 3
    from pta_replicator_amd.white_noise import epoch_map
    epoch_of, first = epoch_map(t / 86400.0, 0.1)
    t_d, f_d, phi_d = dv.f64(t), dv.f64(f), dv.f64(phi)
    sig_d, ep_d, ec_d = dv.f64(np.full(N, 0.25e-12)), dv.i32(epoch_of), dv.f64(np.full(N, 4e-14))
    Ft = dv.empty((2 * nm, N))
    _lib.call("pta_rn_basis", dv.ptr(t_d), N, 0.0, dv.ptr(f_d), None, nm, 0, dv.ptr(Ft), N, s)
    C = dv.zeros((B, N, N))
    info = dv.zeros((B,), dtype=torch.int32)

    def assemble():
        for b in range(B):
            _lib.call("pta_td_cov_assemble", dv.ptr(Ft), N, N, 2 * nm, dv.ptr(phi_d), dv.ptr(sig_d), dv.ptr(ep_d), dv.ptr(ec_d),
                      ctypes.c_void_p(C.data_ptr() + 8 * b * N * N), N, s)

    def wall(fn):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
        return time.perf_counter() - t0

    assemble()
    ta = wall(assemble)
    tp = wall(lambda: _lib.call("pta_potrf_batched", dv.ptr(C), N, B, dv.ptr(info), s))
    ok = int(info.abs().sum().item()) == 0
    z, out = dv.empty((R, N)), dv.empty((R, N))
    _lib.call("pta_rng_fill_normal", 1, 0, R, (5 << 24), N // 2, 1, dv.ptr(z), None, N, s)
    _lib.call("pta_td_trmm", dv.ptr(C), N, N, dv.ptr(z), N, R, dv.ptr(out), N, 0, s)
    tt = wall(lambda: _lib.call("pta_td_trmm", dv.ptr(C), N, N, dv.ptr(z), N, R, dv.ptr(out), N, 0, s))
    potrf_tf = N ** 3 / 3.0 * B / tp / 1e12
    return {"n_toa": N, "n_psr": B, "positive_definite": ok,
            "cov_assemble_GBps": 8.0 * N * (N + 64) / 2 * B / ta / 1e9,
            "potrf_TFLOPs": potrf_tf, "potrf_frac_of_fp64_mfma_peak": potrf_tf / FP64_MFMA_PEAK_TFLOPS, "potrf_ms": tp * 1e3,
            "trmm_TFLOPs_executed": 2.0 * N * N * R / tt / 1e12, "trmm_realisations_per_s_per_pulsar": R / tt}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=960, help="realisations per step per GPU")
    ap.add_argument("--psr", type=int, default=68)
    ap.add_argument("--toa", type=int, default=5000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-td", action="store_true", help="skip the TD-mode (dense covariance / Cholesky / L.z) side measurement")
    ap.add_argument("--gather", action="store_true", help="also time the gather of the residual arrays to rank 0")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"

    from pta_replicator_amd import _lib, device as dv
    eng, psrs, noise = build_engine(args.psr, args.toa, seed=20260921)
    R, K, W = args.batch, args.steps, args.warmup
    out = dv.empty((R, eng.n_toa))
    base = rank * R * (K + W)   # disjoint realisation ranges per rank: same stream, any GPU count

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(W):
        eng.generate(R, r0=base + i * R, out=out)
    barrier()
    t0 = time.perf_counter()
    for i in range(K):
        eng.generate(R, r0=base + (W + i) * R, out=out)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- the same K steps with the opt-in fp32 Gaussian transform (secondary number; `value` stays the fp64-accurate one) ----
    _lib.call("pta_set_rng_math", 1)
    eng.generate(R, r0=base, out=out)
    barrier()
    t0 = time.perf_counter()
    for i in range(K):
        eng.generate(R, r0=base + (W + i) * R, out=out)
    barrier()
    elapsed_fast = time.perf_counter() - t0
    _lib.call("pta_set_rng_math", 0)
    if world > 1:
        t = torch.tensor([elapsed_fast], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed_fast = float(t.item())

    # ---- per-kernel times of one step: HIP events on the stream the kernels are launched on ----
    kern = {}
    s = dv.stream_ptr()
    ws = eng.workspace(R)
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

    import ctypes
    timed("pta_engine_rn_coef", lambda: _lib.call("pta_engine_rn_coef", eng.seed, 0, R, P, eng.K, dv.ptr(eng.d_amp), dv.ptr(ws["coef"]), s))
    timed("pta_gwb_idft_rng", lambda: _lib.call("pta_gwb_idft_rng", eng.seed, 0, R, P, Nf, dv.ptr(eng.d_Tsym), dv.ptr(eng.d_rot), npts, dv.ptr(ws["G0"]), npts, s))
    if eng.use_czt:
        timed("pta_gwb_czt", lambda: _lib.call("pta_gwb_czt", eng.seed, 0, None, 0, R, P, Nf, npts, 10, *[dv.ptr(x) for x in eng.d_czt], dv.ptr(ws["G0"]), npts, s))
    timed("pta_gwb_mix", lambda: _lib.call("pta_gwb_mix", dv.ptr(eng.d_M), P, dv.ptr(ws["G0"]), R, npts, npts, dv.ptr(ws["G"]), s))
    timed("pta_engine_synth", lambda: _lib.call("pta_engine_synth", ctypes.byref(eng.plan), eng.seed, 0, R, dv.ptr(out), out.stride(0), s))

    gather_ms = None
    if args.gather and world > 1:
        from pta_replicator_amd.distributed import gather_to_rank0
        barrier()
        tg = time.perf_counter()
        gather_to_rank0(out)
        barrier()
        gather_ms = (time.perf_counter() - tg) * 1e3

    if rank != 0:
        if world > 1:
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
        t = PMC_TRAFFIC.get(k)
        if t and t["R"] == R and t["n_toa"] == eng.n_toa:   # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (KiB) of this very launch shape
            d["traffic"] = (t["fetch_kib"] + t["write_kib"]) * 1024.0
            d["traffic_source"] = t["source"]
            if "insts_valu" in t:   # 4 issue cycles per wave64 VALU instruction, 256 CUs x 4 SIMDs, 2.4 GHz peak engine clock
                issue_ms = t["insts_valu"] * 4.0 / (256 * 4) / 2.4e9 * 1e3
                d["valu_issue"] = {"insts_valu": t["insts_valu"], "issue_ms_at_2.4GHz": issue_ms, "frac_of_launch": issue_ms / kern[k],
                                   "valu_busy_pmc": t.get("valu_busy")}   # SQ_ACTIVE_INST_VALU / CU_NUM / GRBM_GUI_ACTIVE, same profile
        return d

    def flop_roof(k):
        ach = alg_flops_gwb / (kern[k] * 1e-3) / 1e12
        bound = "mfma" if k == "pta_gwb_idft_rng" else "valu-fp64"
        return {"kernel": k, "bound": bound, "achieved": ach, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": ach / FP64_MFMA_PEAK_TFLOPS, "traffic": None, "executed_tflops": exe_flops[k] / (kern[k] * 1e-3) / 1e12,
                "avg_launch_ms": kern[k]}

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
    try:
        if world > 1:   # the scaling runs only need the headline number; microbench / TD mode are N=1 extras
            raise RuntimeError("skipped at N>1")
        res = ctypes.c_double(0.0)
        for kind, name in ((0, "fp64_mfma_tflops"), (1, "fp64_fma_tflops"), (2, "hbm_write_TBps"), (4, "normals_T_per_s")):
            _lib.call("pta_microbench", kind, 1 << 30, 2000 if kind in (0, 1) else (20 if kind == 2 else 200), ctypes.byref(res))
            micro[name] = round(res.value, 3)
    except Exception as e:  # pragma: no cover
        micro["error"] = str(e)

    td = None
    if not args.no_td and world == 1:
        try:
            td = td_mode_numbers(args.toa, args.psr, 512)   # the whole array: 68 x 5000^2 fp64 = 13.6 GB of covariance
        except Exception as e:  # pragma: no cover
            td = {"error": str(e)}

    line = {
        "metric": "realizations/sec, 68 psr x 5000 TOAs GWB+RN+WN", "value": world * R * K / elapsed, "unit": "realizations/s",
        "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{args.psr} pulsars x {args.toa} TOAs synthetic array (BASELINE.json config 3 geometry): HD GWB + per-pulsar "
                               f"power-law RN (30 components) + EFAC/EQUAD + ECORR, on-chip Philox draws, {R} realisations per step per GPU",
                   "realisations_per_step_per_gpu": R, "n_toa_total": eng.n_toa, "Nf": Nf, "npts": npts, "parallelism": f"replica-shard x{world}"},
        "value_fast_rng_math": world * R * K / elapsed_fast,
        "roofline": roof, "kernels_ms": {k: round(v, 4) for k, v in kern.items()}, "microbench": micro,
    }
    if td is not None:
        line["td_mode"] = td
    if gather_ms is not None:
        line["gather_ms"] = gather_ms
    if world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(psrs, noise)
        line["speedup_vs_cpu_baseline"] = line["value"] / line["cpu_baseline"]["value"]
    print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
