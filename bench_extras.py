#!/usr/bin/env python3
"""Secondary measurements of bench.py (kept out of the headline file): the dense TD path (covariance assembly, batched
Cholesky, L.z) on the headline array, on an ng15-like ragged array and on BASELINE config 2's shape; one cell of the (N_psr, N_toa)
grid with the CPU port timed beside it; engine clocks; the drop-in add_* API; the anisotropic ORF basis of config 5's geometry.
bench.py calls these at N = 1 unless --no-td / --no-extras; scripts/gpu_*.py call them directly.  `bench.<name>` still resolves."""
import json
import os
import sys
import time

import numpy as np

import bench as _b
from bench import (FP64_MFMA_PEAK_TFLOPS, HBM_PEAK_GBS, ROOT, TD_SRC, build_engine, configure_engine, ng15_noise, pmc_entry)  # noqa: F401


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
    # exactly the buffer prepare_td() asks for (orders padded to even, row pitches to 16 doubles: engine_td.prepare_td) - rounds 3-5 sized this
    # block by n x n, 0.16 % SMALLER than the factor buffer, so the caching allocator could not hand it back and the "first call" figure
    # silently contained a fresh 13.6 GB hipMalloc: the 94 ... 249 ms spread between boxes was the driver's allocation time
    nbytes = 8 * sum((n + (n & 1)) * ((n + (n & 1) + 15) // 16 * 16) for n in counts) + (4 << 20)   # + slack: a cached block serves any smaller request
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
        ts_free, bad = [], 0
        for _ in range(2):
            wall(assemble)
            ts_free.append(wall(lambda: _lib.call("pta_potrf_batched_ex", dv.ptr(eng.d_Ltd), n, ld, n * ld, P, dv.ptr(info), 0, s)))
            bad += int(info.abs().sum().item())
        # the panel orders of the workspace scheme, same kernels: left-looking (the engine's default since round 6), right-looking with the next
        # panel's diagonal phase run ahead (rounds 3-5), and the A/B forms of each (left + run-ahead diagonal phases; the C-tile prefetch epilogue)
        orders = {"left": _lib.POTRF_LEFT, "right": _lib.POTRF_DIAG_AHEAD, "left_diag_ahead": _lib.POTRF_LEFT | _lib.POTRF_DIAG_AHEAD,
                  "right_epi1": _lib.POTRF_DIAG_AHEAD | _lib.POTRF_EPI1}
        tord = {k: [] for k in orders}
        for rep in range(4):
            for k, fl in orders.items():
                if rep >= 2 and k not in ("left", "right"):
                    continue
                ta = wall(assemble)
                tord[k].append(wall(lambda: _lib.call("pta_potrf_batched_ws", dv.ptr(eng.d_Ltd), n, ld, n * ld, P, dv.ptr(info), fl, dv.ptr(work), need, s)))
                bad += int(info.abs().sum().item())
        res["potrf_left_looking_ms"] = min(tord["left"]) * 1e3
        res["potrf_right_looking_ms"] = min(tord["right"]) * 1e3
        res["potrf_left_diag_ahead_ms"] = min(tord["left_diag_ahead"]) * 1e3
        res["potrf_epi1_ms"] = min(tord["right_epi1"]) * 1e3
        res["potrf_schedule"] = getattr(eng, "td_potrf_order", "left") + "-looking panels, 2 chains (what prepare_td() runs)"
        info.add_(bad)
        del work
        tp = min(tord["left" if getattr(eng, "td_potrf_order", "left") == "left" else "right"])
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
            _lib.call("pta_potrf_batched_ws", dv.ptr(eng.d_Ltd), n, ld, n * ld, P, dv.ptr(info), _lib.POTRF_LEFT, dv.ptr(work2), need, s)
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


def grid_cell_cpu(psrs, noise, subset=8, repeats=2):
    """the CPU column of a grid cell (north_star: "next to the reference NumPy/libstempo path timed on the node's own host cores in the same
    run"): oracle/cpu_baseline.py on this cell's workload, single BLAS thread (the faster setting on every host measured so far), whole-array
    add_gwb + the per-pulsar calls of <= `subset` pulsars scaled to the array.  On the GPU box this is the NumPy port (kind "port"); the
    unmodified reference per cell is the committed profiles/r06_grid_cpu_reference.json (scripts/cpu_grid_reference.py, build container)."""
    rec = _b.cpu_baseline(psrs, noise, subset=min(len(psrs), subset), repeats=repeats, threads=(1,))
    return {"realisations_per_s": rec["value"], "realisations_per_s_without_ecorr": rec["value_without_ecorr"], "kind": rec["kind"], "cores": rec["cores"],
            "host_cpus": rec["host_cpus"], "sample": rec["sample"], "seconds_parts_last_run": rec["single_thread"]["seconds_parts_last_run"]}


def grid_cell(P, N, td=True, seed=20260921, td_gb_limit=200.0, cpu=False):
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
    if cpu:
        try:
            cell["cpu"] = grid_cell_cpu(psrs, noise)
            cell["gpu_over_cpu"] = cell["throughput"]["realisations_per_s"] / cell["cpu"]["realisations_per_s"]
        except Exception as e:  # pragma: no cover
            cell["cpu"] = {"error": str(e)[:300]}
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
    anis.pair_zeta_cos(locs)
    t0 = time.perf_counter()
    zc = anis.pair_zeta_cos(locs)                      # native arccos arguments + one NumPy arccos / cos + the sampled self-check
    t_host = time.perf_counter() - t0
    t0 = time.perf_counter()
    zc_loop = anis.pair_zeta_cos_loop(locs)            # the reference's per-pair scalar loop (round 5's host path; the checker)
    t_loop = time.perf_counter() - t0
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
            "host_pair_separations_ms": t_host * 1e3, "host_pair_separations_python_loop_ms": t_loop * 1e3,
            "host_pairs_native_equals_loop": bool(np.array_equal(zc, zc_loop)), "finite": bool(torch.isfinite(orf).all()),
            "reference": "spharmORFbasis.correlated_basis: 12-15 min of Python for the same 20 100 pairs (BASELINE.md §2)"}
