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
                (PMC HBM bytes per launch) comes from the committed rocprofv3 passes in profiles/r0N_pmc.json and is
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


PMC_FILE = next((p for p in (os.path.join(ROOT, "profiles", f"r0{n}_pmc.json") for n in (6, 5)) if os.path.exists(p)),
                os.path.join(ROOT, "profiles", "r06_pmc.json"))
SYNTH_SRC = ("pta_engine_kernels.hip", "pta_rng.h", "pta_rng_tables.h", "pta_mfma.h")
TD_SRC = ("pta_td_kernels.hip", "pta_gemm.hip", "pta_potrf.hip", "pta_rng.h", "pta_rng_tables.h", "pta_mfma.h")
CZT_SRC = ("pta_czt_kernels.hip", "pta_fft.h", "pta_rng.h", "pta_rng_tables.h")


def pmc_entry(key, srcs, **shape):
    """counters of kernel `key` from the committed rocprofv3 --pmc passes (scripts/gpu_profile_r6.sh -> profiles/r06_pmc.json; round 5's while that is absent),
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


def cpu_baseline(psrs, noise, subset=8, repeats=3, timeout=600, threads=None):
    """oracle/cpu_baseline.py once per entry of `threads` (default: BLAS threads = 1 and = all host cores) on a bounded sample of the
    same workload (whole-array add_gwb + the per-pulsar calls of `subset` pulsars scaled to the array; one warm-up + `repeats` timed
    repeats each).  kind = "reference" (the unmodified reference under stubs) where /root/reference is mounted, else "port"."""
    import subprocess
    import tempfile
    ncpu = os.cpu_count() or 1
    runs = {}
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "workload.npz")
        dump_workload(psrs, noise, path)
        for nt in (threads or (1, ncpu)):
            label = "single_thread" if nt == 1 else ("all_cores" if nt == ncpu else f"threads_{nt}")
            env = dict(os.environ, OPENBLAS_NUM_THREADS=str(nt), OMP_NUM_THREADS=str(nt), MKL_NUM_THREADS=str(nt))
            out = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "cpu_baseline.py"), path, "--subset", str(subset), "--repeats",
                                  str(repeats)], env=env, capture_output=True, text=True, timeout=timeout)
            if out.returncode != 0:
                raise RuntimeError(f"oracle/cpu_baseline.py failed: {out.stderr[-400:]}")
            runs[label] = json.loads(out.stdout.strip().splitlines()[-1])
            runs[label]["threads"] = nt
    best = min(runs.values(), key=lambda r: r["seconds_per_realisation"])
    P, N = len(psrs), len(psrs[0].toas.get_mjds().value)
    what = "unmodified reference under oracle/_stubs (PINT sink excluded)" if best["kind"] == "reference" else "NumPy port oracle/pta_oracle.py"
    scaled = "" if best["subset"] == P else f" (x{P}/{best['subset']})"
    rec = {"value": 1.0 / best["seconds_per_realisation"], "unit": "realisations/s", "cores": best["threads"], "kind": best["kind"],
           "sample": f"{what}: add_gwb on {P}x{N} + WN/dense-U ECORR/RN on {best['subset']} psr{scaled}; median of {best['repeats']} after 1 warm-up",
           "value_without_ecorr": 1.0 / best["seconds_without_ecorr"], "host_cpus": ncpu}
    for label, r in runs.items():
        rec[label] = {k: r[k] for k in ("seconds_per_realisation", "seconds_runs", "seconds_without_ecorr", "seconds_parts_last_run", "threads")}
    return rec


# what the driver's record keeps of the line (BENCH_r05.json): the first 24 keys of `roofline`, names cut at 40 characters, strings at 120
ROOFLINE_MAX_KEYS = 24
ROOFLINE_MAX_NAME = 40
LINE_MAX_STRING = 120


def compact_line(full):
    """the ONE JSON line of the contract.  `roofline` carries AT MOST 24 flat keys with names of AT MOST 40 characters (the driver's
    record keeps no more: two rounds of TD-mode figures were cut off behind longer lists - tests/test_host_logic.py pins both limits):
    the contract's eight, then the dense path of the north_star (BASELINE.json's "fp64 Cholesky MFMA util %" half of the metric), then
    the step-level and box-level context.  Everything else that used to ride in `roofline` is in `roofline_more`; the complete record
    (nested blocks, run lists, grid cells, API timing) goes to gpurun_out/bench_full.json and to stderr."""
    def g(d, *path, default=None):
        for k in path:
            if not isinstance(d, dict) or k not in d:
                return default
            d = d[k]
        return d

    def r(x, nd=4):
        return round(float(x), nd) if isinstance(x, (int, float)) and not isinstance(x, bool) else x

    def cut(x):
        return x[:LINE_MAX_STRING] if isinstance(x, str) else x

    roof = full.get("roofline") or {}
    td = full.get("td_mode") or {}
    step = full.get("step") or {}
    mb = full.get("microbench") or {}
    orf = full.get("orf_config5") or {}
    cov_gbps = td.get("cov_assemble_GBps_lower_triangle")
    flat = {k: roof.get(k) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms")}
    flat["td_potrf_frac"] = r(td.get("potrf_frac_of_fp64_mfma_peak"))
    flat["td_potrf_mfma_busy_pct"] = r(td.get("potrf_trailing_update_mfma_busy_pct"), 1)
    flat["td_trmm_frac"] = r(td.get("trmm_frac_of_fp64_mfma_peak"))
    flat["td_cov_frac_hbm"] = r(cov_gbps / HBM_PEAK_GBS) if cov_gbps else None
    flat["td_cov_TBps"] = r(cov_gbps / 1e3) if cov_gbps else None
    flat["td_realisations_per_s"] = r(td.get("realisations_per_s"), 1)
    flat["td_ragged_potrf_frac"] = r(g(td, "ragged", "potrf_frac_of_fp64_mfma_peak"))
    flat["td_prepare_first_call_ms"] = r(td.get("prepare_td_first_call_ms"), 1)
    flat["step_frac_of_fp64_peak"] = r(step.get("frac_of_fp64_peak"))
    flat["valu_insts_per_out_elem"] = r(g(roof, "valu_issue", "insts_valu_per_output_element"), 1)
    flat["box_fp64_mfma_TFLOPs"] = r(mb.get("fp64_mfma_tile_tflops"), 2)
    flat["box_hbm_write_TBps"] = r(mb.get("hbm_write_TBps"), 3)
    flat["orf_basis_ms_P200_lmax4"] = r(orf.get("orf_basis_ms"))
    flat["frac_rng"] = r(step.get("frac_of_rng_microbench"))
    flat["td_potrf_ms"] = r(td.get("potrf_ms"), 2)
    flat["td_trmm_ms_per_1024"] = r(td.get("generate_td_ms"), 2)
    flat = {k: cut(r(v) if isinstance(v, float) else v) for k, v in flat.items()}
    assert len(flat) <= ROOFLINE_MAX_KEYS and all(len(k) <= ROOFLINE_MAX_NAME for k in flat)
    # the rest of what rounds 3-5 carried in `roofline` (the driver lists this object's name only; the values are in the line and the side file)
    more = {"engine_clock_GHz": roof.get("engine_clock_GHz"),
            "traffic_source": "committed rocprofv3 PMC pass, keyed by kernel sources + launch shape" if roof.get("traffic") else None,
            "valu_issue_frac_at_measured_clock": r(g(roof, "valu_issue", "frac_of_launch_at_measured_clock")),
            "bound_by_contract_enum": roof.get("bound_by_contract_enum"),
            "step_normals_T_per_s": r(step.get("normals_T_per_s")), "box_normals_T_per_s": r(mb.get("normals_T_per_s")),
            "gwb_stage_kernel": g(roof, "also", "kernel"), "gwb_stage_ms": r(g(roof, "also", "stage_ms")), "gwb_stage_frac_of_fp64_peak": r(g(roof, "also", "frac")),
            "orf_combine_ms_P200_lmax4": r(orf.get("orf_combine_ms")), "orf_host_pairs_ms_P200": r(orf.get("host_pair_separations_ms"), 2),
            "orf_host_pairs_python_loop_ms_P200": r(orf.get("host_pair_separations_python_loop_ms"), 1),
            "td_cov_kernel": td.get("cov_assemble_kernel"), "td_cov_ms": r(td.get("cov_assemble_ms")), "td_cov_walk_ms": r(td.get("cov_assemble_walk_ms")),
            "td_cov_tile_ms": r(td.get("cov_assemble_tile_ms")), "td_potrf_TFLOPs": r(td.get("potrf_TFLOPs"), 2), "td_potrf_schedule": td.get("potrf_schedule"),
            "td_potrf_right_looking_ms": r(td.get("potrf_right_looking_ms"), 2), "td_potrf_left_looking_ms": r(td.get("potrf_left_looking_ms"), 2),
            "td_trmm_TFLOPs": r(td.get("trmm_useful_TFLOPs"), 2), "td_prepare_warm_ms": r(td.get("prepare_td_ms"), 1),
            "td_ragged_potrf_TFLOPs": r(g(td, "ragged", "potrf_TFLOPs"), 2), "td_ragged_trmm_frac": r(g(td, "ragged", "trmm_frac_of_fp64_mfma_peak")),
            "td_ragged_cov_TBps": r(g(td, "ragged", "cov_assemble_TBps")), "td_config2_potrf_frac": r(g(td, "config2_shape", "potrf_frac_of_fp64_mfma_peak")),
            "td_config2_realisations_per_s": r(g(td, "config2_shape", "realisations_per_s"), 1)}
    if mb.get("fp64_mfma_tile_tflops"):   # the TD fractions against what THIS box's matrix pipe delivers in the same run (71-78 TFLOP/s over the pool)
        more["td_potrf_frac_of_box_microbench"] = r((td.get("potrf_TFLOPs") or 0) / mb["fp64_mfma_tile_tflops"]) if td.get("potrf_TFLOPs") else None
        more["td_trmm_frac_of_box_microbench"] = r((td.get("trmm_useful_TFLOPs") or 0) / mb["fp64_mfma_tile_tflops"]) if td.get("trmm_useful_TFLOPs") else None
    cb = full.get("cpu_baseline") or {}
    cpu = {k: cut(cb.get(k)) for k in ("value", "unit", "cores", "kind", "sample", "value_without_ecorr", "host_cpus", "error") if k in cb}
    if g(cb, "reference_container", "value"):   # the UNMODIFIED reference, timed in the build container on the whole array (committed record)
        cpu["reference_in_build_container"] = r(g(cb, "reference_container", "value"), 5)
        cpu["reference_without_ecorr"] = r(g(cb, "reference_container", "value_without_ecorr"), 4)
        cpu["reference_cores"] = g(cb, "reference_container", "cores")
        cpu["reference_date"] = g(cb, "reference_container", "date")
    cfg = dict(full.get("config") or {})
    cfg["workload"] = cut(cfg.get("workload"))
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
                     "td_potrf_frac": r(g(c, "td", "potrf_frac"), 3), "td_trmm_frac": r(g(c, "td", "trmm_frac"), 3),
                     "cpu_real_per_s": r(g(c, "cpu", "realisations_per_s"), 3), "cpu_kind": g(c, "cpu", "kind"), "cpu_cores": g(c, "cpu", "cores")}
                    for c in (full.get("grid") or []) if isinstance(c, dict)]
    line["full_record"] = "gpurun_out/bench_full.json (+ stderr): td_mode, grid, step, engine_clocks, api_mode, cpu_baseline run lists, microbench"
    line["roofline_more"] = more
    line["cpu_baseline"] = cpu
    line["roofline"] = flat
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
    from bench_extras import api_mode_timing, engine_clock_during, grid_cell, orf_numbers, td_mode_numbers, td_ragged_numbers
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
        # `bound` says what the counters say (VERDICT r5 #1b): 219 VALU wave-instructions per output element, VALU busy 74 % - the kernel is
        # bound by fp64 VALU issue (Philox + Box-Muller), not by HBM; `achieved` / `peak` / `frac` stay the contract's algorithmic bytes
        # against the 8 TB/s HBM roof (the contract's enum value for this line would be "hbm": bound_by_contract_enum), and the RNG-side
        # fraction is `frac_rng` = the step's normals/s over the same run's RNG microbenchmark
        d = {"kernel": k, "bound": "fp64-valu (rng)", "bound_by_contract_enum": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
             "frac": ach / HBM_PEAK_GBS, "traffic": None, "avg_launch_ms": kern[k]}
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

    # ---- four cells of the (N_psr, N_toa) grid the north_star asks for (the full grid: scripts/gpu_grid_sweep.py -> profiles/r06_grid.json, CPU reference column: profiles/r06_grid_cpu_reference.json) ----
    grid = None
    if world == 1 and not args.no_extras:
        grid = []
        for gp, gn, gtd in ((3, 122, True), (16, 1000, True), (3, 10000, True), (200, 10000, False)):
            try:
                grid.append(grid_cell(gp, gn, td=gtd, cpu=not args.no_cpu_baseline))
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
        line["grid_full"] = "profiles/r06_grid.json / r06_grid.txt + r06_grid_cpu_reference.json (scripts/gpu_grid_sweep.py: P in {3, 16, 68, 200} x N in {122, 1000, 5000, 10000, 35000})"
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
            ref_rec = os.path.join(ROOT, "profiles", "r06_cpu_baseline_reference.json")
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


def __getattr__(name):
    """the secondary measurements moved to bench_extras.py; `bench.grid_cell`, `from bench import api_mode_timing` ... still resolve"""
    if name.startswith("__"):
        raise AttributeError(name)
    import bench_extras
    try:
        return getattr(bench_extras, name)
    except AttributeError:
        raise AttributeError(f"module 'bench' has no attribute {name!r}") from None


if __name__ == "__main__":
    main()
