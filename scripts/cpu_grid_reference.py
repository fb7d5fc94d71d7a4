"""The CPU column of the (N_psr, N_toa) grid (north_star: "next to the reference NumPy/libstempo path timed on the node's own
host cores"): the UNMODIFIED reference (add_gwb / add_measurement_noise / add_jitter / add_red_noise under the stubs of
oracle/run_reference.py; red_noise.py:106-298, white_noise.py:47-198) on every cell's workload (bench.headline_array(P, N): the grid's
recipe), in the build container - the only host where /root/reference is mounted.  Single BLAS thread and all cores, one warm-up + 2 timed
repeats, ECORR included (the reference's dense-U matvec) and excluded, as BASELINE.md §3 asks.  Cells whose realisation takes longer than a
minute are timed on a sub-sample of pulsars (stated per cell; add_gwb always runs on the whole array).

    python scripts/cpu_grid_reference.py [--cells 3x122,...]  ->  profiles/r06_grid_cpu_reference.json
"""
import datetime
import json
import os
import platform
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import cpu_baseline, headline_array  # noqa: E402

cells = [(P, N) for N in (122, 1000, 5000, 10000) for P in (3, 16, 68, 200)] + [(3, 35000)]
if "--cells" in sys.argv:
    cells = [tuple(int(x) for x in c.split("x")) for c in sys.argv[sys.argv.index("--cells") + 1].split(",")]
out = os.path.join(ROOT, "profiles", "r06_grid_cpu_reference.json")
res = {"date": datetime.date.today().isoformat(), "host": {"container": "build container (no GPU)", "cpus": os.cpu_count(), "machine": platform.machine()},
       "what": "unmodified reference under oracle/_stubs (numeric core, PINT sink excluded); bench.headline_array(P, N) workloads", "cells": []}
try:
    with open("/proc/cpuinfo") as fh:
        res["host"]["cpu_model"] = next((ln.split(":", 1)[1].strip() for ln in fh if ln.startswith("model name")), None)
except OSError:
    pass
for P, N in cells:
    # pulsars timed per cell: all of them while a realisation stays under about a minute (ECORR's dense U costs ~ 9e-9 s x N^2 per pulsar here)
    per_psr = 9e-9 * N * N + 2e-5 * N
    S = P if per_psr * P < 60 else max(4, int(60 / per_psr))
    S = min(S, P)
    t0 = time.perf_counter()
    psrs, noise = headline_array(P, N)
    threads = (1, os.cpu_count()) if N <= 10000 else (1,)
    rec = cpu_baseline(psrs, noise, subset=S, repeats=2 if N <= 10000 else 1, timeout=7200, threads=threads)
    assert rec["kind"] == "reference", "run this where /root/reference is mounted"
    cell = {"n_psr": P, "n_toa": N, "pulsars_timed": S, "realisations_per_s": rec["value"], "realisations_per_s_without_ecorr": rec["value_without_ecorr"],
            "cores": rec["cores"], "kind": rec["kind"], "sample": rec["sample"],
            "single_thread": rec.get("single_thread"), "all_cores": rec.get("all_cores"), "wall_s": time.perf_counter() - t0}
    res["cells"].append(cell)
    print(json.dumps({k: cell[k] for k in ("n_psr", "n_toa", "pulsars_timed", "realisations_per_s", "realisations_per_s_without_ecorr", "cores", "wall_s")}), flush=True)
    with open(out, "w") as fh:
        json.dump(res, fh, indent=1)
