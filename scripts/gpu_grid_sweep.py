"""The north_star's (N_psr, N_toa) grid: realisations/s and roofline fractions off the two shapes every earlier figure was
taken at - including the sizes the reference ships (3 x 122: test_partim_small; 7758 / 23023 / 35037: test_partim).

    python scripts/gpu_grid_sweep.py [--cells 3x122,68x5000] [--no-td] [--out gpurun_out/r04_grid.json]

Every cell = bench.grid_cell(P, N): the headline recipe (ng15 noise values cycled, HD GWB + RN + per-backend EFAC / EQUAD / ECORR) at P
pulsars x N TOAs, throughput mode and TD mode (where the dense factors fit).  The JSON is rewritten after every cell."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--cells", default="")
ap.add_argument("--no-td", action="store_true")
ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r06_grid.json"))
ap.add_argument("--cpu", action="store_true", help="also time the NumPy port on this host per cell (bench_extras.grid_cell_cpu; slow for the large cells)")
args = ap.parse_args()
if args.cells:
    cells = [tuple(int(x) for x in c.split("x")) for c in args.cells.split(",")]
else:
    cells = [(P, N) for N in (122, 1000, 5000, 10000, 35000) for P in (3, 16, 68, 200)]
os.makedirs(os.path.dirname(args.out), exist_ok=True)
res = {"grid": [], "peaks": {"hbm_GBps": bench.HBM_PEAK_GBS, "fp64_TFLOPs": bench.FP64_MFMA_PEAK_TFLOPS}}
t_all = time.perf_counter()
for P, N in cells:
    t0 = time.perf_counter()
    try:
        cell = bench.grid_cell(P, N, td=not args.no_td, cpu=args.cpu)
    except Exception as e:
        cell = {"n_psr": P, "n_toa": N, "error": str(e)[:300]}
    cell["cell_wall_s"] = time.perf_counter() - t0
    res["grid"].append(cell)
    print(json.dumps(cell), flush=True)
    with open(args.out, "w") as fh:
        json.dump(res, fh, indent=1)
res["total_wall_s"] = time.perf_counter() - t_all
with open(args.out, "w") as fh:
    json.dump(res, fh, indent=1)

# the CPU column: the UNMODIFIED reference timed per cell in the build container (scripts/cpu_grid_reference.py ->
# profiles/r06_grid_cpu_reference.json: the only host where /root/reference is mounted), joined by (P, N)
cpu_ref = {}
try:
    with open(os.path.join(ROOT, "profiles", "r06_grid_cpu_reference.json")) as fh:
        cr = json.load(fh)
    cpu_ref = {(c["n_psr"], c["n_toa"]): c for c in cr["cells"]}
    res["cpu_reference"] = {"source": "profiles/r06_grid_cpu_reference.json", "host": cr.get("host"), "date": cr.get("date"), "what": cr.get("what")}
    for c in res["grid"]:
        r = cpu_ref.get((c.get("n_psr"), c.get("n_toa")))
        if r and "throughput" in c:
            c["cpu_reference"] = {k: r[k] for k in ("realisations_per_s", "realisations_per_s_without_ecorr", "cores", "pulsars_timed")}
            c["gpu_over_cpu_reference"] = c["throughput"]["realisations_per_s"] / r["realisations_per_s"]
            c["gpu_over_cpu_reference_without_ecorr"] = c["throughput"]["realisations_per_s"] / r["realisations_per_s_without_ecorr"]
    with open(args.out, "w") as fh:
        json.dump(res, fh, indent=1)
except (OSError, KeyError, ValueError):
    pass

# text table
lines = ["# (N_psr, N_toa) grid, scripts/gpu_grid_sweep.py: throughput mode | TD mode (dense factors); fractions of 8 TB/s (algorithmic bytes) and 78.6 TFLOP/s",
         f"{'P':>4s} {'N':>6s} | {'R':>5s} {'ms/step':>9s} {'real/s':>10s} {'dominant':>18s} {'ms':>8s} {'synth/HBM':>9s} {'step/fp64':>9s} {'fill':>5s} | "
         f"{'GB':>6s} {'asm TB/s':>8s} {'potrf TF':>8s} {'frac':>5s} {'L.z TF':>7s} {'frac':>5s} {'real/s':>9s} | "
         f"{'CPU ref/s':>10s} {'cores':>5s} {'no-ECORR/s':>10s} {'GPU/CPU':>9s}"]
lines.insert(1, "# CPU columns: the UNMODIFIED reference (add_gwb + add_measurement_noise + add_jitter + add_red_noise under oracle/_stubs) timed per cell in the 8-vCPU build "
                "container (profiles/r06_grid_cpu_reference.json); 'no-ECORR' drops add_jitter's dense-U matvec; GPU/CPU = throughput-mode realisations/s over the reference's")
for c in res["grid"]:
    if "error" in c:
        lines.append(f"{c['n_psr']:4d} {c['n_toa']:6d} | ERROR {c['error']}")
        continue
    t = c["throughput"]
    row = (f"{c['n_psr']:4d} {c['n_toa']:6d} | {t['realisations_per_step']:5d} {t['ms_per_step']:9.3f} {t['realisations_per_s']:10.0f} {t['dominant_kernel']:>18s} "
           f"{t['kernels_ms'][t['dominant_kernel']]:8.3f} {t['synth_alg_bytes_frac_of_hbm']:9.3f} {t['step_frac_of_fp64_peak']:9.3f} {t['tile_fill']:5.2f} | ")
    d = c.get("td")
    if d and "potrf_ms" in d:
        row += (f"{d['factor_GB']:6.1f} {d['cov_assemble_TBps_written']:8.2f} {d['potrf_TFLOPs']:8.1f} {d['potrf_frac']:5.2f} {d['trmm_useful_TFLOPs']:7.1f} "
                f"{d['trmm_frac']:5.2f} {d['realisations_per_s']:9.0f}")
    elif d:
        row += f"{str(d.get('skipped') or d.get('error'))[:66]:66s}"
    else:
        row += " " * 66
    r = c.get("cpu_reference")
    if r:
        row += f" | {r['realisations_per_s']:10.4f} {r['cores']:5d} {r['realisations_per_s_without_ecorr']:10.3f} {c['gpu_over_cpu_reference']:9.2e}"
    else:
        row += " | (reference not timed for this cell: a realisation takes minutes)"
    lines.append(row)
txt = "\n".join(lines) + "\n"
with open(args.out.replace(".json", ".txt"), "w") as fh:
    fh.write(txt)
print(txt)
