"""Turn the rocprofv3 CSV passes of scripts/gpu_profile_r3.sh into profiles/r03_pmc.json: the PMC figures bench.py quotes
(roofline.traffic, VALU issue, MFMA-busy %), each keyed by kernel + launch shape + a hash of the kernel sources."""
import collections, csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import CZT_SRC, SYNTH_SRC, TD_SRC, src_sha

out_dir, R, n_toa, n_psr = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
TAG = sys.argv[5] if len(sys.argv) > 5 else "r03"


def sums(pass_dir, match, grid=None):
    """mean per dispatch of every counter, over the dispatches of kernels whose name contains `match` (and, if given, whose
    Grid_Size equals `grid`).  rocprofv3 reports one row per (dispatch, counter), already summed over the 8 XCDs."""
    acc, disp = collections.Counter(), set()
    for p in glob.glob(os.path.join(out_dir, pass_dir, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(p)):
            if match in r["Kernel_Name"] and (grid is None or int(r["Grid_Size"]) == grid):
                acc[r["Counter_Name"]] += float(r["Counter_Value"])
                disp.add((p, r["Dispatch_Id"]))
    n = max(len(disp), 1)
    return {k: v / n for k, v in acc.items()}, len(disp)


def avg_ms(match):
    tot, n = 0.0, 0
    for p in glob.glob(os.path.join(out_dir, "trace", "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(p)):
            if match in r["Kernel_Name"]:
                tot += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
                n += 1
    return (tot / n if n else None), n


res = {}
SIMD_PER_XCD = 128   # GRBM_GUI_ACTIVE arrives summed over the 8 XCDs (checked: value / kernel duration = 8 x ~2 GHz), so
                     # GUI_ACTIVE x 128 SIMDs per XCD = SIMD-cycles of the whole chip during the dispatch
# ---- fused synthesis kernel
k = "k_engine_synth_mfma<false, false>"
f, nf = sums("pmc_fetch", k)
w, nw = sums("pmc_write", k)
a, na = sums("pmc_sq", k)
ms, nt = avg_ms(k)
if nf and nw:
    e = {"R": R, "n_toa": n_toa, "fetch_kib": f.get("FETCH_SIZE"), "write_kib": w.get("WRITE_SIZE"), "avg_launch_ms_rocprof": ms,
         "dispatches": {"fetch": nf, "write": nw, "sq": na, "trace": nt}, "src_sha": src_sha(*SYNTH_SRC),
         "source": f"profiles/{TAG}_rocprofv3_summary.txt (scripts/gpu_profile_{TAG[:1]}{int(TAG[1:])}.sh)"}
    if na:
        e["insts_valu"] = a.get("SQ_INSTS_VALU")
        if a.get("GRBM_GUI_ACTIVE"):
            e["valu_busy"] = a.get("SQ_ACTIVE_INST_VALU", 0.0) * 4 / (a["GRBM_GUI_ACTIVE"] * SIMD_PER_XCD) if a.get("SQ_ACTIVE_INST_VALU") else None
            e["gui_active_cycles_per_xcd_per_ns"] = a["GRBM_GUI_ACTIVE"] / 8 / (ms * 1e6) if ms else None   # NOT the engine clock: launch gaps are in the duration (r04_clocks.txt)
    res[k] = e
# ---- chirp-z kernel of the GWB stage (default variant, fp64 transform): one workgroup per (realisation, pulsar) row
k = "k_gwb_czt<true, false, 15>"
f, nf = sums("pmc_fetch", k)
w, nw = sums("pmc_write", k)
a, na = sums("pmc_sq", k)
ms, nt = avg_ms(k)
if nf and nw:
    e = {"rows": R * n_psr, "fetch_kib": f.get("FETCH_SIZE"), "write_kib": w.get("WRITE_SIZE"), "avg_launch_ms_rocprof": ms,
         "dispatches": {"fetch": nf, "write": nw, "sq": na, "trace": nt}, "src_sha": src_sha(*CZT_SRC),
         "source": f"profiles/{TAG}_rocprofv3_summary.txt (scripts/gpu_profile_{TAG[:1]}{int(TAG[1:])}.sh)"}
    if na:
        e["insts_valu"] = a.get("SQ_INSTS_VALU")
        if a.get("GRBM_GUI_ACTIVE") and a.get("SQ_ACTIVE_INST_VALU"):
            e["valu_busy"] = a["SQ_ACTIVE_INST_VALU"] * 4 / (a["GRBM_GUI_ACTIVE"] * SIMD_PER_XCD)
            e["gui_active_cycles_per_xcd_per_ns"] = a["GRBM_GUI_ACTIVE"] / 8 / (ms * 1e6) if ms else None   # NOT the engine clock: launch gaps are in the duration (r04_clocks.txt)
    res[k] = e
# ---- MFMA kernels of TD mode: busy % = SQ_VALU_MFMA_BUSY_CYCLES / SIMD-cycles, time-weighted over the kernel's dispatches; HBM bytes from
# the FETCH_SIZE / WRITE_SIZE passes (KiB per dispatch; this round those passes run WITH TD mode)
def large_dispatches(match, min_ms=1.0):
    """MFMA-busy % and engine clock over the dispatches of `match` that last at least min_ms, durations from the SAME pass's kernel
    trace (the time-weighted mean over ALL dispatches of a kernel is dominated by its many small launches)."""
    dur = {}
    for p in glob.glob(os.path.join(out_dir, "pmc_mfma", "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(p)):
            if match in r["Kernel_Name"]:
                dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    gui, busy, tot, n = 0.0, 0.0, 0.0, set()
    for p in glob.glob(os.path.join(out_dir, "pmc_mfma", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(p)):
            d = dur.get(r["Dispatch_Id"])
            if match in r["Kernel_Name"] and d is not None and d >= min_ms:
                if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                    gui += float(r["Counter_Value"])
                    if r["Dispatch_Id"] not in n:
                        n.add(r["Dispatch_Id"])
                        tot += d
                elif r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES":
                    busy += float(r["Counter_Value"])
    if not n or not gui:
        return None
    return {"dispatches": len(n), "min_ms": min_ms, "mfma_busy_pct": 100.0 * busy / (gui * SIMD_PER_XCD), "gui_active_cycles_per_xcd_per_ns": gui / 8 / (tot * 1e6)}


for k in ("k_dgemm_glds128", "k_td_trmm_rng", "k_td_trmm_rng<false, true>", "k_td_trmm_rng<false, false>", "k_td_cov128", "k_td_cov_walk", "k_diag128",
          "k_trsm_mfma", "k_potf2", "k_mb_mfma(", "k_mb_mfma_tile("):
    m, nm = sums("pmc_mfma", k)
    ms, nt = avg_ms(k)
    if nm and m.get("GRBM_GUI_ACTIVE"):
        e = {"n_psr": n_psr, "mfma_busy_pct": 100.0 * m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (m["GRBM_GUI_ACTIVE"] * SIMD_PER_XCD),
             "executed_TFLOPs": m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 64 * 2048 / (ms * 1e-3) / 1e12 if ms else None,
             "gui_active_cycles_per_xcd_per_ns": m["GRBM_GUI_ACTIVE"] / 8 / (ms * 1e6) if ms else None,
             "mfma_busy_cycles_per_dispatch": m.get("SQ_VALU_MFMA_BUSY_CYCLES"), "gui_active_cycles_per_dispatch": m["GRBM_GUI_ACTIVE"],
             "insts_valu_per_dispatch": m.get("SQ_INSTS_VALU"), "avg_launch_ms_rocprof": ms, "dispatches": nm, "src_sha": src_sha(*TD_SRC),
             "source": f"profiles/{TAG}_rocprofv3_summary.txt (scripts/gpu_profile_{TAG[:1]}{int(TAG[1:])}.sh)"}
        f, nf = sums("pmc_fetch", k)
        w, nw = sums("pmc_write", k)
        if nf and nw:
            e["fetch_kib_per_dispatch"], e["write_kib_per_dispatch"] = f.get("FETCH_SIZE"), w.get("WRITE_SIZE")
            if ms:
                e["hbm_write_GBps"] = w.get("WRITE_SIZE", 0.0) * 1024.0 / (ms * 1e-3) / 1e9
                e["hbm_fetch_GBps_uncorrected"] = f.get("FETCH_SIZE", 0.0) * 1024.0 / (ms * 1e-3) / 1e9
        big = large_dispatches(k)
        if big:
            e["dispatches_over_1ms"] = big
        res[k] = e
json.dump(res, open(os.path.join(out_dir, f"{TAG}_pmc.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
