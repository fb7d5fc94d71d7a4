# MFMA-busy % of the ragged schedule's tile products (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 128 SIMDs per XCD)), dispatches >= 1 ms
cd $GRAFT_REPO_ROOT && export PYTHONUNBUFFERED=1 TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/ragged_pmc; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/run -o r -- python -c "
import bench
print(bench.td_ragged_numbers(compare_per_matrix=False))" > $OUT/run.log 2>&1; echo "rc=$?" >> $OUT/run.log
tail -2 $OUT/run.log | cut -c1-300
python - <<'PY' | tee gpurun_out/r04_ragged_pmc.txt
import collections, csv, glob
dur = {}
for p in glob.glob("gpurun_out/ragged_pmc/run/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        dur[r["Dispatch_Id"]] = ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, r["Kernel_Name"])
acc = collections.defaultdict(lambda: collections.Counter())
for p in glob.glob("gpurun_out/ragged_pmc/run/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        d = dur.get(r["Dispatch_Id"])
        if d is None:
            continue
        k = "k_dgemm_glds128<true>" if "k_dgemm_glds128<true" in d[1] else ("k_td_trmm_rng" if "k_td_trmm_rng" in d[1] else ("k_td_cov128" if "k_td_cov128" in d[1] else None))
        if k is None or d[0] < 1.0:
            continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        acc[k]["_n_" + r["Counter_Name"]] += 1
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            acc[k]["_ms"] += d[0]
print("# ragged ng15-like array (bench.td_ragged_numbers) under rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES: dispatches >= 1 ms")
for k, c in acc.items():
    busy = 100.0 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] * 128)
    print(f"{k:24s} dispatches {int(c['_n_GRBM_GUI_ACTIVE']):4d}  total {c['_ms']:9.1f} ms  MFMA-busy {busy:5.1f} %")
PY
