# round 5, GPU call 1: NaN root-cause reproducer, the whole -m gpu suite (new: walk kernel in a NaN slab, async prepare_td stress, enterprise
# adapter), first-call timeline of prepare_td, the bench line, and rocprofv3 passes of the assembly kernels
cd $GRAFT_REPO_ROOT && export PYTHONUNBUFFERED=1 TMPDIR=/tmp
O=gpurun_out/r5a; mkdir -p $O
timeout 300 python scripts/gpu_r5_nan_repro.py > $O/nan_repro.log 2>&1; echo "nan_repro rc=$?" >> $O/nan_repro.log; tail -8 $O/nan_repro.log
T0=$(date +%s); timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$? in $(( $(date +%s) - T0 )) s" >> $O/pytest.log; tail -6 $O/pytest.log
timeout 300 python scripts/gpu_r5_first_call.py > $O/first_call.log 2>&1; echo "rc=$?" >> $O/first_call.log; tail -22 $O/first_call.log
timeout 300 python scripts/gpu_r5_cov_only.py --ragged > $O/cov_only.json 2> $O/cov_only.err; cat $O/cov_only.json
T0=$(date +%s); timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$? in $(( $(date +%s) - T0 )) s"; tail -c 2500 $O/bench.json
P=$GRAFT_REPO_ROOT/$O/prof; rm -rf $P; mkdir -p $P
rocprofv3 --kernel-trace --stats --output-format csv -d $P/trace -o c -- python scripts/gpu_r5_cov_only.py > $P/trace.log 2>&1; echo "trace rc=$?" >> $P/trace.log
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $P/pmc_mfma -o c -- python scripts/gpu_r5_cov_only.py > $P/pmc_mfma.log 2>&1; echo "rc=$?" >> $P/pmc_mfma.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $P/pmc_write -o c -- python scripts/gpu_r5_cov_only.py > $P/pmc_write.log 2>&1; echo "rc=$?" >> $P/pmc_write.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $P/pmc_fetch -o c -- python scripts/gpu_r5_cov_only.py > $P/pmc_fetch.log 2>&1; echo "rc=$?" >> $P/pmc_fetch.log
python - <<'PY'
import csv, glob, collections, os
P = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r5a/prof")
def trace():
    d = collections.defaultdict(list)
    for p in glob.glob(P + "/trace/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(p)):
            if "td_cov" in r["Kernel_Name"]:
                d[r["Kernel_Name"][:60]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    return d
for k, v in trace().items():
    print("trace", k, "n", len(v), "avg ms", sum(v) / len(v), "min", min(v))
for pas in ("pmc_mfma", "pmc_write", "pmc_fetch"):
    acc, n = collections.defaultdict(collections.Counter), collections.defaultdict(set)
    for p in glob.glob(P + f"/{pas}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(p)):
            if "td_cov" in r["Kernel_Name"]:
                k = r["Kernel_Name"][:40]
                acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    for k in acc:
        print(pas, k, "dispatches", len(n[k]), {c: v / len(n[k]) for c, v in acc[k].items()})
PY
find $P -name "*.csv" -size +8M -delete; du -sh $P
