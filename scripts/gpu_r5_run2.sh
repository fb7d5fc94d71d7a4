# round 5, GPU call 2: the 128-column / two-steps-ahead form of the walking assembly kernel, L.z with the all-zero tile products of the
# diagonal slabs skipped; TD tests, A/B timings, bench, MFMA-busy / WRITE_SIZE counters of the assembly kernels
cd $GRAFT_REPO_ROOT && export PYTHONUNBUFFERED=1 TMPDIR=/tmp
O=gpurun_out/r5b; mkdir -p $O
T0=$(date +%s); timeout 1200 python -m pytest tests/test_gpu_td.py tests/test_gpu_configs.py tests/test_gpu_kernels.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$? in $(( $(date +%s) - T0 )) s" >> $O/pytest.log; tail -6 $O/pytest.log
timeout 300 python scripts/gpu_r5_cov_only.py --ragged > $O/cov_only.json 2> $O/cov_only.err; cat $O/cov_only.json; tail -3 $O/cov_only.err
T0=$(date +%s); timeout 900 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$? in $(( $(date +%s) - T0 )) s"; tail -c 1800 $O/bench.json
P=$GRAFT_REPO_ROOT/$O/prof; rm -rf $P; mkdir -p $P
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $P/pmc_mfma -o c -- python scripts/gpu_r5_cov_only.py > $P/pmc_mfma.log 2>&1; echo "rc=$?" >> $P/pmc_mfma.log
python - <<'PY'
import csv, glob, collections, os
P = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r5b/prof")
for pas in ("pmc_mfma",):
    acc, n = collections.defaultdict(collections.Counter), collections.defaultdict(set)
    for p in glob.glob(P + f"/{pas}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(p)):
            if "td_cov" in r["Kernel_Name"]:
                k = r["Kernel_Name"][:44]
                acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    for k in acc:
        d = {c: v / len(n[k]) for c, v in acc[k].items()}
        print(pas, k, "dispatches", len(n[k]), "mfma busy %.1f %%" % (100 * d["SQ_VALU_MFMA_BUSY_CYCLES"] / (d["GRBM_GUI_ACTIVE"] * 128)), d)
PY
find $P -name "*.csv" -size +8M -delete
