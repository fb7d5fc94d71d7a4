cd $GRAFT_REPO_ROOT && export PYTHONUNBUFFERED=1 TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/potrf_trace; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/nolook -o t -- python scripts/gpu_potrf_only.py 68 5000 0 > $OUT/nolook.log 2>&1
python scripts/prof_summary.py $OUT/nolook --timeline > $OUT/summary_nolook.txt 2>&1
python - <<PY
import csv, collections, glob
rows = list(csv.DictReader(open(glob.glob("$OUT/nolook/**/*kernel_trace.csv", recursive=True)[0])))
agg = collections.OrderedDict()
for r in rows:
    if "dgemm" in r["Kernel_Name"]:
        key = (r["Kernel_Name"][:22], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])
        a = agg.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
for k, (n, t) in agg.items(): print(k, n, round(t, 3))
PY
head -12 $OUT/summary_nolook.txt
find $OUT -name "*kernel_trace.csv" -size +20M -delete
