cd $GRAFT_REPO_ROOT && export PYTHONUNBUFFERED=1 TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6i; rm -rf $O; mkdir -p $O
for v in fused twostep; do
  rocprofv3 --kernel-trace --output-format csv -d $O/$v -o t -- python scripts/gpu_r6_fused_trace.py $v 1 > $O/$v.log 2>&1; echo "$v rc=$?"
  python - <<PY
import csv, glob
rows = []
for p in glob.glob("$O/$v/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(p)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last run: from the last k_td_cov_walk (twostep) / the last-but-4 k_td_fused_update (fused) on
big = [r for r in rows if any(k in r["Kernel_Name"] for k in ("k_td_fused_update", "k_td_cov_walk")) or ("k_dgemm_glds128" in r["Kernel_Name"] and int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 1500000)]
for r in big[-6:]:
    gx = int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])
    print("$v", r["Kernel_Name"][:40], f"{gx}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']}", round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, 3), "ms")
PY
done
find $O -name "*.csv" -size +8M -delete
