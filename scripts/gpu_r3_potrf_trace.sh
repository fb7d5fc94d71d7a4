# per-kernel table + queue-overlap of ONE prepare_td (assembly + workspace-scheme factorisation of 68 x 5000^2)
cd $GRAFT_REPO_ROOT && export PYTHONUNBUFFERED=1 TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/potrf_tl; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/run -o t -- python scripts/gpu_potrf_only.py 68 5000 ${1:-1} ${2:-ws} > $OUT/run.log 2>&1
tail -3 $OUT/run.log
python scripts/prof_summary.py $OUT/run --timeline | head -40
python - <<PY
import csv, glob
rows = list(csv.DictReader(open(glob.glob("$OUT/run/**/*kernel_trace.csv", recursive=True)[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
with open("$GRAFT_REPO_ROOT/gpurun_out/potrf_timeline.csv", "w") as f:
    f.write("queue,kernel,grid_x,grid_y,grid_z,start_us,dur_us\n")
    for r in rows:
        k = r["Kernel_Name"].replace("void ", "")[:18]
        f.write(f'{r.get("Queue_Id","0")},{k},{r["Grid_Size_X"]},{r["Grid_Size_Y"]},{r["Grid_Size_Z"]},{(int(r["Start_Timestamp"])-t0)/1e3:.1f},{(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3:.1f}\n')
PY
rm -rf $OUT
