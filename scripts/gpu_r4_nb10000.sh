cd $GRAFT_REPO_ROOT && export PYTHONUNBUFFERED=1 TMPDIR=/tmp
for fl in 0x0 0x600 0x800; do
  PTA_TD_POTRF_FLAGS=$fl python - <<PY 2>/dev/null | grep -v amdgpu
import bench
for P, N in ((16, 10000), (68, 10000), (68, 7000)):
    d = bench.grid_cell(P, N, td=True)["td"]
    print("$fl", P, N, round(d["potrf_ms"], 2), round(d["potrf_TFLOPs"], 2))
PY
done
