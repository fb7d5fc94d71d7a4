# small batches: does a narrower panel help when the batch cannot fill the chip?  (uniform schedule, PTA_POTRF_NB(k) = 256 k columns)
cd $GRAFT_REPO_ROOT && export PYTHONUNBUFFERED=1 TMPDIR=/tmp
for fl in 0x0 0x100 0x200; do
  PTA_TD_POTRF_FLAGS=$fl python - <<PY 2>/dev/null | grep -v amdgpu
import json, bench
for P, N in ((3, 5000), (3, 10000), (16, 5000), (16, 1000), (68, 1000), (200, 1000)):
    c = bench.grid_cell(P, N, td=True)
    d = c["td"]
    print("$fl", P, N, round(d["potrf_ms"], 3), round(d["potrf_TFLOPs"], 2))
PY
done
