# round 6: left-looking + diagonal-phase look-ahead, high-priority side streams (A/B by PTA_POTRF_SIDE_PRIO)
cd $GRAFT_REPO_ROOT && export PYTHONUNBUFFERED=1 TMPDIR=/tmp
O=gpurun_out/r6d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "potrf" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for pr in 1 0; do
  PTA_POTRF_SIDE_PRIO=$pr timeout 900 python scripts/gpu_r6_potrf_left.py 68 5000 16 10000 3 10000 > $O/potrf_left_prio$pr.jsonl 2> $O/potrf_left_prio$pr.err; echo "prio $pr rc=$?"
  python - <<PY
import json
for ln in open('gpurun_out/r6d/potrf_left_prio$pr.jsonl'):
    d = json.loads(ln)
    print(f"prio$pr {d['P']:4d} {d['N']:6d} {d['variant']:28s} {d['ms']:8.2f} ms {d['TFLOPs']:6.2f} TF {d['frac']:.4f}  diff {d['max_rel_diff_vs_right']:.1e} det {d['bit_identical_runs']}")
PY
done
