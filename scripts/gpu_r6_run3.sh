# round 6: the left-looking experiment again with the trapezoid tile grid (XCD-balanced block-column updates), + potrf tests
cd $GRAFT_REPO_ROOT && export PYTHONUNBUFFERED=1 TMPDIR=/tmp
O=gpurun_out/r6c; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "potrf or dgemm or gemm" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 900 python scripts/gpu_r6_potrf_left.py 68 5000 16 10000 > $O/potrf_left.jsonl 2> $O/potrf_left.err; echo "left rc=$?"
python - <<'PY'
import json
for ln in open('gpurun_out/r6c/potrf_left.jsonl'):
    d = json.loads(ln)
    print(f"{d['P']:4d} {d['N']:6d} {d['variant']:28s} {d['ms']:8.2f} ms {d['TFLOPs']:6.2f} TF {d['frac']:.4f}  diff {d['max_rel_diff_vs_right']:.1e} det {d['bit_identical_runs']}")
PY
tail -3 $O/potrf_left.err
