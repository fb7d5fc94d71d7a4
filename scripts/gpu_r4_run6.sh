# round 4, GPU call 6: wide panels - ragged default (2048), 3072 / 4096, uniform large cells at 2048; suite subset
cd $GRAFT_REPO_ROOT && export PYTHONUNBUFFERED=1 TMPDIR=/tmp
O=gpurun_out/r4f; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_td.py tests/test_gpu_configs.py tests/test_gpu_kernels.py -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
for fl in 0x0 0x20C00 0x21000; do
  PTA_TD_POTRF_FLAGS=$fl python -c "
import json, bench
r = bench.td_ragged_numbers(compare_per_matrix=False)
print(json.dumps({'flags': '$fl', 'potrf_ms': r['potrf_ms'], 'potrf_TFLOPs': r['potrf_TFLOPs']}))" 2>/dev/null | tail -1 | tee -a $O/ragged_flags.jsonl
done
timeout 900 python scripts/gpu_grid_sweep.py --cells 3x35000,16x35000,16x10000 --out gpurun_out/r4f/grid_wide.json > $O/grid_wide.log 2>&1
tail -6 $O/grid_wide.log
