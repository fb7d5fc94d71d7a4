# round 4, GPU call 3: suite after the Z-clamp fix, api_mode, ragged-schedule A/B (chains / look-ahead; uniform small-N cells through the ragged path)
cd $GRAFT_REPO_ROOT && export PYTHONUNBUFFERED=1 TMPDIR=/tmp
O=gpurun_out/r4c; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -6 $O/pytest.log
python - > $O/api.json 2> $O/api.err <<'PY'
import json, bench
psrs, noise = bench.headline_array(68, 5000)
print(json.dumps(bench.api_mode_timing(psrs, noise)))
PY
tail -c 600 $O/api.json; tail -c 400 $O/api.err
for fl in 0x10000 0x20000 0x30000 0x40000 0x2 0x20200 0x20300; do
  PTA_TD_POTRF_FLAGS=$fl python -c "
import json, bench
r = bench.td_ragged_numbers(compare_per_matrix=False)
print(json.dumps({'flags': '$fl', 'potrf_ms': r['potrf_ms'], 'potrf_TFLOPs': r['potrf_TFLOPs']}))" 2>/dev/null | tail -1 | tee -a $O/ragged_flags.jsonl
done
PTA_TD_POTRF_MODE=ragged timeout 900 python scripts/gpu_grid_sweep.py --cells 3x122,16x122,68x122,200x122,3x1000,16x1000,68x1000,200x1000,3x5000,16x5000,68x5000,200x5000,3x10000,16x10000 --out gpurun_out/r4c/grid_ragged_mode.json > $O/grid_ragged.log 2>&1
tail -18 $O/grid_ragged.log
