# round 4, GPU call 8: A/B of the tile products' C-tile prefetch epilogue (PTA_POTRF_EPI1)
cd $GRAFT_REPO_ROOT && export PYTHONUNBUFFERED=1 TMPDIR=/tmp
O=gpurun_out/r4h; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_td.py -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
python - > $O/td.json 2> $O/td.err <<'PY'
import json, bench
eng, psrs, noise = bench.build_engine(68, 5000, seed=20260921)
print(json.dumps(bench.td_mode_numbers(eng, 1024)))
PY
python -c "
import json; d=json.load(open('gpurun_out/r4h/td.json')); print({k:v for k,v in d.items() if 'potrf' in k and not isinstance(v,(dict,list,str))})"
tail -c 300 $O/td.err
for fl in 0x0 0x100000 0x0 0x100000; do
  PTA_TD_POTRF_FLAGS=$fl python -c "
import json, bench
r = bench.td_ragged_numbers(compare_per_matrix=False)
print(json.dumps({'flags': '$fl', 'potrf_ms': r['potrf_ms'], 'potrf_TFLOPs': r['potrf_TFLOPs']}))" 2>/dev/null | tail -1 | tee -a $O/ragged_flags.jsonl
done
