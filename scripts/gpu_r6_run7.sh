# round 6: partial strip first in L.z (pta_td_plan.item_rows): TD tests + the TD leg of the bench
cd $GRAFT_REPO_ROOT && export PYTHONUNBUFFERED=1 TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6g; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_td.py tests/test_gpu_configs.py -m gpu -x -q > $O/pytest.log 2>&1; tail -4 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 1200 python bench.py --no-cpu-baseline --no-extras > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.load(open('gpurun_out/r6g/bench.json'))
print({k: v for k, v in d['roofline'].items() if k.startswith('td_')})
print({k: v for k, v in d['roofline_more'].items() if k.startswith('td_')})
PY
rocprofv3 --kernel-trace --output-format csv -d $O/trmm -o t -- python scripts/gpu_r6_trmm_trace.py > $O/trmm.log 2>&1; echo "trace rc=$?"
python scripts/trace_timeline.py $O/trmm 5
find $O -name "*.csv" -size +8M -delete
