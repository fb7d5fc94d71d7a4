# round 6: full -m gpu suite with the left-looking default, a default bench run, the timeline of one generate_td(1024)
cd $GRAFT_REPO_ROOT && export PYTHONUNBUFFERED=1 TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6f; rm -rf $O; mkdir -p $O
T0=$(date +%s); timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$? in $(( $(date +%s) - T0 )) s" >> $O/pytest.log; tail -4 $O/pytest.log
rocprofv3 --kernel-trace --output-format csv -d $O/trmm -o t -- python scripts/gpu_r6_trmm_trace.py > $O/trmm.log 2>&1; echo "trace rc=$?"
python scripts/trace_timeline.py $O/trmm 8
T0=$(date +%s); timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$? in $(( $(date +%s) - T0 )) s"; cp gpurun_out/bench_full.json $O/bench_full.json
python - <<'PY'
import json
d = json.load(open('gpurun_out/r6f/bench.json'))
print({k: d[k] for k in ('value', 'ms_per_step')}, len(json.dumps(d)))
print(d['roofline']); print(d['roofline_more'])
PY
find $O -name "*.csv" -size +8M -delete
