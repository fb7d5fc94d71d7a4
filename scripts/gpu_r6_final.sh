# round 6, final check of HEAD: the whole -m gpu suite, smoke(), the default bench run (the line + the full record)
cd $GRAFT_REPO_ROOT && export PYTHONUNBUFFERED=1 TMPDIR=/tmp
O=gpurun_out/r6z; mkdir -p $O
T0=$(date +%s); timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$? in $(( $(date +%s) - T0 )) s" >> $O/pytest.log; grep -n "passed\|failed\|rc=" $O/pytest.log | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
T0=$(date +%s); python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$? in $(( $(date +%s) - T0 )) s"; cp gpurun_out/bench_full.json $O/bench_full.json
python - <<'PY'
import json
d = json.load(open('gpurun_out/r6z/bench.json'))
print({k: d[k] for k in ('value', 'ms_per_step')}, len(json.dumps(d)))
print(d['roofline']); print(d['cpu_baseline']); print(d['grid'])
PY
