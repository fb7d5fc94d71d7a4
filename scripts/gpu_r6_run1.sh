# round 6, first GPU call: the whole -m gpu suite on HEAD (new: left-looking factorisation cases, the 2-rank bench dry run, native ORF pairs),
# smoke, the left-looking experiment, a default bench run
cd $GRAFT_REPO_ROOT && export PYTHONUNBUFFERED=1 TMPDIR=/tmp
O=gpurun_out/r6a; mkdir -p $O
T0=$(date +%s); timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$? in $(( $(date +%s) - T0 )) s" >> $O/pytest.log; tail -15 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python scripts/gpu_r6_potrf_left.py 68 5000 3 10000 16 10000 > $O/potrf_left.jsonl 2> $O/potrf_left.err; echo "left rc=$?"; cat $O/potrf_left.jsonl | cut -c1-260; tail -3 $O/potrf_left.err
T0=$(date +%s); timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$? in $(( $(date +%s) - T0 )) s"; cp gpurun_out/bench_full.json $O/bench_full.json
tail -c 1500 $O/bench.err | tail -5
python - <<'PY'
import json
d = json.load(open('gpurun_out/r6a/bench.json'))
print({k: d[k] for k in ('value', 'ms_per_step')}, len(json.dumps(d)))
print(d['roofline']); print(d['cpu_baseline']); print(d['grid'])
PY
