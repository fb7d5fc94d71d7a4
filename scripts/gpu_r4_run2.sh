# round 4, GPU call 2: full -m gpu suite on the rebuilt library + the bench line with its new keys (clocks, td_mode.ragged, grid)
cd $GRAFT_REPO_ROOT && export PYTHONUNBUFFERED=1 TMPDIR=/tmp
O=gpurun_out/r4b; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -8 $O/pytest.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
tail -c 800 $O/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r4b/bench.json'))
print({k:d[k] for k in ('value','ms_per_step','value_fast_rng_math','value_gwb_grid_draws','value_single_deviate_wn','kernels_ms')})
print('roofline', {k:v for k,v in d['roofline'].items() if k not in ('also','alternative_gwb_transform')})
print('step', d.get('step'))
print('clocks', d.get('engine_clocks'))
td=d['td_mode']
print('td', {k:v for k,v in td.items() if not isinstance(v,(list,dict))})
print('ragged', td.get('ragged'))
for c in d.get('grid',[]): print('cell', json.dumps(c)[:600])
print('api', d.get('api_mode'))
print('cpu', d.get('cpu_baseline',{}).get('value'), d.get('cpu_baseline',{}).get('kind'))
PY
