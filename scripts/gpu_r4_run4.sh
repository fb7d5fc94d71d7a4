# round 4, GPU call 4: suite (chunk-overlapped generate_td), the bench line, rocprofv3 passes for profiles/r04_*
cd $GRAFT_REPO_ROOT && export PYTHONUNBUFFERED=1 TMPDIR=/tmp
O=gpurun_out/r4d; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -6 $O/pytest.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
tail -c 600 $O/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r4d/bench.json'))
print({k:d[k] for k in ('value','ms_per_step','kernels_ms')})
td=d['td_mode']
print('td', {k:v for k,v in td.items() if not isinstance(v,(list,dict))})
print('api', d.get('api_mode'))
PY
timeout 1500 bash scripts/gpu_profile_r4.sh > $O/profile.log 2>&1; echo "profile rc=$?"
tail -60 $O/profile.log
