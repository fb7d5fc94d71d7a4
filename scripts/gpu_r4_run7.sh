# round 4, GPU call 7: final suite, rocprofv3 passes for profiles/r04_* on the final kernels, then the bench line (reads the fresh PMC file)
cd $GRAFT_REPO_ROOT && export PYTHONUNBUFFERED=1 TMPDIR=/tmp
O=gpurun_out/r4g; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
timeout 1500 bash scripts/gpu_profile_r4.sh > $O/profile.log 2>&1; echo "profile rc=$?"
tail -12 $O/profile.log
cp gpurun_out/prof_r4/r04_pmc.json profiles/r04_pmc.json
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
tail -c 300 $O/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r4g/bench.json'))
print({k:d[k] for k in ('value','ms_per_step','kernels_ms')})
print('roofline', {k:v for k,v in d['roofline'].items() if k not in ('also','alternative_gwb_transform')})
td=d['td_mode']
print('td', {k:v for k,v in td.items() if not isinstance(v,(list,dict))})
print('ragged', td.get('ragged'))
PY
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
