cd $GRAFT_REPO_ROOT && export PYTHONUNBUFFERED=1 TMPDIR=/tmp
mkdir -p gpurun_out/r3c
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r3c/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3c/pytest.log
grep -v "^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl" gpurun_out/r3c/pytest.log | tail -8
timeout 600 python bench.py > gpurun_out/r3c/bench.json 2> gpurun_out/r3c/bench.err; echo "bench rc=$?"
tail -c 400 gpurun_out/r3c/bench.err
python -c "
import json; d=json.load(open('gpurun_out/r3c/bench.json'))
print({k:d[k] for k in ('value','ms_per_step','value_fast_rng_math','value_gwb_grid_draws','value_single_deviate_wn','kernels_ms','api_mode_ms')})
print(d.get('api_mode'))
print({k:v for k,v in d['td_mode'].items() if not isinstance(v,(list,dict))})
"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
