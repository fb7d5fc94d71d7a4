cd $GRAFT_REPO_ROOT && export PYTHONUNBUFFERED=1 TMPDIR=/tmp
mkdir -p gpurun_out/r3a
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3a/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3a/pytest.log
tail -15 gpurun_out/r3a/pytest.log
timeout 600 python scripts/gpu_r3_probe.py > gpurun_out/r3a/probe.log 2>&1; echo "probe rc=$?" >> gpurun_out/r3a/probe.log
cat gpurun_out/r3a/probe.log | tail -30
timeout 600 python bench.py > gpurun_out/r3a/bench.json 2> gpurun_out/r3a/bench.err; echo "bench rc=$?"
tail -c 600 gpurun_out/r3a/bench.err
python -c "
import json; d=json.load(open('gpurun_out/r3a/bench.json'))
print({k:d[k] for k in ('value','ms_per_step','value_fast_rng_math','value_gwb_grid_draws','kernels_ms')})
print(d.get('step')); print(d.get('config4_shape')); print(d['roofline']['also'].get('frac'), d['roofline']['also'].get('transform_kernel_alone'))
print({k:v for k,v in d['td_mode'].items() if not isinstance(v,(list,dict))})
print(d['cpu_baseline'].get('value'), d['cpu_baseline'].get('kind'), 'refcont' , 'reference_container' in d['cpu_baseline'])
"
