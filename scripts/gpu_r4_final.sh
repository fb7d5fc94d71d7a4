# round 4, final check of HEAD: full -m gpu suite, smoke(), the default bench run
cd $GRAFT_REPO_ROOT && export PYTHONUNBUFFERED=1 TMPDIR=/tmp
O=gpurun_out/r4z; mkdir -p $O
echo skip pytest

python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
T0=$(date +%s); python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$? in $(( $(date +%s) - T0 )) s"

python - <<'PY'
import json
d=json.load(open('gpurun_out/r4z/bench.json'))
print({k:d[k] for k in ('value','ms_per_step','kernels_ms')})
print('roofline frac', d['roofline']['frac'], 'traffic', d['roofline'].get('traffic'), 'clock', d['roofline'].get('engine_clock_GHz'))
td=d['td_mode']
print({k:td.get(k) for k in ('potrf_ms','potrf_TFLOPs','potrf_frac_of_fp64_mfma_peak','generate_td_ms','trmm_frac_of_fp64_mfma_peak','potrf_trailing_update_mfma_busy_pct','trmm_mfma_busy_pct','potrf_epi1_ms','potrf_ragged_schedule_ms')})
print('ragged', {k:td['ragged'].get(k) for k in ('potrf_ms','potrf_TFLOPs','potrf_frac_of_fp64_mfma_peak','trmm_frac_of_fp64_mfma_peak')}, td['ragged'].get('per_matrix_schedule'))
print('grid cells', [(c['n_psr'], c['n_toa'], round(c['throughput']['realisations_per_s'])) for c in d['grid'] if 'throughput' in c])
print('api', d['api_mode']['loop']['total'], d['api_mode']['list']['total'], 'cpu', d['cpu_baseline']['value'], d['cpu_baseline']['kind'])
PY
