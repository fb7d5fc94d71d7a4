cd $GRAFT_REPO_ROOT && export PYTHONUNBUFFERED=1 TMPDIR=/tmp
O=gpurun_out/r4p; mkdir -p $O
timeout 1500 bash scripts/gpu_profile_r4.sh > $O/profile.log 2>&1; echo "profile rc=$?"
tail -14 $O/profile.log
cp gpurun_out/prof_r4/r04_pmc.json profiles/r04_pmc.json
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r4p/bench.json'))
td=d['td_mode']
print({k:td.get(k) for k in ('potrf_ms','potrf_TFLOPs','generate_td_ms','potrf_trailing_update_mfma_busy_pct','trmm_mfma_busy_pct','cov_assemble_mfma_busy_pct','cov_assemble_GBps_from_WRITE_SIZE')})
print(d['value'], d['roofline'].get('traffic'))
PY
