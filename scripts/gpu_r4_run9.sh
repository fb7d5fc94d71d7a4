# round 4, GPU call 9: the column-walking covariance assembly kernel - parity with the tile kernel, A/B timing
cd $GRAFT_REPO_ROOT && export PYTHONUNBUFFERED=1 TMPDIR=/tmp
O=gpurun_out/r4i; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_td.py tests/test_gpu_kernels.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -12 $O/pytest.log
python - > $O/td.json 2> $O/td.err <<'PY'
import json, bench
eng, psrs, noise = bench.build_engine(68, 5000, seed=20260921)
print(json.dumps(bench.td_mode_numbers(eng, 1024)))
PY
python -c "
import json; d=json.load(open('gpurun_out/r4i/td.json')); print(d.get('cov_assemble_kernels'), d.get('cov_assemble_ms'))"
tail -c 300 $O/td.err
