# round 4, GPU call 1: the whole -m gpu suite (new: ragged Cholesky, config 2 in TD mode), the ragged bench, engine clocks, the (P, N) grid
cd $GRAFT_REPO_ROOT && export PYTHONUNBUFFERED=1 TMPDIR=/tmp
O=gpurun_out/r4a; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -25 $O/pytest.log
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_td.py tests/test_gpu_configs.py -m gpu -q -k "ragged or config2 or td_" > $O/pytest_ragged.log 2>&1; echo "pytest ragged rc=$?" >> $O/pytest_ragged.log
tail -12 $O/pytest_ragged.log
timeout 600 python -c "
import json, bench
print(json.dumps(bench.td_ragged_numbers()))
" > $O/ragged.json 2> $O/ragged.err; echo "ragged rc=$?"; tail -c 1500 $O/ragged.json; tail -c 600 $O/ragged.err
timeout 600 python scripts/gpu_r4_clocks.py > $O/clocks.log 2>&1; echo "clocks rc=$?"; tail -20 $O/clocks.log
timeout 1500 python scripts/gpu_grid_sweep.py --out gpurun_out/r04_grid.json > $O/grid.log 2>&1; echo "grid rc=$?"; tail -30 $O/grid.log
