cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out && export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout 900 > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 900 python bench.py --steps 5 --warmup 2 > gpurun_out/bench.log 2>&1
echo "bench rc=$?" >> gpurun_out/bench.log
tail -25 gpurun_out/pytest_gpu.log; tail -3 gpurun_out/bench.log
