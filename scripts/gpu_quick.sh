cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out && export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider --timeout 600 2>&1 | tail -5 > gpurun_out/pytest_gpu.log
timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench.log 2>&1
cat gpurun_out/pytest_gpu.log; tail -1 gpurun_out/bench.log
