cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out && export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --tb=short -p no:cacheprovider 2>&1 | tail -4 > gpurun_out/pytest_gpu.log
timeout 900 python scripts/gpu_td.py 5000 16 512 > gpurun_out/td.log 2>&1
echo "td rc=$?" >> gpurun_out/td.log
cat gpurun_out/pytest_gpu.log; grep -E "^\{|rc=" gpurun_out/td.log
