cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out && export PYTHONUNBUFFERED=1
timeout 900 python scripts/gpu_perf.py > gpurun_out/perf.log 2>&1
echo "perf rc=$?" >> gpurun_out/perf.log
cat gpurun_out/perf.log
