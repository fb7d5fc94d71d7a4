cd $GRAFT_REPO_ROOT && export PYTHONUNBUFFERED=1 TMPDIR=/tmp
mkdir -p gpurun_out/r3b
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3b/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3b/pytest.log
tail -12 gpurun_out/r3b/pytest.log
timeout 600 python scripts/gpu_r3_probe.py potrf > gpurun_out/r3b/probe.log 2>&1; echo "probe rc=$?" >> gpurun_out/r3b/probe.log
tail -12 gpurun_out/r3b/probe.log
