cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/prof_td && export PYTHONUNBUFFERED=1 TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_td/trace -o td -- python scripts/gpu_td.py 5000 68 512 > gpurun_out/td.log 2>&1
echo "td rc=$?" >> gpurun_out/td.log
grep -E "^\{|rc=" gpurun_out/td.log
