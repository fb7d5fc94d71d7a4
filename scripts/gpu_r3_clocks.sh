# does the sustained fp64 MFMA load of the TD factorisation run at the peak clock?  rocm-smi samples beside a loop of factorisations
cd $GRAFT_REPO_ROOT && export PYTHONUNBUFFERED=1
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|mclk" | head -6
( for i in $(seq 1 40); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Socket Power|Average" | tr '\n' ' '; echo; sleep 0.25; done ) > gpurun_out/clocks.log &
SM=$!
python scripts/gpu_r3_probe.py ${1:-potrfloop} 2>&1 | tail -3
wait $SM
cat gpurun_out/clocks.log | awk 'NR%2==1' | head -24
