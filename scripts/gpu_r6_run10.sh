# round 6: the (N_psr, N_toa) grid with the CPU reference column joined, then BASELINE config 5 in TD mode
cd $GRAFT_REPO_ROOT && export PYTHONUNBUFFERED=1 TMPDIR=/tmp
mkdir -p gpurun_out/r6grid
timeout 2400 python scripts/gpu_grid_sweep.py --out gpurun_out/r6grid/r06_grid.json > gpurun_out/r6grid/grid.log 2>&1; echo "grid rc=$?"
cat gpurun_out/r6grid/r06_grid.txt
timeout 900 python scripts/gpu_config5_td.py > gpurun_out/r6grid/r06_config5_td_mode.json 2> gpurun_out/r6grid/config5.err; echo "config5 rc=$?"; cat gpurun_out/r6grid/r06_config5_td_mode.json
