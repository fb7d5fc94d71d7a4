cd $GRAFT_REPO_ROOT && export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_td.py -m gpu -x -q 2>&1 | tail -4
python scripts/gpu_r5_cov_only.py --ragged 2>/dev/null | tail -1
PTA_REPLICATOR_AMD_LIB=$GRAFT_REPO_ROOT/scripts/probe_src/libpta_tcw_diag.so python scripts/gpu_r5_tcw_diag.py 2>&1 | tail -1
