# round 6: ragged left-looking A/B on the ng15-like array and config 2's shape
cd $GRAFT_REPO_ROOT && export PYTHONUNBUFFERED=1 TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6k; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "ragged or potrf" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for fl in 0x0 0x200000 0x200400 0x200800 0x400 0x800; do
  PTA_TD_POTRF_FLAGS=$fl python - <<PY 2>&1 | tail -2
import bench_extras as b, json
r = b.td_ragged_numbers(compare_per_matrix=False)
print("ng15-like flags $fl", {k: round(r[k], 3) for k in ("potrf_ms", "potrf_TFLOPs", "potrf_frac_of_fp64_mfma_peak", "trmm_frac_of_fp64_mfma_peak", "realisations_per_s")})
r = b.td_ragged_numbers(R=256, compare_per_matrix=False, counts=(7758, 23023, 35037))
print("config2    flags $fl", {k: round(r[k], 3) for k in ("potrf_ms", "potrf_TFLOPs", "potrf_frac_of_fp64_mfma_peak", "trmm_frac_of_fp64_mfma_peak", "realisations_per_s")})
PY
done
