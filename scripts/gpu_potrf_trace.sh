cd $GRAFT_REPO_ROOT && export PYTHONUNBUFFERED=1 TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/potrf_trace; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/look -o t -- python scripts/gpu_potrf_only.py 68 5000 1 > $OUT/look.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/nolook -o t -- python scripts/gpu_potrf_only.py 68 5000 0 > $OUT/nolook.log 2>&1
python scripts/prof_summary.py $OUT/look --timeline > $OUT/summary_look.txt 2>&1
python scripts/prof_summary.py $OUT/nolook --timeline > $OUT/summary_nolook.txt 2>&1
find $OUT -name "*.csv" | head; head -30 $OUT/summary_look.txt; head -12 $OUT/summary_nolook.txt; tail -3 $OUT/look.log
find $OUT -name "*kernel_trace.csv" -size +20M -delete
