# round 6: kernel traces again (balanced trapezoid grid, equal-priority side streams): right-looking look-ahead, left-looking 2 chains / 1 chain
cd $GRAFT_REPO_ROOT && export PYTHONUNBUFFERED=1 TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6e; rm -rf $O; mkdir -p $O
for v in "right_la:0x80" "left:0x200000" "left_1chain:0x210000" "left_diagla:0x200080"; do
  name=${v%%:*}; fl=${v##*:}
  rocprofv3 --kernel-trace --output-format csv -d $O/$name -o t -- python scripts/gpu_r6_potrf_trace.py 68 5000 $fl > $O/$name.log 2>&1; echo "$name rc=$?"
  echo "## $name (pta_potrf_batched_ws flags $fl), 68 x 5000^2, under rocprofv3 --kernel-trace" >> $O/classes.txt
  python scripts/potrf_schedule_classes.py $O/$name >> $O/classes.txt 2>&1
  echo >> $O/classes.txt
done
cat $O/classes.txt
find $O -name "*.csv" -size +8M -delete
