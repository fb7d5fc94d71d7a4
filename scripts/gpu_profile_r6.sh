# rocprofv3 evidence for profiles/ (round 6): kernel trace of the bench command, separate PMC passes, and the kernel classes of one ragged factorisation.
# (--pmc passes carry --kernel-trace only: gpurun refuses PMC combined with sys/runtime/hip/hsa tracing.)
cd $GRAFT_REPO_ROOT && export PYTHONUNBUFFERED=1 TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r6; rm -rf $OUT; mkdir -p $OUT
CMD="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o b -- $CMD > $OUT/trace.log 2>&1; echo "trace rc=$?" >> $OUT/trace.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o b -- $CMD > $OUT/pmc_fetch.log 2>&1; echo "rc=$?" >> $OUT/pmc_fetch.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o b -- $CMD > $OUT/pmc_write.log 2>&1; echo "rc=$?" >> $OUT/pmc_write.log
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU --output-format csv -d $OUT/pmc_sq -o b -- $CMD --no-td > $OUT/pmc_sq.log 2>&1; echo "rc=$?" >> $OUT/pmc_sq.log
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_mfma -o b -- $CMD > $OUT/pmc_mfma.log 2>&1; echo "rc=$?" >> $OUT/pmc_mfma.log
# the anisotropic ORF basis of config 5's geometry (VERDICT r4 #5 / missing #5): its own kernel trace
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/orf -o o -- python -c "
import bench
print(bench.orf_numbers(reps=20))" > $OUT/orf.log 2>&1; echo "rc=$?" >> $OUT/orf.log
python scripts/prof_summary.py $OUT > $OUT/r06_rocprofv3_summary.txt 2>&1
python scripts/make_pmc_json.py $OUT 1024 340000 68 r06 > $OUT/make_pmc.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/ragged -o t -- python -c "
import bench
print(bench.td_ragged_numbers(compare_per_matrix=False))" > $OUT/ragged.log 2>&1; echo "rc=$?" >> $OUT/ragged.log
python scripts/potrf_kernel_classes.py $OUT/ragged > $OUT/r06_potrf_ragged_kernel_classes.txt 2>&1
tail -2 $OUT/*.log | head -60
head -30 $OUT/r06_rocprofv3_summary.txt
cat $OUT/r06_potrf_ragged_kernel_classes.txt
find $OUT -name "*.csv" -size +8M -delete
du -sh $OUT
# round 6: kernel classes of ONE uniform factorisation in the default (left-looking) order and in the right-looking one, two chains each
for v in "left:0x200000" "right_la:0x80"; do
  name=${v%%:*}; fl=${v##*:}
  rocprofv3 --kernel-trace --output-format csv -d $OUT/potrf_$name -o t -- python scripts/gpu_r6_potrf_trace.py 68 5000 $fl > $OUT/potrf_$name.log 2>&1; echo "potrf_$name rc=$?"
  echo "## $name (pta_potrf_batched_ws flags $fl), 68 x 5000^2, two chains, under rocprofv3 --kernel-trace" >> $OUT/r06_potrf_uniform_kernel_classes.txt
  python scripts/potrf_schedule_classes.py $OUT/potrf_$name >> $OUT/r06_potrf_uniform_kernel_classes.txt 2>&1
done
cat $OUT/r06_potrf_uniform_kernel_classes.txt
find $OUT -name "*.csv" -size +8M -delete
du -sh $OUT
