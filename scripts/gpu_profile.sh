# rocprofv3 evidence for profiles/: kernel-trace stats of the bench command, then separate PMC passes (HBM bytes)
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/prof && export PYTHONUNBUFFERED=1 TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof
rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_under_rocprof.log 2>&1
echo "trace rc=$?" >> $OUT/bench_under_rocprof.log
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/pmc_fetch.log 2>&1
echo "fetch rc=$?" >> $OUT/pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/pmc_write.log 2>&1
echo "write rc=$?" >> $OUT/pmc_write.log
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU -d $OUT/pmc_mfma -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/pmc_mfma.log 2>&1
echo "mfma rc=$?" >> $OUT/pmc_mfma.log
find $OUT -type f | head -50; du -sh $OUT
python scripts/summarize_prof.py $OUT > $OUT/summary.txt 2>&1; cat $OUT/summary.txt | head -80
