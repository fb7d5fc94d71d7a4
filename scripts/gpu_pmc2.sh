# stall analysis of the two hot kernels: separate PMC passes (SQ block: 8 counters per pass)
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/pmc2 && export PYTHONUNBUFFERED=1 TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc2
CMD="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-td"
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU -d $OUT/p1 -o bench -- $CMD > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU -d $OUT/p2 -o bench -- $CMD > $OUT/p2.log 2>&1
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum -d $OUT/p3 -o bench -- $CMD > $OUT/p3.log 2>&1
python scripts/summarize_prof.py $OUT > $OUT/summary.txt 2>&1; grep -E "synth_mfma<false>|k_gwb_czt<true, false" $OUT/summary.txt
