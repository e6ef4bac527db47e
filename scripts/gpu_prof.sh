# rocprofv3 passes for profiles/: kernel-trace stats, then PMC passes (each alone).
set -x
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --no-cpu-baseline --no-extra"   # the default workload (--gpus 1 --steps 20 --warmup 3, 18 launches per step) without the extra legs
rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- $BENCH > $OUT/kt.log 2>&1
tail -3 $OUT/kt.log
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc -- $BENCH > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc -- $BENCH > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/pmc_sq1 -o pmc -- $BENCH > $OUT/pmc_sq1.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_WR -d $OUT/pmc_sq2 -o pmc -- $BENCH > $OUT/pmc_sq2.log 2>&1
find $OUT -type f | head -50
du -sh $OUT
# summarise on the box too (the .db files can be large; only the text summaries are needed)
( cd $REPO && python scripts/prof_summary.py gpurun_out/prof ${PROF_TAG:-r02} > $OUT/summary.log 2>&1; tail -3 $OUT/summary.log )
find $OUT -name "*.db" -size +8M -delete
mkdir -p $REPO/gpurun_out/profiles_out; cp $REPO/profiles/${PROF_TAG:-r02}_* $REPO/profiles/pmc_traffic.json $REPO/profiles/pmc_counters.json $REPO/gpurun_out/profiles_out/
