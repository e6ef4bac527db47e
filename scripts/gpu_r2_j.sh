# round 2, call j: counters of seg against segb (same instruction count, cheaper VALU forms)
cd /tmp && export TMPDIR=/tmp
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/prof_segb
for v in seg segb; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_segb/$v
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS GRBM_GUI_ACTIVE -d $OUT/a -o pmc -- python $GRAFT_REPO_ROOT/bench.py --variant $v --steps 3 --warmup 1 --launches 6 --no-cpu-baseline --no-extra --rounds 1 > $OUT.a.log 2>&1
  rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_WR -d $OUT/b -o pmc -- python $GRAFT_REPO_ROOT/bench.py --variant $v --steps 3 --warmup 1 --launches 6 --no-cpu-baseline --no-extra --rounds 1 > $OUT.b.log 2>&1
  python - <<PY
import sqlite3, glob
for db in sorted(glob.glob("$OUT/*/*.db")):
    con = sqlite3.connect(db)
    for k, c, n, a, d in con.execute("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection where kernel_name like '%synth_tile%' and grid_size > 4000000 group by kernel_name, counter_name"):
        print("%-5s %-26s n=%-3d avg %16.0f   avg kernel ns %9.0f" % ("$v", c, n, a, d))
PY
done
find $GRAFT_REPO_ROOT/gpurun_out/prof_segb -name "*.db" -delete
