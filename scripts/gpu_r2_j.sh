cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_segb
for b in 4096 4130; do
for e in "" 1; do for v in seg segb; do
[ "$v" = seg ] && [ "$e" = 1 ] && continue
GPSIQ_SEGB_EXP=$e timeout 300 python bench.py --variant $v --blocks $b --steps 10 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('blocks $b exp [$e] $v', d['value'], d['roofline']['kernel_ms'])"
done; done; done
cd /tmp && export TMPDIR=/tmp
for cfg in "seg 0" "segb 0" "segb 1"; do
  set -- $cfg
  OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_segb/$1_$2
  GPSIQ_SEGB_EXP=$2 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS GRBM_GUI_ACTIVE -d $OUT/a -o pmc -- python $GRAFT_REPO_ROOT/bench.py --variant $1 --steps 3 --warmup 1 --launches 4 --no-cpu-baseline --no-extra > $OUT.a.log 2>&1
  GPSIQ_SEGB_EXP=$2 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS -d $OUT/b -o pmc -- python $GRAFT_REPO_ROOT/bench.py --variant $1 --steps 3 --warmup 1 --launches 4 --no-cpu-baseline --no-extra > $OUT.b.log 2>&1
  python - <<PY
import sqlite3, glob
for db in sorted(glob.glob("$OUT/*/*.db")):
    con = sqlite3.connect(db)
    for k, c, n, a in con.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection where kernel_name like '%synth_tile%' group by kernel_name, counter_name"):
        print("$1 exp$2", c, n, round(a))
PY
done
find $GRAFT_REPO_ROOT/gpurun_out/prof_segb -name "*.db" -delete
