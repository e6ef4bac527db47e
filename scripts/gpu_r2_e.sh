set -x
REPO=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
mkdir -p $REPO/gpurun_out/prof_segm
rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_segm/kt -o kt -- python $REPO/bench.py --fs 25e6 --sample-size 2 --variant segm --steps 5 --launches 4 --no-cpu-baseline --no-extra > $REPO/gpurun_out/prof_segm/kt.log 2>&1
cd $REPO && python - <<PY
import sqlite3,glob
db=glob.glob("gpurun_out/prof_segm/kt/*.db")[0]
con=sqlite3.connect(db)
for r in con.execute("select name,total_calls,total_duration,average,percentage from top_kernels"): print(r[0][:60],r[1:])
PY
find $REPO/gpurun_out/prof_segm -name "*.db" -delete
