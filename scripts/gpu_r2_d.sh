set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "segm" 2>&1 | tail -15 ) > gpurun_out/pytest_segm.log 2>&1; tail -5 gpurun_out/pytest_segm.log
for cfg in "25e6 2" "10e6 2" "2.6e6 2" "25e6 1"; do set -- $cfg
  for v in seg segm; do
    timeout 300 python bench.py --fs $1 --sample-size $2 --variant $v --steps 10 --launches 4 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$1 ss$2 $v', d['value'], d['roofline']['frac'], d['roofline']['kernel_ms'])"
  done
done
