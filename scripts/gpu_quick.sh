cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 ) > gpurun_out/pytest_q.log 2>&1; tail -4 gpurun_out/pytest_q.log
for cfg in "2600000 1 auto" "3000000 1 auto" "2048000 1 auto" "10000000 2 auto" "25000000 2 auto" "2600000 2 auto" "4000000 1 auto" "2600000 1 tile"; do
  set -- $cfg
  python bench.py --no-cpu-baseline --fs $1 --sample-size $2 --variant $3 --steps 10 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fs $1 ss $2 variant $3', 'value', j['value'], 'kernel_ms', j['roofline']['kernel_ms'], 'blocks', j['config']['blocks_per_gpu'])"
done > gpurun_out/rates.log 2>&1
cat gpurun_out/rates.log
