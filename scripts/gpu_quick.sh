cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) > gpurun_out/pytest_q.log 2>&1; tail -6 gpurun_out/pytest_q.log
for rep in 1 2; do
  for cfg in "2600000 1 16" "2600000 2 16" "2600000 1 8" "10000000 2 16"; do
  set -- $cfg
  python bench.py --no-cpu-baseline --steps 30 --fs $1 --sample-size $2 --nchan $3 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fs $1 ss $2 nchan $3', 'value', j['value'], 'kernel_ms', j['roofline']['kernel_ms'], 'issue frac', j['issue_roofline']['frac'])"
  done
done > gpurun_out/fast2.log 2>&1
cat gpurun_out/fast2.log
