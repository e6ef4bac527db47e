cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -5 ) > gpurun_out/pytest_q.log 2>&1; tail -3 gpurun_out/pytest_q.log
for nb in 4130 4096 4224; do
 for tail in 0 512 1024 2048 4096; do
  for minw in 8192 4096; do
  GPSIQ_SEG_TAIL_WGS=$tail GPSIQ_SEG_MIN_WGS=$minw python bench.py --no-cpu-baseline --blocks $nb --steps 20 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('blocks', $nb, 'tail', $tail, 'minwgs', $minw, 'value', j['value'], 'kernel_ms', j['roofline']['kernel_ms'])"
  done
 done
done > gpurun_out/tail.log 2>&1
cat gpurun_out/tail.log
