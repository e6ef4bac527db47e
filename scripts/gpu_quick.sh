cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( GPSIQ_SEG_MAX_CHUNKS=8 GPSIQ_SEG_MIN_WGS=1 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "seg or sharding or baseline" 2>&1 | tail -5 ) > gpurun_out/pytest_q.log 2>&1; tail -3 gpurun_out/pytest_q.log
for nb in 4130 4224; do
 for cfg in "4 8192 512" "8 4096 512" "8 4096 1024" "8 4096 2048" "8 2048 1024" "16 2048 1024"; do
  set -- $cfg
  GPSIQ_SEG_MAX_CHUNKS=$1 GPSIQ_SEG_MIN_WGS=$2 GPSIQ_SEG_TAIL_WGS=$3 python bench.py --no-cpu-baseline --blocks $nb --steps 20 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('blocks', $nb, 'maxchunks $1 minwgs $2 tail $3', 'value', j['value'], 'kernel_ms', j['roofline']['kernel_ms'])"
 done
done > gpurun_out/tail.log 2>&1
cat gpurun_out/tail.log
