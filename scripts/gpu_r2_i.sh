# round 2, call i: the both-polarity LUT variant (segb): parity, then A/B against seg at every rate
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "segb or fuzz_long" 2>&1 | tail -5 ) | tee gpurun_out/pytest_segb.log
for cfg in "2.6e6 1" "2.6e6 2" "10e6 2" "25e6 2" "3e6 1"; do
  set -- $cfg
  for v in seg segb seg segb; do
    timeout 300 python bench.py --fs $1 --sample-size $2 --variant $v --steps 10 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('fs $1 ss $2 $v', d['value'], d['roofline']['kernel_ms'])"
  done
done
