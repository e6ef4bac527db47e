cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for e in 0 1 2 3; do
  GPSIQ_MASK_EXP=$e timeout 300 python bench.py --fs 25e6 --sample-size 2 --variant segm --steps 10 --launches 4 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('exp $e', d['value'], d['roofline']['kernel_ms'])"
done
