cd $GRAFT_REPO_ROOT
for cfg in "2.6e6 1 12" "2.6e6 1 8" "2.6e6 1 4" "3e6 1 12" "10e6 2 12" "2.6e6 1 16"; do
  set -- $cfg
  for v in seg segb seg segb; do
    timeout 300 python bench.py --fs $1 --sample-size $2 --nchan $3 --variant $v --steps 10 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('fs $1 ss $2 nchan $3 $v', d['value'], d['roofline']['kernel_ms'])"
  done
done
