# round 2, call g: the streamed end-to-end leg, the descriptor-swap test, gpsiq_runahead's position forms
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_pipeline.py -m gpu -x -q -k "swap or position or runahead or asynchronous" 2>&1 | tail -15 ) > gpurun_out/pytest_g.log 2>&1
tail -4 gpurun_out/pytest_g.log
( timeout 600 python bench.py --no-cpu-baseline --no-extra --steps 10 ) > gpurun_out/bench_g1.log 2>&1; tail -1 gpurun_out/bench_g1.log | python -c "
import sys, json
d = json.loads(sys.stdin.read()); e = d['end_to_end']; e.pop('what'); e['streamed'].pop('what'); print(d['value'], e)"
for r in 2 4 16; do
( GPSIQ_TRACE=1 timeout 600 python bench.py --no-cpu-baseline --no-extra --steps 5 --rounds $r ) > gpurun_out/bench_g_r$r.log 2>&1; tail -1 gpurun_out/bench_g_r$r.log | python -c "
import sys, json
d = json.loads(sys.stdin.read()); e = d['end_to_end']['streamed']; e.pop('what'); print($r, d['value'], e)"
done
( GPSIQ_BENCH_SHARE_GPU=1 GPSIQ_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 5 --warmup 1 --blocks 1000 --launches 4 ) > gpurun_out/bench_g_2rank.log 2>&1; tail -1 gpurun_out/bench_g_2rank.log | python -c "
import sys, json
d = json.loads(sys.stdin.read()); e = d['end_to_end']; e.pop('what'); e['streamed'].pop('what'); print(d['value'], e)"
