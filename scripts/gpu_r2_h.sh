# round 2, call h: end-to-end legs with the fused refresh+quantise; full GPU suite
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2 3; do
timeout 600 python bench.py --no-cpu-baseline --no-extra --steps 5 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); e = d['end_to_end']; s = e['streamed']
print('serial', e['value'], e['host_refresh_and_quantise_ms'], e['seed_exchange_ms'], e['validate_upload_ms'], e['kernel_ms'], 'streamed', s['value'], s['host_refresh_and_quantise_ms_per_round'])"
done
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 ) > gpurun_out/pytest_gpu.log 2>&1
tail -3 gpurun_out/pytest_gpu.log
