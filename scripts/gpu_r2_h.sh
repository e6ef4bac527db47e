# round 2, call h: worker-pool spin phase A/B on the streamed and serial end-to-end legs
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for us in 0 60 0 60 200; do
GPSIQ_SPIN_US=$us timeout 600 python bench.py --no-cpu-baseline --no-extra --steps 5 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); e = d['end_to_end']; s = e['streamed']
print('spin_us $us', 'serial', e['value'], e['host_refresh_ms'], e['quantise_and_seed_exchange_ms'], e['validate_upload_ms'], 'streamed', s['value'], s['host_refresh_and_quantise_ms_per_round'])"
done
