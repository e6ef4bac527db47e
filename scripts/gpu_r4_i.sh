# round 4, ninth GPU session: default bench at HEAD (device_dst_batch after the quantise-first restructure, reference_nco best of 8),
# batch tests
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_nco.py tests/test_gpu_long_runs.py -m gpu -q -x 2>&1 | tail -4 )
for i in 1 2; do
( timeout 900 python bench.py ) > gpurun_out/r4i_bench_$i.json 2> gpurun_out/r4i_bench.err; tail -2 gpurun_out/r4i_bench.err
python - $i <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r4i_bench_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "roofline", d["roofline"]["frac"])
for k, v in d["reference_nco"]["legs"].items():
    print(k, json.dumps({a: b for a, b in v.items() if not isinstance(b, dict)}))
print("e2e", d["end_to_end"]["value"], d["end_to_end"]["streamed"]["value"])
print("device_dst_batch", d["extra"]["device_dst_batch"]["value"], "block_call", d["extra"]["block_call"]["median_us"], d["extra"]["block_call_reference_nco"]["median_us"], d["extra"]["block_call_async"]["us_per_block"], d["extra"]["block_call_async_reference_nco"]["us_per_block"])
PY
done
