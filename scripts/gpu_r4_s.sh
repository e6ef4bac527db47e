# round 4: the synchronous block call as the asynchronous form + a wait for that block (one host round trip instead of two)
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_config5_shares.py::test_all_eight_shares_of_the_whole_run_each_alone --deselect "tests/test_config35.py::test_reference_thread_on_the_gpu_at_full_length[cfg5]" 2>&1 | tail -4 )
( timeout 900 python bench.py --no-cpu-baseline ) > gpurun_out/r4s_bench.json 2> gpurun_out/r4s_bench.err; tail -2 gpurun_out/r4s_bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4s_bench.json").read().strip().splitlines()[-1])
e = d["extra"]
for k in ("block_call", "block_call_reference_nco", "block_call_async", "host_dst_batch", "device_dst_batch"):
    print(k, json.dumps({a: b for a, b in e[k].items() if a != "what"}))
print("value", d["value"])
PY
