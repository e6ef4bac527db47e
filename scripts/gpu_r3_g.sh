# round 3, seventh GPU session: BASELINE configs 3 and 5 (first GPU's share) against the reference program's bytes, then the whole suite
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
df -h /tmp | tail -1; free -g | head -2
( timeout 2400 python -m pytest tests/test_config35.py -m gpu -q --durations=5 2>&1 | tail -15 ) > gpurun_out/r3g_config35.log 2>&1; tail -12 gpurun_out/r3g_config35.log
( timeout 2400 python -m pytest tests -m gpu -q --durations=8 2>&1 | tail -16 ) > gpurun_out/r3g_pytest_gpu.log 2>&1; tail -12 gpurun_out/r3g_pytest_gpu.log
( timeout 900 python bench.py ) > gpurun_out/r3g_bench.json 2> gpurun_out/r3g_bench.err; tail -2 gpurun_out/r3g_bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3g_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["roofline"]["frac"], d["end_to_end"]["streamed"]["value"])
for k,v in d["reference_nco"]["legs"].items(): print(k, v["value"], v["call_ms"], v["host_walk_and_candidates_ms"], v["kernel_and_patches_ms"], v["bound"])
for k in ("block_call","block_call_reference_nco","block_call_async","block_call_async_reference_nco"): print(k, d["extra"][k])
PY
