# round 3, last check after the kLow change (host-only code): the reference-NCO GPU tests (without the 45 GB run) and the bench line
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_reference_nco.py tests/test_config4.py tests/test_reference_program.py -m gpu -q 2>&1 | tail -4 ) > gpurun_out/r3p_pytest.log 2>&1; tail -3 gpurun_out/r3p_pytest.log
( timeout 900 python bench.py ) > gpurun_out/r3p_bench.json 2> gpurun_out/r3p_bench.err; tail -2 gpurun_out/r3p_bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3p_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["roofline"]["frac"], d["end_to_end"]["streamed"]["value"], d["extra"]["device_dst_batch"]["value"])
for k,v in d["reference_nco"]["legs"].items(): print(k, v["value"], v["call_ms"], v["host_walk_and_candidates_ms"], v["kernel_and_patches_ms"], v["bound"])
for k in ("block_call_reference_nco","block_call_async_reference_nco"): print(k, d["extra"][k])
PY
