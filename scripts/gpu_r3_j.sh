# round 3, tenth GPU session: after the tie handling of the walk's table -- reference-NCO tests, a soak, the bench line
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 1800 python -m pytest tests/test_gpu_reference_nco.py tests/test_config4.py tests/test_config35.py tests/test_reference_program.py -m gpu -q 2>&1 | tail -6 ) > gpurun_out/r3j_pytest_gpu.log 2>&1; tail -3 gpurun_out/r3j_pytest_gpu.log
( timeout 400 python tests/soak_reference.py 240 ) > gpurun_out/r3j_soak_reference.txt 2>&1; tail -1 gpurun_out/r3j_soak_reference.txt
( timeout 300 python tests/soak_carrier_walk.py 120 31 ) > gpurun_out/r3j_soak_carrier_walk.txt 2>&1; tail -1 gpurun_out/r3j_soak_carrier_walk.txt
( timeout 900 python bench.py ) > gpurun_out/r3j_bench.json 2> gpurun_out/r3j_bench.err; tail -2 gpurun_out/r3j_bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3j_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["roofline"]["frac"], d["end_to_end"]["streamed"]["value"])
for k,v in d["reference_nco"]["legs"].items(): print(k, v["value"], v["call_ms"], v["host_walk_and_candidates_ms"], v["kernel_and_patches_ms"], v["bound"])
for k in ("block_call","block_call_reference_nco","block_call_async","block_call_async_reference_nco"): print(k, d["extra"][k])
PY
