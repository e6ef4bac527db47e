# round 3, fourth GPU session: the one-pass reference walker -- the reference-NCO GPU tests, config 4, the piece-size A/B, the bench line
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 1800 python -m pytest tests/test_gpu_reference_nco.py tests/test_config4.py tests/test_reference_program.py tests/test_gpu_long_runs.py tests/test_host_c.py -m gpu -q 2>&1 | tail -15 ) > gpurun_out/r3d_pytest_gpu.log 2>&1
tail -6 gpurun_out/r3d_pytest_gpu.log
sed -n '/^cat > \/tmp\/ref_ab.py/,/^PY$/p' scripts/gpu_r3_c.sh | sed '1d;$d' > /tmp/ref_ab.py
( timeout 600 python /tmp/ref_ab.py ) > gpurun_out/r3d_reference_pieces.txt 2>&1; grep -v "descriptors\|candidates" gpurun_out/r3d_reference_pieces.txt
( timeout 900 python bench.py ) > gpurun_out/r3d_bench.json 2> gpurun_out/r3d_bench.err; tail -3 gpurun_out/r3d_bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3d_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["roofline"]["frac"], d.get("counters"))
for k,v in d["reference_nco"]["legs"].items(): print(k, v["value"], v["call_ms"], v["host_walk_and_candidates_ms"], v["kernel_and_patches_ms"], v["bound"])
for k in ("block_call","block_call_reference_nco","block_call_async","block_call_async_reference_nco"): print(k, d["extra"][k])
PY
