# round 3, fifth GPU session: the generalised wrap-to-wrap walkers -- whole GPU suite, a soak of the reference NCO against the
# reference's own loop, the piece-size A/B, the bench line
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -8 ) > gpurun_out/r3e_pytest_gpu.log 2>&1
tail -4 gpurun_out/r3e_pytest_gpu.log
( timeout 500 python tests/soak_reference.py 300 ) > gpurun_out/r3e_soak_reference.txt 2>&1; tail -3 gpurun_out/r3e_soak_reference.txt
( timeout 200 python tests/soak_carrier_walk.py 120 11 ) > gpurun_out/r3e_soak_carrier_walk.txt 2>&1; tail -2 gpurun_out/r3e_soak_carrier_walk.txt
sed -n '/^cat > \/tmp\/ref_ab.py/,/^PY$/p' scripts/gpu_r3_c.sh | sed '1d;$d' > /tmp/ref_ab.py
( timeout 600 python /tmp/ref_ab.py ) > gpurun_out/r3e_reference_pieces.txt 2>&1; grep -v "descriptors\|candidates" gpurun_out/r3e_reference_pieces.txt
( timeout 900 python bench.py ) > gpurun_out/r3e_bench.json 2> gpurun_out/r3e_bench.err; tail -3 gpurun_out/r3e_bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3e_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["roofline"]["frac"])
for k,v in d["reference_nco"]["legs"].items(): print(k, v["value"], v["call_ms"], v["host_walk_and_candidates_ms"], v["kernel_and_patches_ms"], v["bound"])
for k in ("block_call","block_call_reference_nco","block_call_async","block_call_async_reference_nco"): print(k, d["extra"][k])
PY
