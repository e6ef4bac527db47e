# round 4: soaks at HEAD on the GPU box (the task-based host side, drift enclosure, batched table build, 16-lane apply_patches,
# asynchronous piece submission, ramped pieces)
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 600 python tests/soak_reference.py 480 ) > gpurun_out/r4s_soak_reference.txt 2>&1; tail -1 gpurun_out/r4s_soak_reference.txt
( timeout 300 python tests/soak_fixed.py 180 ) > gpurun_out/r4s_soak_fixed.txt 2>&1; tail -1 gpurun_out/r4s_soak_fixed.txt
( timeout 200 python tests/soak_carrier_walk.py 120 4242 ) > gpurun_out/r4s_soak_host.txt 2>&1; ( timeout 200 python tests/soak_drift.py 4242 90 ) >> gpurun_out/r4s_soak_host.txt 2>&1; cat gpurun_out/r4s_soak_host.txt
( timeout 600 python bench.py --no-cpu-baseline ) > gpurun_out/r4s_bench.json 2> gpurun_out/r4s_bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4s_bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "device_dst_batch", d["extra"]["device_dst_batch"]["value"])
for k, v in d["reference_nco"]["legs"].items():
    print(k, v["value"], v["call_ms"], v["bound"])
PY
