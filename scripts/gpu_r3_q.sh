# round 3, closing check at HEAD inside the last GPU minutes: the whole GPU suite, smoke, the default bench line
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 250 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 ) > gpurun_out/r3q_pytest_gpu.log 2>&1; tail -2 gpurun_out/r3q_pytest_gpu.log
( timeout 60 python __graft_entry__.py smoke ) > gpurun_out/r3q_smoke.log 2>&1; tail -1 gpurun_out/r3q_smoke.log
( time timeout 200 python bench.py ) > gpurun_out/r3q_bench.json 2> gpurun_out/r3q_bench.err; tail -4 gpurun_out/r3q_bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3q_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["roofline"]["frac"], d["end_to_end"]["bound"], d["end_to_end"]["streamed"]["value"])
for k,v in d["reference_nco"]["legs"].items(): print(k, v["value"], v["bound"])
PY
