# round 4: the default bench at the round's HEAD, as the driver runs it
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python bench.py ) > gpurun_out/r4q_bench.json 2> gpurun_out/r4q_bench.err; tail -2 gpurun_out/r4q_bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4q_bench.json").read().strip().splitlines()[-1])
for k, v in d["reference_nco"]["legs"].items():
    print(k, v["value"], "call", v["call_ms"], "host", v["host_walk_and_candidates_ms"], "chain", v["host_chain_only_ms"], "eval", v["host_evaluation_only_ms"], v["bound"])
s = d["end_to_end"]["streamed"]
print("value", d["value"], "roofline", d["roofline"]["frac"], "streamed", s["value"], s["seconds_each_pass"], s["per_rank"])
print("e2e", d["end_to_end"]["value"], "e2e reference", d["reference_nco"]["end_to_end"]["value"], "device_dst_batch", d["extra"]["device_dst_batch"]["value"])
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["all_cores"]["value"], "rccl", json.dumps(d["extra"]["rccl_selftest"])[:300])
PY
