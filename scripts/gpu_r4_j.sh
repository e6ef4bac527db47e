# round 4: reference-NCO batch call after the prefetch / per-context buffers (two default bench runs), reference-mode GPU tests
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_reference_nco.py -m gpu -q -x 2>&1 | tail -3 )
for i in 1 2 3; do
( timeout 900 python bench.py --no-cpu-baseline ) > gpurun_out/r4j_bench_$i.json 2> gpurun_out/r4j_bench.err; tail -1 gpurun_out/r4j_bench.err
python - $i <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r4j_bench_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
for k, v in d["reference_nco"]["legs"].items():
    print(k, v["value"], "call", v["call_ms"], "host", v["host_walk_and_candidates_ms"], "chain", v["host_chain_only_ms"], "eval", v["host_evaluation_only_ms"], v["bound"])
print("e2e reference", d["reference_nco"].get("end_to_end"))
print("device_dst_batch", d["extra"]["device_dst_batch"]["value"], "value", d["value"])
PY
done
