# round 4: the whole GPU suite with the full-hour config 5 fixture, smoke, the default bench as the driver runs it, a second bench
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -q --durations=8 2>&1 | tail -16 ) | tee gpurun_out/r4l_pytest.txt
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 )
( timeout 900 python bench.py ) > gpurun_out/r4l_bench_1.json 2> gpurun_out/r4l_bench.err; tail -2 gpurun_out/r4l_bench.err
( timeout 900 python bench.py --no-cpu-baseline ) > gpurun_out/r4l_bench_2.json 2>> gpurun_out/r4l_bench.err
for i in 1 2; do
python - $i <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r4l_bench_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
for k, v in d["reference_nco"]["legs"].items():
    print(k, v["value"], "call", v["call_ms"], "host", v["host_walk_and_candidates_ms"], "chain", v["host_chain_only_ms"], "eval", v["host_evaluation_only_ms"], v["bound"])
print("e2e reference", d["reference_nco"]["end_to_end"]["value"], "device_dst_batch", d["extra"]["device_dst_batch"]["value"], "value", d["value"], "streamed", d["end_to_end"]["streamed"]["value"])
print("roofline", d["roofline"], "cpu", d.get("cpu_baseline"), "rccl", d.get("rccl_selftest"), "stale", d.get("stale_profile"))
PY
done
