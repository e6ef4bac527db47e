# round 3, sixth GPU session: where the streamed end-to-end leg spends its rounds (GPSIQ_TRACE), bench with more rounds
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( GPSIQ_TRACE=1 timeout 600 python bench.py --no-cpu-baseline --no-extra --steps 5 ) > gpurun_out/r3f_bench_trace.json 2> gpurun_out/r3f_bench_trace.err
grep -c "descriptors" gpurun_out/r3f_bench_trace.err; grep "descriptors 4130" gpurun_out/r3f_bench_trace.err | tail -20
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3f_bench_trace.json").read().strip().splitlines()[-1])
print(d["value"], d["end_to_end"]["value"], d["end_to_end"]["streamed"])
PY
( timeout 600 python bench.py --no-cpu-baseline --no-extra --steps 5 --rounds 32 ) > gpurun_out/r3f_bench_r32.json 2>/dev/null
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3f_bench_r32.json").read().strip().splitlines()[-1])
print(d["value"], d["end_to_end"]["streamed"]["value"], d["end_to_end"]["streamed"]["seconds_each_pass"], d["end_to_end"]["streamed"]["per_rank"])
PY
