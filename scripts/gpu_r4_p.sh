# round 4: gpsiq_generate_batch (fixed-point model) after the context-owned piece buffer + smaller first piece; kernel trace of it
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_long_runs.py tests/test_host_c.py tests/test_pipeline.py -m gpu -q -x 2>&1 | tail -3 )
cp scripts/gpu_r4_o.sh /tmp/o.sh
sed -n '/^cat > \/tmp\/tb.py/,/^PY$/p' scripts/gpu_r4_o.sh | sed '1d;$d' > /tmp/tb.py
GPSIQ_TRACE=2 python /tmp/tb.py 4 2>&1 | grep -v "trace\] descriptors" | tail -9
for v in "GPSIQ_POOL_LINGER_US=0" "GPSIQ_POOL_LINGER_US=200" "GPSIQ_PIECE_GROWTH=160" "GPSIQ_PIECE_GROWTH=180" "GPSIQ_PIECE_GROWTH=200" "GPSIQ_PIECE_GROWTH=300" "GPSIQ_BATCH_PIECE_BLOCKS=2048 GPSIQ_PIECE_GROWTH=180" "GPSIQ_THREADS=1"; do
  env $v python /tmp/tb.py 12 2>&1 | tail -1
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_tb -- python /tmp/tb.py 6 > /tmp/prof_tb.log 2>&1; tail -1 /tmp/prof_tb.log
python - <<'PY'
import csv, glob
f = glob.glob("/tmp/prof_tb/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
synth = [r for r in rows if "synth" in r["Kernel_Name"]]
n = len(synth)
per = 6 if n % 6 == 0 else None
print("synth launches", n)
last = synth[-6:] if n >= 6 else synth
t0 = int(last[0]["Start_Timestamp"])
for r in last:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("  start %8.1f us  end %8.1f us  dur %8.1f us  grid %s" % (s / 1e3, e / 1e3, (e - s) / 1e3, r.get("Grid_Size")))
PY
cd $GRAFT_REPO_ROOT
for l in 0 200; do
( GPSIQ_POOL_LINGER_US=$l timeout 900 python bench.py --no-cpu-baseline ) > gpurun_out/r4p_bench_linger$l.json 2> gpurun_out/r4p_bench.err; tail -1 gpurun_out/r4p_bench.err
python - $l <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r4p_bench_linger%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
print("linger", sys.argv[1])
for k, v in d["reference_nco"]["legs"].items():
    print(" ", k, v["value"], "call", v["call_ms"], "host", v["host_walk_and_candidates_ms"], "chain", v["host_chain_only_ms"], "eval", v["host_evaluation_only_ms"], v["bound"])
print("  e2e reference", d["reference_nco"]["end_to_end"]["value"], "device_dst_batch", d["extra"]["device_dst_batch"]["value"], "value", d["value"], "streamed", d["end_to_end"]["streamed"]["value"], "e2e", d["end_to_end"]["value"])
PY
done
