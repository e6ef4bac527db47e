# round 4, eighth GPU session: the whole GPU suite, smoke, the default bench line, a 2-rank run on the one GPU (gloo), reference-NCO
# piece timings at HEAD
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 2400 python -m pytest tests -m gpu -q --durations=8 2>&1 | tail -16 ) > gpurun_out/r4h_pytest_gpu.log 2>&1; tail -12 gpurun_out/r4h_pytest_gpu.log
( timeout 300 python __graft_entry__.py smoke ) > gpurun_out/r4h_smoke.log 2>&1; tail -1 gpurun_out/r4h_smoke.log
( timeout 900 python bench.py ) > gpurun_out/r4h_bench.json 2> gpurun_out/r4h_bench.err; tail -2 gpurun_out/r4h_bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4h_bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "roofline", d["roofline"]["frac"], d["roofline"]["traffic"], "counters", (d.get("counters") or {}).get("valu_issue_frac"), "selftest", d["extra"]["rccl_selftest"]["ok"])
for k, v in d["reference_nco"]["legs"].items():
    print(k, json.dumps({a: b for a, b in v.items() if not isinstance(b, dict)}))
print("e2e", d["end_to_end"]["value"], d["end_to_end"]["streamed"]["value"], d["end_to_end"]["streamed"]["per_rank"])
print("device_dst_batch", d["extra"]["device_dst_batch"]["value"], "block_call", d["extra"]["block_call"]["median_us"], d["extra"]["block_call_reference_nco"]["median_us"], d["extra"]["block_call_async"]["us_per_block"], d["extra"]["block_call_async_reference_nco"]["us_per_block"])
PY
( GPSIQ_BENCH_SHARE_GPU=1 GPSIQ_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 5 --warmup 1 --blocks 1000 --launches 4 ) > gpurun_out/r4h_bench_2rank.log 2>&1
python - <<'PY'
import json
for ln in open("gpurun_out/r4h_bench_2rank.log"):
    if ln.startswith("{"):
        d = json.loads(ln)
        print("2 ranks: value", d["value"]); print(json.dumps(d["reference_nco"])[:2500])
PY
python /dev/stdin > gpurun_out/r4h_ref_pieces.txt 2>&1 <<'PY'
import os, sys, time
sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "multi-sdr-gps-sim_amd"))
import numpy as np, torch, gpsiq
from gpsiq.abi import NCO_REFERENCE
from gpsiq.scenario import synth_blocks
ctx = gpsiq.Context(0); ctx.set_nco_mode(NCO_REFERENCE)
ring = torch.empty((2 << 30) + (64 << 20), dtype=torch.uint8, device="cuda")
pat = synth_blocks(64, 16, seed=20250215)
for fs, ss, nb in ((25e6, 2, 200), (10e6, 2, 536), (2.6e6, 1, 2000), (2.6e6, 1, 8000)):
    d = pat[np.arange(nb) % 64]
    best = 1e9
    for _ in range(10):
        t = time.perf_counter(); ctx.generate_batch(d, int(fs) // 10, fs, ss, device_ptr=ring.data_ptr()); best = min(best, time.perf_counter() - t)
    print("fs %.1f, %d blocks: call %.3f ms = %.1f Gsamples/s" % (fs / 1e6, nb, best * 1e3, nb * fs / 10 / best / 1e9), flush=True)
    os.environ["GPSIQ_TRACE"] = "1"
    ctx.generate_batch(d, int(fs) // 10, fs, ss, device_ptr=ring.data_ptr())
    os.environ.pop("GPSIQ_TRACE")
PY
grep -v "trace\] descriptors" gpurun_out/r4h_ref_pieces.txt
