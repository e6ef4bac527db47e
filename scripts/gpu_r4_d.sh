# round 4, fourth GPU session: container-aware thread count (cgroup cpu.max), targeted task wake-ups.  Reference-NCO tests incl. config 5's
# shares, piece timings, default bench
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests/test_gpu_reference_nco.py tests/test_config5_shares.py -m gpu -q -x -s --durations=8 2>&1 | tail -30 ) > gpurun_out/r4d_pytest_gpu.log 2>&1; tail -22 gpurun_out/r4d_pytest_gpu.log
python /dev/stdin > gpurun_out/r4d_ref_pieces.txt 2>&1 <<'PY'
import os, sys, time
sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "multi-sdr-gps-sim_amd"))
import numpy as np, torch, gpsiq
from gpsiq.abi import NCO_REFERENCE
from gpsiq.scenario import synth_blocks
ctx = gpsiq.Context(0); ctx.set_nco_mode(NCO_REFERENCE)
ring = torch.empty((2 << 30) + (64 << 20), dtype=torch.uint8, device="cuda")
pat = synth_blocks(64, 16, seed=20250215)
for fs, ss, nb in ((25e6, 2, 200), (10e6, 2, 536), (2.6e6, 1, 2000)):
    d = pat[np.arange(nb) % 64]
    for chunk in (None, "64", "128", "512") if fs < 1e7 else (None, "8", "13", "52"):
        if chunk: os.environ["GPSIQ_REF_CHUNK_BLOCKS"] = chunk
        else: os.environ.pop("GPSIQ_REF_CHUNK_BLOCKS", None)
        best = 1e9
        for _ in range(6):
            t = time.perf_counter(); ctx.generate_batch(d, int(fs) // 10, fs, ss, device_ptr=ring.data_ptr()); best = min(best, time.perf_counter() - t)
        print("fs %.1f chunk %s: call %.3f ms = %.1f Gsamples/s" % (fs / 1e6, chunk, best * 1e3, nb * fs / 10 / best / 1e9), flush=True)
    os.environ.pop("GPSIQ_REF_CHUNK_BLOCKS", None)
    os.environ["GPSIQ_TRACE"] = "1"
    ctx.generate_batch(d, int(fs) // 10, fs, ss, device_ptr=ring.data_ptr())
    os.environ.pop("GPSIQ_TRACE")
    for _ in range(3):
        t = time.perf_counter(); q, p, ce = gpsiq.reference_blocks(d, fs, int(fs) // 10); th = time.perf_counter() - t
        cin = gpsiq.chain_inputs(d)
        t = time.perf_counter(); st, _, _ = gpsiq.reference_chain(cin, fs, int(fs) // 10); tc = time.perf_counter() - t
        t = time.perf_counter(); gpsiq.reference_seeded(d, fs, int(fs) // 10, st); te = time.perf_counter() - t
    print("fs %.1f host: whole %.3f ms, chain only %.3f ms, evaluation only %.3f ms" % (fs / 1e6, th * 1e3, tc * 1e3, te * 1e3), flush=True)
PY
grep -v "trace\] descriptors" gpurun_out/r4d_ref_pieces.txt
( timeout 900 python bench.py ) > gpurun_out/r4d_bench.json 2> gpurun_out/r4d_bench.err; tail -3 gpurun_out/r4d_bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4d_bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "roofline", d["roofline"]["frac"], d["roofline"]["traffic"], "counters", (d.get("counters") or {}).get("valu_issue_frac"))
for k, v in d["reference_nco"]["legs"].items():
    print(k, json.dumps({a: b for a, b in v.items() if not isinstance(b, dict)}))
print("e2e", d["end_to_end"]["value"], d["end_to_end"]["streamed"]["value"], d["end_to_end"]["streamed"]["per_rank"])
print("device_dst_batch", d["extra"]["device_dst_batch"]["value"], "block_call", d["extra"]["block_call"]["median_us"], d["extra"]["block_call_reference_nco"]["median_us"])
PY
