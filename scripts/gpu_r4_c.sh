# round 4, third GPU session: asynchronous piece submission + 16-lane apply_patches.  Reference-NCO tests (incl. two contexts on two
# host threads, GPSIQ_THREADS=2, config 5's two shares against the reference's digests), piece timings and trace, default bench,
# kernel trace of the reference batch, then the rocprofv3 passes for profiles/ (PROF_TAG=r04)
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests/test_gpu_reference_nco.py tests/test_config5_shares.py tests/test_gpu_parity.py tests/test_reference_program.py tests/test_host_c.py -m gpu -q -x -s --durations=8 2>&1 | tail -30 ) > gpurun_out/r4c_pytest_gpu.log 2>&1; tail -22 gpurun_out/r4c_pytest_gpu.log
python /dev/stdin > gpurun_out/r4c_ref_pieces.txt 2>&1 <<'PY'
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
    for chunk in (None, "32", "64", "128", "512") if fs < 1e7 else (None, "4", "8", "13", "52"):
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
PY
grep -v "trace\] descriptors" gpurun_out/r4c_ref_pieces.txt
( timeout 900 python bench.py ) > gpurun_out/r4c_bench.json 2> gpurun_out/r4c_bench.err; tail -3 gpurun_out/r4c_bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4c_bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "roofline", d["roofline"], "counters", d.get("counters"))
for k, v in d["reference_nco"]["legs"].items():
    print(k, json.dumps({a: b for a, b in v.items() if not isinstance(b, dict)}))
print("device_dst_batch", d["extra"]["device_dst_batch"], "block_call", d["extra"]["block_call"]["median_us"], d["extra"]["block_call_reference_nco"]["median_us"], d["extra"]["block_call_async_reference_nco"])
PY
PROF_TAG=r04 bash scripts/gpu_prof.sh > gpurun_out/r4c_prof.log 2>&1; tail -5 gpurun_out/r4c_prof.log
