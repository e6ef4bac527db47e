# round 4, second GPU session: the reference-NCO GPU tests on the task-based host side, the default bench line (RCCL self-test,
# reference_nco with chain-only / evaluation-only), a 2-rank run on the one GPU (gloo: the sharded reference_nco leg on a device),
# the host microbenchmarks on this host (EPYC 9575F), GPSIQ_TRACE of the reference batch
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests/test_gpu_reference_nco.py tests/test_gpu_parity.py tests/test_reference_program.py tests/test_config4.py -m gpu -q -x --durations=5 2>&1 | tail -15 ) > gpurun_out/r4b_pytest_gpu.log 2>&1; tail -8 gpurun_out/r4b_pytest_gpu.log
( timeout 900 python bench.py ) > gpurun_out/r4b_bench.json 2> gpurun_out/r4b_bench.err; tail -3 gpurun_out/r4b_bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4b_bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "roofline", d["roofline"]["frac"])
print("rccl_selftest", json.dumps(d["extra"].get("rccl_selftest")))
for k, v in d["reference_nco"]["legs"].items():
    print(k, json.dumps({a: b for a, b in v.items() if not isinstance(b, dict)}))
PY
( GPSIQ_BENCH_SHARE_GPU=1 GPSIQ_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 5 --warmup 1 --blocks 1000 --launches 4 ) > gpurun_out/r4b_bench_2rank.log 2>&1
python - <<'PY'
import json
for ln in open("gpurun_out/r4b_bench_2rank.log"):
    if ln.startswith("{"):
        d = json.loads(ln)
        print("2 ranks: value", d["value"]); print(json.dumps(d["reference_nco"])[:3000])
PY
g++ -O3 -std=c++17 -ffp-contract=off -I multi-sdr-gps-sim_amd/csrc scripts/ubench_refhost.cpp multi-sdr-gps-sim_amd/csrc/gpsiq_host.cpp -lpthread -o /tmp/refhost && taskset -c 3 /tmp/refhost > gpurun_out/r4b_ubench_refhost.txt 2>&1; cat gpurun_out/r4b_ubench_refhost.txt
g++ -O3 -std=c++17 -ffp-contract=off -I multi-sdr-gps-sim_amd/csrc scripts/ubench_walk.cpp multi-sdr-gps-sim_amd/csrc/gpsiq_host.cpp -lpthread -o /tmp/walk && taskset -c 3 /tmp/walk > gpurun_out/r4b_ubench_walk.txt 2>&1; cat gpurun_out/r4b_ubench_walk.txt
cat > /tmp/ref_trace.py <<'PY'
import os, sys, time
sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "multi-sdr-gps-sim_amd"))
import numpy as np, torch, gpsiq
from gpsiq.abi import NCO_REFERENCE
from gpsiq.scenario import synth_blocks
ctx = gpsiq.Context(0); ctx.set_nco_mode(NCO_REFERENCE)
ring = torch.empty((2 << 30) + (64 << 20), dtype=torch.uint8, device="cuda")
pat = synth_blocks(64, 16, seed=20250215)
for fs, ss, nb in ((25e6, 2, 200), (2.6e6, 1, 2000)):
    d = pat[np.arange(nb) % 64]
    for chunk in (None, "64", "128", "512") if fs < 1e7 else (None, "8", "13", "52"):
        if chunk: os.environ["GPSIQ_REF_CHUNK_BLOCKS"] = chunk
        else: os.environ.pop("GPSIQ_REF_CHUNK_BLOCKS", None)
        best = 1e9
        for _ in range(5):
            t = time.perf_counter(); ctx.generate_batch(d, int(fs) // 10, fs, ss, device_ptr=ring.data_ptr()); best = min(best, time.perf_counter() - t)
        print("fs %.1f chunk %s: call %.3f ms" % (fs / 1e6, chunk, best * 1e3), flush=True)
    os.environ.pop("GPSIQ_REF_CHUNK_BLOCKS", None)
    os.environ["GPSIQ_TRACE"] = "1"
    ctx.generate_batch(d, int(fs) // 10, fs, ss, device_ptr=ring.data_ptr())
    os.environ.pop("GPSIQ_TRACE")
PY
python /tmp/ref_trace.py > gpurun_out/r4b_ref_pieces.txt 2>&1; cat gpurun_out/r4b_ref_pieces.txt
