# round 4, seventh GPU session: the wrap-to-wrap table built eight cycles at a time (AVX-512) on the EPYC host: microbenchmarks with and
# without (GPSIQ_WALK_NOBATCH=1), reference-NCO piece timings, the reference-NCO GPU tests, default bench
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
g++ -O3 -std=c++17 -ffp-contract=off -I multi-sdr-gps-sim_amd/csrc scripts/ubench_walk.cpp multi-sdr-gps-sim_amd/csrc/gpsiq_host.cpp -lpthread -o /tmp/walk
g++ -O3 -std=c++17 -ffp-contract=off -I multi-sdr-gps-sim_amd/csrc scripts/ubench_refhost.cpp multi-sdr-gps-sim_amd/csrc/gpsiq_host.cpp -lpthread -o /tmp/refhost
g++ -O2 -std=c++17 -ffp-contract=off -I include -I multi-sdr-gps-sim_amd/csrc -o /tmp/batch_walk tests/batch_walk.cpp multi-sdr-gps-sim_amd/csrc/gpsiq_host.cpp -lpthread -lm && /tmp/batch_walk 11
( echo "== eight cycles at a time (AVX-512) =="; taskset -c 5 /tmp/walk; echo "== GPSIQ_WALK_NOBATCH=1: every cycle of the table walked when the chain reaches it =="; GPSIQ_WALK_NOBATCH=1 taskset -c 5 /tmp/walk ) > gpurun_out/r4g_ubench_walk.txt 2>&1; cat gpurun_out/r4g_ubench_walk.txt
( taskset -c 5 /tmp/refhost; echo "== GPSIQ_WALK_NOBATCH=1 =="; GPSIQ_WALK_NOBATCH=1 taskset -c 5 /tmp/refhost ) > gpurun_out/r4g_ubench_refhost.txt 2>&1; cat gpurun_out/r4g_ubench_refhost.txt
python /dev/stdin > gpurun_out/r4g_ref_pieces.txt 2>&1 <<'PY'
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
    best = 1e9
    for _ in range(10):
        t = time.perf_counter(); ctx.generate_batch(d, int(fs) // 10, fs, ss, device_ptr=ring.data_ptr()); best = min(best, time.perf_counter() - t)
    print("fs %.1f: call %.3f ms = %.1f Gsamples/s" % (fs / 1e6, best * 1e3, nb * fs / 10 / best / 1e9), flush=True)
    os.environ["GPSIQ_TRACE"] = "1"
    ctx.generate_batch(d, int(fs) // 10, fs, ss, device_ptr=ring.data_ptr())
    os.environ.pop("GPSIQ_TRACE")
    th = tc = 1e9
    cin = gpsiq.chain_inputs(d)
    for _ in range(4):
        t = time.perf_counter(); gpsiq.reference_blocks(d, fs, int(fs) // 10); th = min(th, time.perf_counter() - t)
        t = time.perf_counter(); gpsiq.reference_chain(cin, fs, int(fs) // 10); tc = min(tc, time.perf_counter() - t)
    print("fs %.1f host: whole %.3f ms, chain only %.3f ms" % (fs / 1e6, th * 1e3, tc * 1e3), flush=True)
PY
grep -v "trace\] descriptors" gpurun_out/r4g_ref_pieces.txt
( timeout 900 python -m pytest tests/test_gpu_reference_nco.py tests/test_config5_shares.py tests/test_config4.py -m gpu -q -x 2>&1 | tail -4 )
( timeout 900 python bench.py ) > gpurun_out/r4g_bench.json 2> gpurun_out/r4g_bench.err; tail -2 gpurun_out/r4g_bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4g_bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "roofline", d["roofline"]["frac"], d["roofline"]["traffic"], "counters", (d.get("counters") or {}).get("valu_issue_frac"))
for k, v in d["reference_nco"]["legs"].items():
    print(k, json.dumps({a: b for a, b in v.items() if not isinstance(b, dict)}))
PY
