# round 4: piece sizes by bound (kernel-bound: few growing pieces), batch + reference GPU tests, default bench twice
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_nco.py tests/test_gpu_long_runs.py tests/test_config5_shares.py tests/test_host_c.py -m gpu -q -x 2>&1 | tail -4 )
for i in 1 2; do
( timeout 900 python bench.py --no-cpu-baseline ) > gpurun_out/r4k_bench_$i.json 2> gpurun_out/r4k_bench.err; tail -1 gpurun_out/r4k_bench.err
python - $i <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r4k_bench_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
for k, v in d["reference_nco"]["legs"].items():
    print(k, v["value"], "call", v["call_ms"], "host", v["host_walk_and_candidates_ms"], "chain", v["host_chain_only_ms"], "eval", v["host_evaluation_only_ms"], v["bound"])
print("e2e reference", d["reference_nco"]["end_to_end"]["value"], "device_dst_batch", d["extra"]["device_dst_batch"]["value"], "value", d["value"], "streamed", d["end_to_end"]["streamed"]["value"])
PY
done
GPSIQ_TRACE=1 python - <<'PY' 2>&1 | grep -v "trace\] descriptors" | tail -8
import os, sys, time
sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "multi-sdr-gps-sim_amd"))
import numpy as np, torch, gpsiq
from gpsiq.abi import NCO_REFERENCE
from gpsiq.scenario import synth_blocks
ctx = gpsiq.Context(0); ctx.set_nco_mode(NCO_REFERENCE)
ring = torch.empty((2 << 30) + (64 << 20), dtype=torch.uint8, device="cuda")
pat = synth_blocks(64, 16, seed=20250215)
for fs, ss, nb in ((25e6, 2, 200), (10e6, 2, 536)):
    d = pat[np.arange(nb) % 64]
    for _ in range(3): ctx.generate_batch(d, int(fs) // 10, fs, ss, device_ptr=ring.data_ptr())
PY
