# round 4, sixth GPU session (host microbenchmarks on the EPYC + a check of the symmetric piece ramp): bucket count of the wrap-to-wrap table
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for b in 256 512 1024 2048; do
  g++ -O3 -std=c++17 -ffp-contract=off -DBUCKETS=$b -I multi-sdr-gps-sim_amd/csrc scripts/ubench_walk.cpp multi-sdr-gps-sim_amd/csrc/gpsiq_host.cpp -lpthread -o /tmp/walk_$b
done
( for rep in 1 2; do for b in 256 512 1024 2048; do echo "== BUCKETS = $b (pass $rep) =="; taskset -c 5 /tmp/walk_$b; done; done ) > gpurun_out/r4f_ubench_buckets.txt 2>&1; cat gpurun_out/r4f_ubench_buckets.txt
python /dev/stdin > gpurun_out/r4f_ref_pieces.txt 2>&1 <<'PY'
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
PY
grep -v "trace\] descriptors" gpurun_out/r4f_ref_pieces.txt
( timeout 600 python -m pytest tests/test_host_c.py tests/test_config5_shares.py -m gpu -q -x 2>&1 | tail -5 )
