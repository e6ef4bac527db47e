# round 4: where a 25 Msps reference-NCO batch call spends its 2 ms -- per-piece timeline, piece-size variants
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
cat > /tmp/t25.py <<'PY'
import os, sys, time
sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "multi-sdr-gps-sim_amd"))
import numpy as np, torch, gpsiq
from gpsiq.abi import NCO_REFERENCE
from gpsiq.scenario import synth_blocks
ctx = gpsiq.Context(0); ctx.set_nco_mode(NCO_REFERENCE)
ring = torch.empty((2 << 30) + (64 << 20), dtype=torch.uint8, device="cuda")
pat = synth_blocks(64, 16, seed=20250215)
d = pat[np.arange(200) % 64]
best = 1e9
for i in range(int(sys.argv[1])):
    t = time.perf_counter(); ctx.generate_batch(d, 2500000, 25e6, 2, device_ptr=ring.data_ptr()); best = min(best, time.perf_counter() - t)
print("best call %.3f ms" % (best * 1e3), {k: os.environ[k] for k in os.environ if k.startswith("GPSIQ_")})
PY
GPSIQ_TRACE=2 python /tmp/t25.py 4 2>&1 | grep -v "trace\] descriptors" | tail -34
for v in "GPSIQ_REF_CHUNK_BLOCKS=13" "GPSIQ_REF_CHUNK_BLOCKS=40" "GPSIQ_REF_CHUNK_BLOCKS=60" "GPSIQ_REF_CHUNK_RAMP=0" "GPSIQ_REF_CHUNK_RAMP=0 GPSIQ_REF_CHUNK_BLOCKS=50" "GPSIQ_THREADS=8" "GPSIQ_THREADS=12" "GPSIQ_NO_DRIFT=0"; do
  env $v python /tmp/t25.py 12 2>&1 | tail -1
done
GPSIQ_TRACE=2 GPSIQ_REF_CHUNK_RAMP=0 GPSIQ_REF_CHUNK_BLOCKS=50 python /tmp/t25.py 3 2>&1 | grep -v "trace\] descriptors" | tail -6
