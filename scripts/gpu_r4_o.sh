# round 4: where gpsiq_generate_batch (fixed-point model, gpsiq_chan_t -> device memory, 4130 blocks at 2.6 Msps) spends its time
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
cat > /tmp/tb.py <<'PY'
import os, sys, time
sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "multi-sdr-gps-sim_amd"))
import numpy as np, torch, gpsiq
from gpsiq.scenario import synth_blocks
ctx = gpsiq.Context(0)
ring = torch.empty((2 << 30) + (64 << 20), dtype=torch.uint8, device="cuda")
pat = synth_blocks(64, 16, seed=20250215)
nb = 4130
d = np.ascontiguousarray(pat[np.arange(nb) % 64])
best = 1e9
for i in range(int(sys.argv[1])):
    t = time.perf_counter(); ctx.generate_batch(d, 260000, 2.6e6, 1, device_ptr=ring.data_ptr()); best = min(best, time.perf_counter() - t)
print("best call %.3f ms = %.1f G samples/s" % (best * 1e3, nb * 260000 / best / 1e9), {k: os.environ[k] for k in os.environ if k.startswith("GPSIQ_")})
PY
GPSIQ_TRACE=2 python /tmp/tb.py 4 2>&1 | grep -v "trace\] descriptors" | tail -12
for v in "GPSIQ_BATCH_PIECE_BLOCKS=128" "GPSIQ_BATCH_PIECE_BLOCKS=256" "GPSIQ_BATCH_PIECE_BLOCKS=512" "GPSIQ_BATCH_PIECE_BLOCKS=1024" "GPSIQ_BATCH_PIECE_BLOCKS=2048" "GPSIQ_REF_CHUNK_RAMP=0 GPSIQ_BATCH_PIECE_BLOCKS=1024" "GPSIQ_THREADS=4"; do
  env $v python /tmp/tb.py 12 2>&1 | tail -1
done
GPSIQ_TRACE=2 GPSIQ_BATCH_PIECE_BLOCKS=256 python /tmp/tb.py 3 2>&1 | grep -v "trace\] descriptors" | tail -8
