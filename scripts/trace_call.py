import os, sys, time
import numpy as np
ROOT = os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0, os.path.join(ROOT, "multi-sdr-gps-sim_amd"))
import gpsiq
from gpsiq.abi import NCO_REFERENCE
from gpsiq.scenario import synth_blocks
import torch
ctx = gpsiq.Context(0)
ring = torch.empty(2 << 30, dtype=torch.uint8, device="cuda:0")
pat = synth_blocks(64, 16)
ctx.set_nco_mode(NCO_REFERENCE)
d = pat[np.arange(2000) % 64]
for _ in range(12):
    t0 = time.perf_counter()
    ctx.generate_batch(d, 260000, 2.6e6, 1, device_ptr=ring.data_ptr())
    print("call ms", (time.perf_counter() - t0) * 1e3, flush=True)
    time.sleep(0.02)
sys.stdout.flush()
os._exit(0) if os.environ.get("HARD_EXIT") else None
