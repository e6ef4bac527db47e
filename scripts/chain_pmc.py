"""The device carrier chain alone (gpsiq_chain_maps_device), for rocprofv3 passes: a few calls at 25 Msps / 200 blocks and at
2.6 Msps / 2000 blocks.  No torch (a profiled interpreter with torch loaded has been seen to hang in teardown: run it under `timeout`)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multi-sdr-gps-sim_amd"))
import gpsiq  # noqa: E402
from gpsiq.scenario import synth_blocks  # noqa: E402

ctx = gpsiq.Context(0)
pat = synth_blocks(64, 16)
for fs, nb, seg in ((25e6, 200, 16), (2.6e6, 2000, 32), (2.6e6, 2000, 16)):
    ns = int(round(fs / 10))
    cin = gpsiq.chain_inputs(pat[np.arange(nb) % 64])
    for _ in range(4):
        ms = gpsiq.chain_maps(cin, fs, ns, max_stretches=seg, ctx=ctx)[2]
    print(f"{fs / 1e6:g} Msps, {nb} blocks, {seg} stretches: kernels {ms:.3f} ms", flush=True)
sys.stdout.flush()
ctx.close()
